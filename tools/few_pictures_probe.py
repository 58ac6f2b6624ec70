#!/usr/bin/env python3
"""GPU box: the few-large-pictures legs of bench.py alone (real 1080p content, config 4, one 12 MP picture as 48 tiles):
the light-load regime of the prediction-chain kernel, where an iteration's latency counts rather than its instructions."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

pkg = g.load_package(test_knobs=True)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
rc = bench.real_content(torch, pkg, dev, st)
c4 = bench.config4(torch, pkg, dev, st)
print(json.dumps({"real_content_MP_per_s": {k: v["MP_per_s"] for k, v in rc.items() if isinstance(v, dict)},
                  "config4_MP_per_s": c4["MP_per_s"], "config4_kernels": c4["kernels_ms_per_step"]}))
