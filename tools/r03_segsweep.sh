#!/bin/bash
mkdir -p gpurun_out
{
for s in 1 4 7 12 16 30; do echo -n "HM_RESID_SEGS=$s (cap lifted by force): "; HM_RESID_SEGS=$s python3 - <<'PY' 2>/dev/null | tail -1
import json, os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import torch
import __graft_entry__ as g
import bench
pkg = g.load_package(); torch.cuda.set_device(0); dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
rc = bench.real_content(torch, pkg, dev, st)
print(json.dumps({k: (v["MP_per_s"], v["ms_per_MP"]["k_residual"], v["ms_per_MP"]["k_chain"]) for k, v in rc.items() if isinstance(v, dict)}))
PY
done
} > gpurun_out/r03_segsweep.log 2>&1
