#!/usr/bin/env python3
"""r05 probe: can the idle issue slots of one kernel be filled by another?  The images of a step in G groups on streams of their own
(hm_batch_set_concurrency), with the chain kernel's workgroups of 4 waves (5 per CU: all of its LDS) or of 8 (HM_CHAIN_NP=8: 2 per CU =
16 waves, which leaves 36 KB of LDS, 128 registers per lane and four wave slots per SIMD to the other kernels).  ms per 384 images."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import __graft_entry__ as g
    import bench
    pkg = g.load_package(test_knobs=True)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    B = int(os.environ.get("IMAGES", "384"))
    NT = 48
    gb = bench.GridBatch(pkg, dev, 8, 6, 512, 4032, 3024)
    made = bench.make_streams(pkg.capi, (1200000 + k for k in range(B * NT)), keep_data=False)
    for j in range(B):
        gb.add_image([next(made)[1] for _ in range(NT)])
    gb.finish(st, 0)
    res = {}
    for groups in [int(x) for x in os.environ.get("GROUPS", "0,2,3,4,6,8").split(",")]:
        gb.batch.set_concurrency(groups)
        for _ in range(2):
            gb.step(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            gb.step(st)
        torch.cuda.synchronize()
        res[groups] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    gb.batch.check()
    print(json.dumps({"chain_np": os.environ.get("HM_CHAIN_NP", "default"), "ms_per_step_by_groups": res}))


if __name__ == "__main__":
    main()
