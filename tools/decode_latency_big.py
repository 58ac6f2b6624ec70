#!/usr/bin/env python3
"""hm_decode_item on ONE 16384 x 16384 grid (BASELINE config 5's shape: 32 x 32 tiles of 512 x 512, 24 distinct tiles), 16 host threads: the slabs under the entropy
decode (default) against the one batch behind it (HM_GRID_SLAB_ROWS=0).  usage (repo root, GPU box): [HM_GRID_SLAB_ROWS=n] python3 tools/decode_latency_big.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, heifwriter, pipeline
import __graft_entry__ as g
pkg = g.load_package(test_knobs=True)
pool = [bench.tile_stream(5000000 + i) for i in range(24)]
data = heifwriter.write_heic([pool[(7 * t + 3 * (t // 32)) % 24] for t in range(1024)], (512, 512), grid=(32, 32, 16384, 16384))
f = pipeline.HeifFile(pkg.lib(), data)
for _ in range(2):
    f.decode(f.primary(), 10, threads=16, copy=False)
ts = []
for _ in range(6):
    t0 = time.perf_counter(); f.decode(f.primary(), 10, threads=16, copy=False); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(f"16384 x 16384 grid, 16 threads: median {ts[len(ts) // 2]:.1f} ms, best {ts[0]:.1f} = {268.4 / ts[0]:.2f} GP/s")
f.close()
