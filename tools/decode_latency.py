#!/usr/bin/env python3
"""hm_decode_item on ONE 12 MP grid (48 tiles of the bench), mean of N calls per thread count; HM_GRID_SLAB_ROWS = 0 / n through the test hook.
usage (repo root, GPU box): [HM_GRID_SLAB_ROWS=n] python3 tools/decode_latency.py [threads ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, heifwriter, pipeline
import __graft_entry__ as g
pkg = g.load_package(test_knobs=True)
tiles = [bench.tile_stream(9100 + i) for i in range(48)]
data = heifwriter.write_heic(tiles, (bench.TILE, bench.TILE), grid=(bench.GRID_ROWS, bench.GRID_COLS, bench.OUT_W, bench.OUT_H))
f = pipeline.HeifFile(pkg.lib(), data)
for th in [int(a) for a in sys.argv[1:]] or [16, 8]:
    for _ in range(5):
        f.decode(f.primary(), 10, threads=th, copy=False)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); f.decode(f.primary(), 10, threads=th, copy=False); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"threads {th}: median {ts[len(ts) // 2]:.3f} ms, best {ts[0]:.3f}, mean {sum(ts) / len(ts):.3f}")
f.close()
