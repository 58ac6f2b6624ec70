#!/usr/bin/env python3
"""Throughput probe of hm_pipeline_* (GPU box): N different 12 MP grid .heic files, several crew sizes / depths.
usage: HM_PIPELINE_STATS=1 python3 tools/pipeline_probe.py [n_files]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import heifwriter  # noqa: E402
import pipeline  # noqa: E402
import __graft_entry__ as g  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pkg = g.load_package()
hm = pkg.lib()
NT = 48
made = bench.make_streams(pkg.capi, (1200000 + k for k in range(n * NT)))
files = []
for j in range(n):
    tiles = [next(made)[0] for _ in range(NT)]
    files.append(heifwriter.write_heic(tiles, (512, 512), grid=(6, 8, 4032, 3024)))
print("files ready", len(files), sum(map(len, files)) / 1e6, "MB", flush=True)
for threads, depth in ((14, 16), (15, 16), (16, 16), (17, 16), (18, 16), (20, 24), (24, 32), (32, 32)):
    pl = pipeline.Pipeline(hm, 10, host_threads=threads, max_in_flight=depth)
    for rnd in range(2):
        pend = 0
        t0 = time.perf_counter()
        for i, data in enumerate(files):
            while not pl.submit(data, i):
                pl.next(copy=False); pend -= 1
            pend += 1
        while pend:
            pl.next(copy=False); pend -= 1
        dt = time.perf_counter() - t0
    print(f"threads {threads:4d} depth {depth:3d}: {dt / n * 1e3:7.3f} ms/image  {n * 12.19 / dt / 1e3:6.2f} GP/s", flush=True)
    pl.close()
