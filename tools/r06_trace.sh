#!/bin/bash
# wall-clock laps of hm_decode_item for one 12 MP grid (HM_TRACE=1), 16 host threads
HM_TRACE=1 python3 - <<'PY' 2>&1 | tail -40
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench, heifwriter, pipeline
import __graft_entry__ as g
pkg = g.load_package()
tiles = [bench.tile_stream(9100 + i) for i in range(48)]
data = heifwriter.write_heic(tiles, (bench.TILE, bench.TILE), grid=(bench.GRID_ROWS, bench.GRID_COLS, bench.OUT_W, bench.OUT_H))
f = pipeline.HeifFile(pkg.lib(), data)
for _ in range(3):
    f.decode(f.primary(), 10, threads=16, copy=False)
print("==== traced call", flush=True)
t0 = time.perf_counter(); f.decode(f.primary(), 10, threads=16, copy=False); print("total ms", (time.perf_counter() - t0) * 1e3)
f.close()
PY
