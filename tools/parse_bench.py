"""Host entropy decode alone (hm_hevc_parse: CABAC -> command stream) on the bench's tile streams, one thread:
MP/s per core, bytes per picture, best of several passes.  CPU only.
usage: python3 tools/parse_bench.py [n_streams] [passes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import importlib

pkg = importlib.import_module("heif-decoder-lib_amd")
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    streams = [bench.tile_stream(9100 + i) for i in range(n)]
    nbytes = sum(len(s) for s in streams)
    best = 1e9
    for _ in range(passes):
        t0 = time.perf_counter()
        for s in streams:
            pkg.capi.parse_hevc(s)
        best = min(best, time.perf_counter() - t0)
    mp = n * bench.TILE * bench.TILE / 1e6
    print(f"{mp / best:.1f} MP/s/core  ({best / n * 1e3:.3f} ms per {bench.TILE}x{bench.TILE} tile, {nbytes / n:.0f} bytes per tile = {8 * nbytes / (mp * 1e6):.2f} bits/px)")


if __name__ == "__main__":
    main()
