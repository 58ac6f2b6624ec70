import importlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
pkg = importlib.import_module("heif-decoder-lib_amd")
tiles = [d for d, _ in bench.make_streams(pkg.capi, (1200000 + k for k in range(48 * 16)))]
for threads in (1, 8, 14, 16, 18, 24, 32):
    with ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(pkg.capi.parse_hevc, tiles[:threads * 2]))
        t0 = time.perf_counter()
        for _ in range(3): list(pool.map(pkg.capi.parse_hevc, tiles))
        dt = (time.perf_counter() - t0) / 3
    print(f"threads {threads:3d}: {len(tiles) * 0.262144 / dt:8.1f} MP/s  ({len(tiles) * 0.262144 / dt / threads:6.1f} per thread)", flush=True)
