"""Wall-clock split of hm_decode_item for one 12 MP grid (HM_TRACE laps on stderr).  GPU box: python tools/e2e_trace.py"""
import sys, os, time
os.environ["HM_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, heifwriter, pipeline
import __graft_entry__ as g
pkg = g.load_package()
streams = [bench.tile_stream(0, t) for t in range(48)]
data = heifwriter.write_heic(streams, (512, 512), grid=(6, 8, 4032, 3024))
f = pipeline.HeifFile(pkg.lib(), data)
for thr in (48,):
    for _ in range(3): f.decode(f.primary(), 10, threads=thr, copy=False)
    print("---- traced calls ----", file=sys.stderr)
    f.decode(f.primary(), 10, threads=thr, copy=False)
    f.decode(f.primary(), 10, threads=thr, copy=False)
