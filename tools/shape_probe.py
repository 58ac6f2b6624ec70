"""few large pictures: k_recon_quad vs k_recon by class and picture shape (HM_QUAD_CLASS=1 / 0)"""
import importlib, os, sys, json
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synthutil
pkg = importlib.import_module("heif-decoder-lib_amd")
capi, L = pkg.capi, pkg.lib()
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
out = {}
for name, (w, h, n, kw) in {
    "10b422_2048x1536_ctb32_x32": (2048, 1536, 32, dict(chroma_format=2, bit_depth=10, log2_ctb=5)),
    "10b422_2048x1536_ctb32_x4": (2048, 1536, 4, dict(chroma_format=2, bit_depth=10, log2_ctb=5)),
    "10b420_1920x1080_ctb64_x8": (1920, 1080, 8, dict(bit_depth=10, log2_ctb=6)),
    "10b420_1024x1024_ctb32_x64": (1024, 1024, 64, dict(bit_depth=10, log2_ctb=5)),
    "8b420_1920x1080_ctb16_x8": (1920, 1080, 8, dict(log2_ctb=4)),
    "8b420_1920x1080_ctb64_x8": (1920, 1080, 8, dict(log2_ctb=6)),
    "8b420_1920x1080_ctb64_x1": (1920, 1080, 1, dict(log2_ctb=6)),
    "8b420_4032x3024_ctb32_x1": (4032, 3024, 1, dict(log2_ctb=5)),
    "8b420_4096x2304_ctb32_x4": (4096, 2304, 4, dict(log2_ctb=5)),
}.items():
    blob = capi.parse_hevc(synthutil.picture(4220010, width=w, height=h, qp=30, **kw))
    bps = 2 if kw.get("bit_depth", 8) > 8 else 1
    cf = kw.get("chroma_format", 1)
    ys, cs = L.hm_plane_stride(w, bps), L.hm_plane_stride(w // 2, bps)
    ch = h // 2 if cf == 1 else h
    batch = capi.Batch(); keep = []
    for _ in range(n):
        y = torch.zeros((h, ys), dtype=torch.uint8, device=dev); cb = torch.zeros((ch, cs), dtype=torch.uint8, device=dev); cr = torch.zeros((ch, cs), dtype=torch.uint8, device=dev)
        keep.append((y, cb, cr))
        d = capi.TileDest()
        d.plane[0], d.plane[1], d.plane[2] = y.data_ptr(), cb.data_ptr(), cr.data_ptr()
        d.pitch[0], d.pitch[1], d.pitch[2] = ys, cs, cs
        d.canvas_width, d.canvas_height = w, h
        batch.add(blob, d)
    batch.upload(st); batch.execute(3, st); torch.cuda.synchronize()
    batch.set_profiling(3)
    for _ in range(3): batch.execute(3, st)
    torch.cuda.synchronize()
    out[name] = round(sum(batch.timings_ms(s)[0] for s in range(3)) / 3, 3)
    batch.close()
print(os.environ.get("HM_QUAD_CLASS"), os.environ.get("HM_QUAD_WAVES"), json.dumps(out))
