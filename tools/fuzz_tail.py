"""differential fuzz of the fused tails: corrupted-but-parsable tiles as 2x2 grids - 512x512 8-bit 4:2:0 through k_tail420, 256x256
10-bit / 4:2:2 tiles through k_tailf - against the separate kernels (k_deblock, k_sao_paste, colour kernel) on the same batch"""
import importlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, corpus, hevcutil, synthutil
pkg = importlib.import_module("heif-decoder-lib_amd")
hm = pkg.lib()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
blobs = []
tries = 0
while len(blobs) < 64 and tries < 20000:
    tries += 1
    data = synthutil.picture(1200000 + tries % 8, vui=1, full_range=1, matrix=6, **corpus.TILE)
    b = bytearray(data)
    for _ in range(rng.randrange(1, 5)):
        b[rng.randrange(len(b) // 4, len(b))] ^= 1 << rng.randrange(8)
    try:
        blobs.append(hevcutil.parse(hm, bytes(b)))
    except RuntimeError:
        pass
print(len(blobs), "corrupted tiles parse after", tries, "tries", flush=True)
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
out = []
n_img = len(blobs) // 4
for group in (0, -1):
    gb = bench.GridBatch(pkg, dev, 2, 2, 512, 1000, 1010)
    for j in range(n_img):
        gb.add_image(blobs[4 * j:4 * j + 4])
    gb.finish(st, group)
    gb.batch.execute(3, st)
    torch.cuda.synchronize()
    assert gb.batch.tail_fused() == (group == 0)
    out.append([im["rgb"].cpu().numpy()[:1010, :3000].copy() for im in gb.images])
    gb.batch.close()
bad = sum(not np.array_equal(a, b) for a, b in zip(*out))
print("images that differ between the fused tail and the separate kernels:", bad, "of", n_img)

# ---- the same for the float-chain classes (k_tailf): corrupted 256x256 tiles as 2x2 grids -------------------------------------
import ctypes as C
capi, L = pkg.capi, pkg.lib()
# (r06: the hdr_* classes - deep full-range 4:2:0 to RGB24 / RGBA32 - run on k_tail420's 16-bit instantiation; 7th element: log2 of the CTB size)
for name, (bd, cf, full, matrix, fmt, obpp, l2) in {"422_10_rrggbb_le": (10, 2, 0, 9, "HM_OUT_RRGGBB_LE", 6, 5), "420_10_rgb24": (10, 1, 0, 6, "HM_OUT_RGB", 3, 5),
                                                      "422_8_rgba": (8, 2, 1, 6, "HM_OUT_RGBA", 4, 5), "hdr_420_10_full_rgb24": (10, 1, 1, 9, "HM_OUT_RGB", 3, 5),
                                                      "hdr_420_10_ctb16_rgba": (10, 1, 1, 6, "HM_OUT_RGBA", 4, 4), "hdr_420_11_ctb64_rgb24": (11, 1, 1, 1, "HM_OUT_RGB", 3, 6)}.items():
    fb, tries = [], 0
    while len(fb) < 32 and tries < 20000:
        tries += 1
        data = synthutil.picture(8300000 + tries % 8, width=256, height=256, chroma_format=cf, bit_depth=bd, log2_ctb=l2, qp=27, vui=1, full_range=full, matrix=matrix)
        b = bytearray(data)
        for _ in range(rng.randrange(1, 5)):
            b[rng.randrange(len(b) // 4, len(b))] ^= 1 << rng.randrange(8)
        try:
            fb.append(hevcutil.parse(hm, bytes(b)))
        except RuntimeError:
            pass
    w, h, bps = 500, 505, (2 if bd > 8 else 1)
    ys, cs, os_ = L.hm_plane_stride(w, bps), L.hm_plane_stride((w + 1) // 2, bps), L.hm_plane_stride(w, obpp)
    ch = (h + 1) // 2 if cf == 1 else h
    n_img = len(fb) // 4
    res = []
    for group in (0, -1):
        batch, ims = capi.Batch(), []
        for j in range(n_img):
            im = [torch.zeros((max(64, r), s), dtype=torch.uint8, device=dev) for r, s in ((h, ys), (ch, cs), (ch, cs), (h, os_))]
            for t in range(4):
                d = capi.TileDest()
                d.plane[0], d.plane[1], d.plane[2] = im[0].data_ptr(), im[1].data_ptr(), im[2].data_ptr()
                d.pitch[0], d.pitch[1], d.pitch[2] = ys, cs, cs
                d.canvas_width, d.canvas_height, d.x0, d.y0 = w, h, (t % 2) * 256, (t // 2) * 256
                batch.add(fb[4 * j + t], d)
            ims.append(im)
        batch.upload(st)
        PtrArr = C.c_void_p * n_img
        ptrs = [PtrArr(*[im[k].data_ptr() for im in ims]) for k in range(4)]
        batch.set_colour(capi.ColourDesc(w, h, bd, cf, 1, matrix, 1, full, getattr(capi, fmt), ys, cs, cs, os_), n_img, *ptrs, group)
        batch.execute(3, st)
        torch.cuda.synchronize()
        assert batch.tail_fused() == (group == 0), name
        res.append([im[3].cpu().numpy()[:h, :w * obpp].copy() for im in ims])
        batch.close()
    print(name + ": images that differ between the fused float tail and the separate kernels:", sum(not np.array_equal(a, b) for a, b in zip(*res)), "of", n_img, flush=True)
