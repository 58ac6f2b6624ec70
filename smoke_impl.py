"""smoke(): one small invocation of the hot path on cuda:0, checked against the oracle.

A 2x1 grid of 128x128 synthetic tiles (one with VUI full-range, one without VUI so the paste
rescale quirk is exercised) is entropy-decoded on the host (product parser), reconstructed /
deblocked / SAO-filtered / pasted and converted to RGB24 by the HIP kernels through the C ABI,
and compared bit-for-bit with the CPU oracle (oracle/liboracle.so)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run():
    import numpy as np
    import torch
    import __graft_entry__ as g
    import orc
    import synthutil

    pkg = g.load_package()
    capi, L = pkg.capi, pkg.lib()
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    assert L.hm_device_count() >= 1
    dev = torch.device("cuda:0")
    W, H, T = 250, 120, 128  # canvas smaller than the 2x1 tile grid: right/bottom tiles are cropped
    tiles = [synthutil.picture(4242, width=T, height=T, vui=1, full_range=1, matrix=6),
             synthutil.picture(4243, width=T, height=T, vui=0)]
    blobs = [capi.parse_hevc(t) for t in tiles]
    ys, cs, os_ = L.hm_plane_stride(W, 1), L.hm_plane_stride((W + 1) // 2, 1), L.hm_plane_stride(W, 3)
    rows = max(64, H + 1)
    y = torch.zeros((rows, ys), dtype=torch.uint8, device=dev)
    cb = torch.zeros((rows, cs), dtype=torch.uint8, device=dev)
    cr = torch.zeros((rows, cs), dtype=torch.uint8, device=dev)
    rgb = torch.zeros((rows, os_), dtype=torch.uint8, device=dev)
    batch = capi.Batch()
    nclx = []
    for i, blob in enumerate(blobs):
        h = capi.stream_header(blob)
        d = capi.TileDest()
        d.plane[0], d.plane[1], d.plane[2] = y.data_ptr(), cb.data_ptr(), cr.data_ptr()
        d.pitch[0], d.pitch[1], d.pitch[2] = ys, cs, cs
        d.canvas_width, d.canvas_height, d.x0, d.y0 = W, H, i * T, 0
        # the libde265 plugin always attaches an nclx built from the VUI (defaults 2,2,2,full=0)
        d.tile_has_nclx, d.tile_full_range, d.tile_matrix = 1, h["full_range"], h["matrix"]
        nclx.append((1, h["full_range"], h["matrix"]))
        batch.add(blob, d)
    st = torch.cuda.current_stream().cuda_stream
    batch.upload(st)
    batch.execute(3, st)
    desc = capi.ColourDesc(W, H, 8, 1, 0, 0, 0, 0, capi.HM_OUT_RGB, ys, cs, cs, os_)
    capi.check(L.hm_colour_convert(C.byref(desc), y.data_ptr(), cb.data_ptr(), cr.data_ptr(), rgb.data_ptr(), st))
    torch.cuda.synchronize()
    got = rgb.cpu().numpy()

    # ---- oracle ----
    o = orc.load()
    oy, ocb, ocr = (np.zeros((rows, s), np.uint8) for s in (ys, cs, cs))
    for i, blob in enumerate(blobs):
        planes, _ = orc.oracle_decode(blob, 3)
        for c, (canvas, stride) in enumerate(((oy, ys), (ocb, cs), (ocr, cs))):
            p8 = np.ascontiguousarray(planes[c].astype(np.uint8))
            assert o.orc_paste_tile_plane(orc.ptr(p8), p8.shape[1], p8.shape[1], p8.shape[0], orc.ptr(canvas), stride,
                                          W, H, i * T, 0, c, 1, 8, *nclx[i]) == 0
    exp = np.zeros((rows, os_), np.uint8)
    o.orc_ycbcr420_to_rgb_int(orc.ptr(oy), ys, orc.ptr(ocb), cs, orc.ptr(ocr), cs, W, H, 0, 0, 0, orc.ptr(exp), os_, 10)
    assert np.array_equal(y.cpu().numpy()[:H, :W], oy[:H, :W]), "canvas Y mismatch"
    assert np.array_equal(got[:H, :W * 3], exp[:H, :W * 3]), "RGB mismatch vs oracle"
    print(f"smoke ok: {len(tiles)} tiles -> {W}x{H} RGB24 bit-exact vs oracle; {L.hm_version().decode()}")


if __name__ == "__main__":
    run()
