"""Micro-benchmark of the fused YCbCr->RGB kernels (device-resident in/out).
Prints achieved algorithmic GB/s (read Y+Cb+Cr once, write interleaved once) per config."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
L = pkg.lib()
capi = pkg.capi


def run(w, h, bit_depth, chroma, nclx, out_fmt, iters=50):
    dev = torch.device("cuda:0")
    bps = 2 if bit_depth > 8 else 1
    cw = w if chroma == 3 else (w + 1) // 2
    ch = (h + 1) // 2 if chroma == 1 else h
    ys, cs = L.hm_plane_stride(w, bps), L.hm_plane_stride(cw, bps)
    obpp = L.hm_out_bytes_per_pixel(out_fmt)
    os_ = L.hm_plane_stride(w, obpp)
    mk = lambda rows, stride: torch.randint(0, 256 if bps == 1 else 4, (max(64, rows + 1), stride), dtype=torch.uint8, device=dev)
    y, cb, cr = mk(h, ys), mk(ch, cs), mk(ch, cs)
    out = torch.empty((max(64, h + 1), os_), dtype=torch.uint8, device=dev)
    d = capi.ColourDesc(w, h, bit_depth, chroma, *nclx, out_fmt, ys, cs, cs, os_)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: capi.check(L.hm_colour_convert(C.byref(d), y.data_ptr(), cb.data_ptr(), cr.data_ptr(), out.data_ptr(), st))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    alg = (w * h + 2 * cw * ch) * bps + w * h * obpp
    return {"w": w, "h": h, "bit_depth": bit_depth, "chroma": chroma, "out_fmt": out_fmt, "pipe": L.hm_colour_pipeline(C.byref(d)),
            "us": round(ms * 1e3, 2), "GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4),
            "MPps": round(w * h / ms / 1e3, 1)}


if __name__ == "__main__":
    for cfg in [
        (4032, 3024, 8, 1, (0, 0, 0, 0), 10),
        (16384, 16384, 8, 1, (0, 0, 0, 0), 10),
        (16384, 16384, 8, 1, (0, 0, 0, 0), 11),
        (16384, 16384, 8, 1, (1, 2, 2, 0), 10),
        (2048, 1536, 10, 2, (1, 9, 9, 0), 14),
        (16384, 8192, 10, 2, (1, 9, 9, 0), 14),
    ]:
        print(json.dumps(run(*cfg)), flush=True)
