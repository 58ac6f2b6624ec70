"""Helpers: run the HIP tile-decode path (through the C ABI) on command streams."""
import ctypes as C

import numpy as np


def decode_pictures(pkg, blobs, stages=3, dests=None):
    """Decode a batch of command streams on cuda:0; every picture gets its own canvas of the size of its conformance
    window (what a decoder plugin hands out).  Returns list of [Y, Cb, Cr] uint16 arrays."""
    import torch
    capi = pkg.capi
    dev = torch.device("cuda:0")
    batch = capi.Batch()
    outs = []
    keep = []
    for blob in blobs:
        h = capi.stream_header(blob)
        cl, cr, ct, cb = h["crop"]
        w, hh, cf, bd = h["width"] - cl - cr, h["height"] - ct - cb, h["chroma_format"], h["bit_depth"]
        bps = 2 if bd > 8 else 1
        cw, ch = (w if cf == 3 else w // 2), (hh // 2 if cf == 1 else hh)
        planes = []
        for (pw, ph) in (((w, hh),) if cf == 0 else ((w, hh), (cw, ch), (cw, ch))):  # monochrome: luma only
            pitch = (pw * bps + 63) // 64 * 64
            planes.append((torch.zeros((ph, pitch), dtype=torch.uint8, device=dev), pitch, pw, ph))
        d = capi.TileDest()
        for c in range(len(planes)):
            d.plane[c] = planes[c][0].data_ptr()
            d.pitch[c] = planes[c][1]
        d.canvas_width, d.canvas_height, d.x0, d.y0 = w, hh, 0, 0
        d.tile_has_nclx = 0
        batch.add(blob, d)
        keep.append(planes)
        outs.append((planes, bps))
    st = torch.cuda.current_stream().cuda_stream
    batch.upload(st)
    batch.execute(stages, st)
    torch.cuda.synchronize()
    batch.check()
    res = []
    for planes, bps in outs:
        pic = []
        for (t, pitch, pw, ph) in planes:
            a = t.cpu().numpy()
            if bps == 1:
                pic.append(a[:, :pw].astype(np.uint16))
            else:
                pic.append(a[:, :pw * 2].copy().view(np.uint16).reshape(ph, pw))
        res.append(pic)
    batch.close()
    return res
