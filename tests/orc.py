"""ctypes bindings of the test oracle (oracle/liboracle.so, oracle/_ref/libde265_ref.so).
Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libde265_ref.so")

_o = None


def load():
    global _o
    if _o is None:
        if not os.path.exists(ORACLE_SO):
            import subprocess
            subprocess.run(["make", "port"], cwd=os.path.join(ROOT, "oracle"), check=True)
        _o = C.CDLL(ORACLE_SO)
        _o.orc_fnv1a64.restype = C.c_uint64
        _o.orc_fnv1a64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        _o.orc_fnv1a64_rows.restype = C.c_uint64
        _o.orc_fnv1a64_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64]
    return _o


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def plane_stride(width, bpp):
    return load().orc_plane_stride(width, bpp)


def alloc_plane(width, height, bytes_per_px=1, fill=None, rng=None, maxval=255):
    """A libheif-style plane: stride from pixelimage.cc rules, rows padded."""
    stride = plane_stride(width, bytes_per_px)
    rows = max(64, (height + 1) & ~1)
    buf = np.zeros((rows, stride), np.uint8)
    if rng is not None:
        if bytes_per_px == 1:
            buf[:height, :width] = rng.integers(0, maxval + 1, (height, width), dtype=np.uint8)
        else:
            v = rng.integers(0, maxval + 1, (height, width), dtype=np.uint16)
            buf[:height, :width * 2] = v.view(np.uint8).reshape(height, width * 2)
    elif fill is not None:
        buf[:] = fill
    return buf, stride


def colour_int(y, cb, cr, w, h, has_nclx, matrix, primaries, out_fmt):
    o = load()
    bpp = 4 if out_fmt == 11 else 3
    out, os_ = alloc_plane(w, h, bpp)
    o.orc_ycbcr420_to_rgb_int(ptr(y[0]), y[1], ptr(cb[0]), cb[1], ptr(cr[0]), cr[1], w, h,
                              has_nclx, matrix, primaries, ptr(out), os_, out_fmt)
    return out, os_


def colour_float(y, cb, cr, w, h, bpp, chroma, has_nclx, matrix, primaries, full_range, out_fmt):
    o = load()
    obpp = {10: 3, 11: 4, 12: 6, 14: 6}[out_fmt]
    out, os_ = alloc_plane(w, h, obpp)
    o.orc_ycbcr_to_rgb_float(ptr(y[0]), y[1], ptr(cb[0]), cb[1], ptr(cr[0]), cr[1], w, h, bpp, chroma,
                             has_nclx, matrix, primaries, full_range, ptr(out), os_, out_fmt)
    return out, os_


def fnv_rows(buf, stride, row_bytes, rows):
    return load().orc_fnv1a64_rows(ptr(buf), stride, row_bytes, rows, 0)
