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


OUT_BYTES = {10: 3, 11: 4, 12: 6, 13: 8, 14: 6, 15: 8}


def colour_chain(y, cb, cr, w, h, bpp, chroma, has_nclx, matrix, primaries, full_range, out_fmt, alpha=None, alpha_bits=0):
    """the float-op chains that change the sample depth or end in RRGGBBAA (oracle_colour.c: orc_ycbcr_to_rgb_chain);
    alpha: (buffer, stride) of the image's size or None"""
    o = load()
    out, os_ = alloc_plane(w, h, OUT_BYTES[out_fmt])
    o.orc_ycbcr_to_rgb_chain(ptr(y[0]), y[1], ptr(cb[0]), cb[1], ptr(cr[0]), cr[1], ptr(alpha[0]) if alpha else None,
                             alpha[1] if alpha else 0, alpha_bits, w, h, bpp, chroma, has_nclx, matrix, primaries, full_range,
                             ptr(out), os_, out_fmt)
    return out, os_


def to_hdr(plane, w, h, bits):
    """Op_to_hdr_planes on one 8-bit plane -> `bits`-bit (buffer, stride) in 16-bit storage"""
    o = load()
    out, os_ = alloc_plane(w, h, 2)
    o.orc_to_hdr_plane(ptr(plane[0]), plane[1], w, h, bits, ptr(out), os_)
    return out, os_


def to_sdr(plane, w, h, bits):
    """Op_to_sdr_planes on one plane of `bits` > 8 -> 8-bit (buffer, stride)"""
    o = load()
    out, os_ = alloc_plane(w, h, 1)
    o.orc_to_sdr_plane(ptr(plane[0]), plane[1], w, h, bits, ptr(out), os_)
    return out, os_


def upsample_bilinear(plane, w, h, bpp, chroma):
    """Op_YCbCr420/422_bilinear_to_YCbCr444 on one chroma plane (buffer, stride) -> (buffer, stride) of size w x h."""
    o = load()
    bps = 2 if bpp > 8 else 1
    out, os_ = alloc_plane(w, h, bps)
    fn = {(1, 1): o.orc_upsample_bilinear_420, (1, 2): o.orc_upsample_bilinear_420_u16,
          (2, 1): o.orc_upsample_bilinear_422, (2, 2): o.orc_upsample_bilinear_422_u16}[(chroma, bps)]
    fn(ptr(plane[0]), plane[1] // bps, w, h, ptr(out), os_ // bps)
    return out, os_


def transform_planes(planes, dims, img_w, img_h, bd, transforms):
    """Apply ('irot', q) / ('imir', axis) / ('clap', 8-tuple) to [(buf, stride)] planes of sizes dims = [(w, h)] like
    context.cc:1957-2020 does.  Returns (planes, dims, img_w, img_h)."""
    o = load()
    bps = 2 if bd > 8 else 1
    planes, dims = list(planes), list(dims)
    for kind, v in transforms:
        if kind == "irot":
            ang = (v & 3) * 90
            if ang == 0:
                continue
            for c in range(3):
                w, h = dims[c]
                nw, nh = (w, h) if ang == 180 else (h, w)
                out, os_ = alloc_plane(nw, nh, bps)
                o.orc_rotate_ccw_plane(ptr(planes[c][0]), planes[c][1], w, h, bps, ang, ptr(out), os_)
                planes[c], dims[c] = (out, os_), (nw, nh)
            if ang != 180:
                img_w, img_h = img_h, img_w
        elif kind == "imir":
            assert bd == 8
            for c in range(3):
                buf = planes[c][0].copy()
                o.orc_mirror_plane(ptr(buf), planes[c][1], dims[c][0], dims[c][1], v & 1)
                planes[c] = (buf, planes[c][1])
        else:
            clap = (C.c_int64 * 8)(*v)
            rect = (C.c_int * 4)()
            rc = o.orc_clap_rect(clap, img_w, img_h, rect)
            if rc:
                raise ValueError(f"clap invalid ({rc})")
            for c in range(3):
                wh = (C.c_int * 2)()
                o.orc_crop_plane(ptr(planes[c][0]), planes[c][1], dims[c][0], dims[c][1], bps, img_w, img_h, rect, None, 0, wh)
                out, os_ = alloc_plane(wh[0], wh[1], bps)
                o.orc_crop_plane(ptr(planes[c][0]), planes[c][1], dims[c][0], dims[c][1], bps, img_w, img_h, rect, ptr(out), os_, wh)
                planes[c], dims[c] = (out, os_), (wh[0], wh[1])
            img_w, img_h = rect[1] - rect[0] + 1, rect[3] - rect[2] + 1
    return planes, dims, img_w, img_h


def fnv_rows(buf, stride, row_bytes, rows):
    return load().orc_fnv1a64_rows(ptr(buf), stride, row_bytes, rows, 0)


# ---- HEVC: reference decoder (oracle/_ref) and oracle executors -------------------------------

class RefPicture(C.Structure):
    _fields_ = [("width", C.c_int * 3), ("height", C.c_int * 3), ("bit_depth", C.c_int * 3),
                ("chroma", C.c_int), ("full_range", C.c_int), ("primaries", C.c_int),
                ("transfer", C.c_int), ("matrix", C.c_int),
                ("plane_bytes", C.c_size_t * 3), ("plane", C.c_void_p * 3)]


REF_F_ANNEXB, REF_F_NO_DEBLOCK, REF_F_NO_SAO, REF_F_SCALAR = 1, 2, 4, 8
_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def load_ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
        _ref.ref_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(RefPicture)]
    return _ref


def ref_decode(data, flags=0, threads=0):
    """Decode with the real reference libde265; returns (planes[3] as uint16 arrays, info dict)."""
    R = load_ref()
    pic = RefPicture()
    rc = R.ref_decode(data, len(data), flags, threads, C.byref(pic))
    if rc != 0:
        raise RuntimeError(f"reference decoder failed: {rc}")
    planes = []
    for c in range(1 if pic.chroma == 0 else 3):  # a monochrome picture has a luma plane only
        w, h, bd = pic.width[c], pic.height[c], pic.bit_depth[c]
        n = pic.plane_bytes[c]
        raw = np.frombuffer(C.string_at(pic.plane[c], n), dtype=np.uint8)
        a = raw.astype(np.uint16).reshape(h, w) if bd <= 8 else raw.view(np.uint16).reshape(h, w).copy()
        planes.append(a)
    info = dict(chroma=pic.chroma, bit_depth=pic.bit_depth[0], full_range=pic.full_range,
                primaries=pic.primaries, transfer=pic.transfer, matrix=pic.matrix)
    R.ref_free_picture(C.byref(pic))
    return planes, info


def conformance_window(blob):
    """(left, right, top, bottom) of hm_pic in luma samples"""
    import struct
    return struct.unpack_from("<4H", blob, 12)


def oracle_decode(blob, stages=3, crop=False):
    """Run the oracle's scalar executors on a command-stream blob (bytes).  crop=True: the conformance window only,
    i.e. the planes a decoder plugin hands out (de265_get_image_plane / width / height)."""
    o = load()
    info = (C.c_int * 8)()
    buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
    if o.orc_stream_info(buf, len(blob), info) != 0:
        raise RuntimeError("bad command stream")
    w, h, cf = info[0], info[1], info[2]
    cw, ch = (w if cf == 3 else w // 2), (h // 2 if cf == 1 else h)
    y = np.zeros((h, w), np.uint16)
    cb = np.zeros((ch, cw), np.uint16)
    cr = np.zeros((ch, cw), np.uint16)
    rc = o.orc_decode_picture(buf, len(blob), stages, ptr(y), ptr(cb), ptr(cr))
    if rc != 0:
        raise RuntimeError(f"oracle decode failed: {rc}")
    if crop:
        cl, cr_, ct, cb_ = conformance_window(blob)
        sw, sh = (1 if cf == 3 else 2), (2 if cf == 1 else 1)
        y = np.ascontiguousarray(y[ct:h - cb_, cl:w - cr_])
        cb = np.ascontiguousarray(cb[ct // sh:(h - cb_) // sh, cl // sw:(w - cr_) // sw])
        cr = np.ascontiguousarray(cr[ct // sh:(h - cb_) // sh, cl // sw:(w - cr_) // sw])
        w, h = w - cl - cr_, h - ct - cb_
    return ([y] if cf == 0 else [y, cb, cr]), dict(width=w, height=h, chroma=cf, bit_depth=info[3], full_range=info[4],
                             matrix=info[5], primaries=info[6], has_vui_colour=info[7])


def convert_by_search(planes, w, h, bpp, chroma, nclx, out_fmt, has_alpha=False, forced_bilinear=False):
    """The reference's conversion of Y / Cb / Cr planes [(buffer, stride)] to an interleaved target, op by op along the
    chain its pipeline search picks (oracle/pipeline_search.py), every op by its oracle restatement.  nclx = (has_nclx,
    matrix, primaries, full_range) of the image.  The alpha plane itself is not part of this function (it does not
    enter the colour values); has_alpha only steers the search.  Returns (buffer, stride, chain)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import pipeline_search as ps
    n = ps.Nclx(nclx[1], nclx[2], 2, bool(nclx[3])) if nclx[0] else None
    opts = ps.Options(ps.DOWN_AVERAGE, ps.UP_BILINEAR, forced_bilinear)
    inp, tgt = ps.conversion_states(ps.CS_MONO if chroma == 0 else ps.CS_YCBCR, chroma, has_alpha, bpp, n, ps.CS_RGB, out_fmt)
    steps = ps.construct_pipeline(inp, tgt, opts)
    assert steps is not None, "the reference finds no chain"
    chain = [name for name, _ in steps]
    cw, ch = ((w + 1) // 2 if chroma != 3 else w), ((h + 1) // 2 if chroma == 1 else h)
    y, cb, cr = planes
    bits, cur_chroma = bpp, chroma
    seen = tuple(nclx)  # the profile the next op reads off its input image: the image's own for the first op
    # ... afterwards the state's: undefined values replaced by the sRGB defaults (colorconversion.cc:452-455, 520-527)
    m2, p2 = (nclx[1], nclx[2]) if nclx[0] else (2, 2)
    state_profile = (1, 6 if m2 == 2 else m2, 1 if p2 == 2 else p2, nclx[3] if nclx[0] else 1)
    for k, name in enumerate(chain):
        if name == "Op_drop_alpha_plane":
            pass
        elif name == "Op_mono_to_YCbCr420":  # monochrome.cc:26-155: neutral chroma planes at 4:2:0 size
            cur_chroma, cw, ch = 1, (w + 1) // 2, (h + 1) // 2
            bps = 2 if bits > 8 else 1
            cb = alloc_plane(cw, ch, bps)
            if bps == 1:
                cb[0][:] = 128
            else:
                cb[0].view(np.uint16)[:] = 128 << (bits - 8)
            cr = cb
            # the op's output state is a fresh ColorState: its profile is the sRGB default set (nclx.h:124), and that is
            # what every later op reads off its input image (colorconversion.cc:452-455)
            state_profile = (1, 6, 1, 1)
        elif name == "Op_to_hdr_planes":
            y, cb, cr = to_hdr(y, w, h, tgt.bpp), to_hdr(cb, cw, ch, tgt.bpp), to_hdr(cr, cw, ch, tgt.bpp)
            bits = tgt.bpp
        elif name == "Op_to_sdr_planes":
            y, cb, cr = to_sdr(y, w, h, bits), to_sdr(cb, cw, ch, bits), to_sdr(cr, cw, ch, bits)
            bits = 8
        elif "bilinear_to_YCbCr444" in name:
            cb, cr = upsample_bilinear(cb, w, h, bits, cur_chroma), upsample_bilinear(cr, w, h, bits, cur_chroma)
            cur_chroma, cw, ch = 3, w, h
        elif name in ("Op_YCbCr420_to_RGB24", "Op_YCbCr420_to_RGB32"):
            assert k == len(chain) - 1
            out, os_ = colour_int(y, cb, cr, w, h, seen[0], seen[1], seen[2], out_fmt)
            return out, os_, chain
        elif name.startswith("Op_YCbCr_to_RGB<") or name == "Op_YCbCr420_to_RRGGBBaa":
            rest = chain[k + 1:]
            assert all(r in ("Op_to_hdr_planes", "Op_to_sdr_planes", "Op_RGB_to_RGB24_32", "Op_RGB_HDR_to_RRGGBBaa_BE",
                             "Op_RRGGBBaa_swap_endianness") for r in rest), rest
            out, os_ = colour_chain(y, cb, cr, w, h, bits, cur_chroma, *seen, out_fmt)
            return out, os_, chain
        else:
            raise AssertionError("op outside the decode path: " + name)
        seen = state_profile
    raise AssertionError("chain without a YCbCr -> RGB op")
