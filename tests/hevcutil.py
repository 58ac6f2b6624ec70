"""Test helpers around the product's host parser (C ABI)."""
import ctypes as C


def parse(hm, data, annexb=False):
    """hm_hevc_parse -> command-stream blob as bytes (raises on error)."""
    hm.hm_hevc_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    hm.hm_free.argtypes = [C.c_void_p]
    hm.hm_last_error.restype = C.c_char_p
    blob = C.POINTER(C.c_uint8)()
    size = C.c_size_t()
    rc = hm.hm_hevc_parse(data, len(data), 1 if annexb else 0, C.byref(blob), C.byref(size))
    if rc != 0:
        raise RuntimeError(f"hm_hevc_parse failed: {rc}: {hm.hm_last_error().decode()}")
    out = C.string_at(blob, size.value)
    hm.hm_free(blob)
    return out


HM_PARSE_CONCEAL = 0x100


class _ParseOptions(C.Structure):
    _fields_ = [("annexb", C.c_int32), ("threads", C.c_int32), ("record_order", C.c_int32)]


def parse_concealing(hm, data, record_order=0):
    """hm_hevc_parse_opts with HM_PARSE_CONCEAL: damaged slice data is concealed instead of refused.
    -> (blob, concealed CTBs, raster address of the first one or -1)"""
    import struct
    hm.hm_hevc_parse_opts.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_ParseOptions), C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    hm.hm_free.argtypes = [C.c_void_p]
    hm.hm_last_error.restype = C.c_char_p
    o = _ParseOptions(0, 1, record_order | HM_PARSE_CONCEAL)
    blob = C.POINTER(C.c_uint8)()
    size = C.c_size_t()
    rc = hm.hm_hevc_parse_opts(data, len(data), C.byref(o), C.byref(blob), C.byref(size))
    if rc != 0:
        raise RuntimeError(f"hm_hevc_parse_opts failed: {rc}: {hm.hm_last_error().decode()}")
    out = C.string_at(blob, size.value)
    hm.hm_free(blob)
    n, first = struct.unpack_from("<II", out, 80)  # hm_pic.concealed_ctbs, first_concealed_ctb
    return out, n, first - 1


def split_nals(data):
    """[u32 BE length][NAL] records -> list of NAL byte strings"""
    import struct
    out, p = [], 0
    while p + 4 <= len(data):
        n = struct.unpack_from(">I", data, p)[0]
        out.append(data[p + 4:p + 4 + n])
        p += 4 + n
    return out


def join_nals(nals):
    import struct
    return b"".join(struct.pack(">I", len(n)) + n for n in nals)
