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
