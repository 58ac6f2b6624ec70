"""ctypes bindings of include/heif_mi355x.h.  No CPU fallback: if the shared library
is missing, loading raises; if there is no GPU the device entry points return
HM_ERR_NO_DEVICE and HmError is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libheif_mi355x.so")

HM_CHROMA_420, HM_CHROMA_422, HM_CHROMA_444 = 1, 2, 3
HM_OUT_RGB, HM_OUT_RGBA, HM_OUT_RRGGBB_BE, HM_OUT_RRGGBB_LE = 10, 11, 12, 14
HM_PIPE_INT420, HM_PIPE_FLOAT = 1, 2


class HmError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__(f"heif_mi355x status {status}: {detail}")
        self.status = status


class ColourDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "bit_depth", "chroma", "has_nclx", "matrix", "primaries",
        "full_range", "out_format", "y_stride", "cb_stride", "cr_stride", "out_stride")]


_lib = None


def lib():
    """Load libheif_mi355x.so (built in-tree by __graft_entry__.build()); fail loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} not built - run `python -c 'import __graft_entry__ as g; g.build()'`")
        # torch ships its own libamdhip64; it must be the one HIP runtime of the process, so
        # load it before our library binds libamdhip64 (two runtimes => "no ROCm-capable device")
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.hm_status_string.restype = C.c_char_p
        L.hm_last_error.restype = C.c_char_p
        L.hm_version.restype = C.c_char_p
        L.hm_colour_convert.argtypes = [C.POINTER(ColourDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hm_colour_pipeline.argtypes = [C.POINTER(ColourDesc)]
        L.hm_ycbcr_coefficients.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        _lib = L
    return _lib


def check(status):
    if status < 0:
        L = lib()
        raise HmError(status, f"{L.hm_status_string(status).decode()}: {L.hm_last_error().decode()}")
    return status
