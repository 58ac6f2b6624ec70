"""ctypes bindings of include/heif_mi355x.h.  No CPU fallback: if the shared library
is missing, loading raises; if there is no GPU the device entry points return
HM_ERR_NO_DEVICE and HmError is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libheif_mi355x.so")
# the same objects + csrc/test_hooks.cpp (hm_debug_set, hm_debug_kernel_regs): tests and measurement scripts that force a cut of the
# chain kernel, inject a fault or read a kernel's registers load THIS one (use_test_hooks() before the first lib()); the shipping
# library exports no such entry point
TEST_LIB_PATH = os.path.join(_HERE, "libheif_mi355x_test.so")

HM_CHROMA_420, HM_CHROMA_422, HM_CHROMA_444 = 1, 2, 3
HM_OUT_RGB, HM_OUT_RGBA, HM_OUT_RRGGBB_BE, HM_OUT_RRGGBB_LE = 10, 11, 12, 14
HM_OUT_RRGGBBAA_BE, HM_OUT_RRGGBBAA_LE = 13, 15
HM_PIPE_INT420, HM_PIPE_FLOAT, HM_PIPE_BILINEAR_FLOAT, HM_PIPE_TO_HDR_FLOAT, HM_PIPE_MONO = 1, 2, 3, 4, 5
HM_PIPE_SDR_INT420, HM_PIPE_FLOAT_SDR, HM_PIPE_FLOAT_HDR = 6, 7, 8


class HmError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__(f"heif_mi355x status {status}: {detail}")
        self.status = status


class ColourDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "bit_depth", "chroma", "has_nclx", "matrix", "primaries",
        "full_range", "out_format", "y_stride", "cb_stride", "cr_stride", "out_stride", "chroma_upsampling", "has_alpha")]


_lib = None


def use_test_hooks():
    """Make lib() load libheif_mi355x_test.so (test infrastructure: tests/knobs.py, tools/).  Must come before the first lib()."""
    global LIB_PATH
    if _lib is not None and LIB_PATH != TEST_LIB_PATH:
        raise RuntimeError("the shipping library is already loaded in this process: call use_test_hooks() first")
    LIB_PATH = TEST_LIB_PATH


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
        bind_decode(L)
        _lib = L
    return _lib


def check(status):
    if status < 0:
        L = lib()
        raise HmError(status, f"{L.hm_status_string(status).decode()}: {L.hm_last_error().decode()}")
    return status


class TileDest(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("pitch", C.c_int32 * 3),
                ("canvas_width", C.c_int32), ("canvas_height", C.c_int32),
                ("x0", C.c_int32), ("y0", C.c_int32),
                ("tile_has_nclx", C.c_int32), ("tile_full_range", C.c_int32), ("tile_matrix", C.c_int32)]


def bind_decode(L):
    """argtypes of the parse / batch entry points (called once from lib())."""
    L.hm_hevc_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.hm_free.argtypes = [C.c_void_p]
    L.hm_batch_create.argtypes = [C.POINTER(C.c_void_p)]
    L.hm_batch_destroy.argtypes = [C.c_void_p]
    L.hm_batch_clear.argtypes = [C.c_void_p]
    L.hm_batch_add.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(TileDest)]
    L.hm_batch_size.argtypes = [C.c_void_p]
    L.hm_batch_upload.argtypes = [C.c_void_p, C.c_void_p]
    L.hm_batch_execute.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.hm_batch_upload_execute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.hm_batch_tail_fused.argtypes = [C.c_void_p]
    L.hm_batch_tail_fused.restype = C.c_int
    L.hm_batch_set_colour.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.hm_batch_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.hm_batch_set_concurrency.argtypes = [C.c_void_p, C.c_int]
    L.hm_batch_get_timings.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.hm_batch_get_timings4.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.hm_batch_get_timings5.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.hm_batch_algorithmic_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]


class ParseOptions(C.Structure):
    """hm_parse_options (include/heif_mi355x.h)"""
    _fields_ = [("annexb", C.c_int32), ("threads", C.c_int32), ("record_order", C.c_int32)]


def parse_hevc(data, annexb=False, threads=1, record_order=None):
    """hm_hevc_parse[_mt] - or, with a record order (HM_RECORDS_*), hm_hevc_parse_opts - -> command-stream blob (bytes)."""
    L = lib()
    blob = C.POINTER(C.c_uint8)()
    size = C.c_size_t()
    L.hm_hevc_parse_mt.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.hm_hevc_parse_mt.restype = C.c_int
    if record_order is not None:
        L.hm_hevc_parse_opts.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(ParseOptions), C.c_void_p, C.c_void_p]
        L.hm_hevc_parse_opts.restype = C.c_int
        o = ParseOptions(1 if annexb else 0, threads, record_order)
        check(L.hm_hevc_parse_opts(data, len(data), C.byref(o), C.byref(blob), C.byref(size)))
    elif threads > 1:
        check(L.hm_hevc_parse_mt(data, len(data), 1 if annexb else 0, threads, C.byref(blob), C.byref(size)))
    else:
        check(L.hm_hevc_parse(data, len(data), 1 if annexb else 0, C.byref(blob), C.byref(size)))
    out = C.string_at(blob, size.value)
    L.hm_free(blob)
    return out


def stream_header(blob):
    """(width, height, chroma_format, bit_depth, flags, full_range, matrix, primaries, has_vui_colour) of a command stream."""
    import struct
    magic, total, w, h = struct.unpack_from("<IIHH", blob, 0)
    cl, cr, ct, cb = struct.unpack_from("<4H", blob, 12)
    cf, bdy, bdc, l2ctb = struct.unpack_from("<BBBB", blob, 20)
    flags, = struct.unpack_from("<I", blob, 36)
    prim, trc, mat, fr = struct.unpack_from("<BBBB", blob, 40)
    return dict(width=w, height=h, crop=(cl, cr, ct, cb), chroma_format=cf, bit_depth=bdy, log2_ctb=l2ctb, flags=flags,
                primaries=prim, transfer=trc, matrix=mat, full_range=fr, has_vui_colour=bool(flags & 0x10))


class Batch:
    """Thin RAII wrapper of hm_batch."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p()
        check(self.L.hm_batch_create(C.byref(self.h)))

    def add(self, blob, dest):
        return check(self.L.hm_batch_add(self.h, blob, len(blob), C.byref(dest)))

    def upload(self, stream=None):
        check(self.L.hm_batch_upload(self.h, stream))

    def execute(self, stages=3, stream=None):
        check(self.L.hm_batch_execute(self.h, stages, stream))

    def upload_execute(self, stages, chunks, copy_stream, stream):
        check(self.L.hm_batch_upload_execute(self.h, stages, chunks, copy_stream, stream))

    def set_colour(self, desc, n_images, p_y, p_cb, p_cr, p_out, images_per_group=0):
        check(self.L.hm_batch_set_colour(self.h, C.byref(desc) if desc is not None else None, n_images, p_y, p_cb, p_cr, p_out, images_per_group))

    def clear(self):
        self.L.hm_batch_clear(self.h)

    def set_profiling(self, slots=1):
        check(self.L.hm_batch_set_profiling(self.h, int(slots)))

    def set_concurrency(self, groups):
        check(self.L.hm_batch_set_concurrency(self.h, int(groups)))

    def timings_ms(self, slot=0):
        ms = (C.c_float * 3)()
        check(self.L.hm_batch_get_timings(self.h, slot, ms))
        return [ms[0], ms[1], ms[2]]

    def timings4_ms(self, slot=0):
        ms = (C.c_float * 4)()
        check(self.L.hm_batch_get_timings4(self.h, slot, ms))
        return [ms[0], ms[1], ms[2], ms[3]]

    def timings5_ms(self, slot=0):
        """[chains (or the whole reconstruction), deblocking, SAO + paste (or the fused tail), colour, residual pre-pass]"""
        ms = (C.c_float * 5)()
        check(self.L.hm_batch_get_timings5(self.h, slot, ms))
        return [ms[i] for i in range(5)]

    def check(self):
        """waits for the batch; raises when a reconstruction wave gave up a bounded wait (hm_batch_check)"""
        self.L.hm_batch_check.argtypes = [C.c_void_p]
        check(self.L.hm_batch_check(self.h))

    def tail_fused(self):
        """True when the batch's executes run the fused tail kernel (timings4_ms: its time is in slot 2)"""
        return bool(self.L.hm_batch_tail_fused(self.h))

    def algorithmic_bytes4(self):
        """(command streams, reconstructed samples, levels, residual samples) in bytes"""
        v = (C.c_uint64 * 4)()
        self.L.hm_batch_algorithmic_bytes4.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        check(self.L.hm_batch_algorithmic_bytes4(self.h, v))
        return [int(x) for x in v]

    def algorithmic_bytes(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(self.L.hm_batch_algorithmic_bytes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self.h:
            self.L.hm_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
