"""ctypes wrapper of the test-stream synthesiser (tests/synth/libhm_synth.so)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "synth", "libhm_synth.so")


class SynthParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "chroma_format", "bit_depth", "log2_ctb", "log2_min_cb", "log2_min_tb", "log2_max_tb",
        "max_th_depth_intra", "qp", "cu_qp_delta", "diff_cu_qp_delta_depth", "sao", "deblock_disable",
        "sign_hiding", "transform_skip", "strong_intra", "cb_qp_offset", "cr_qp_offset",
        "beta_offset_div2", "tc_offset_div2", "vui", "full_range", "matrix", "primaries", "density", "wpp", "scaling_list", "pcm", "pcm_bits_y", "pcm_bits_c",
        "pcm_log2_min", "pcm_log2_max", "pcm_loop_filter_disable", "tq_bypass",
        "slices", "dependent", "tile_cols", "tile_rows", "tiles_uniform", "lf_across_tiles", "pps_lf_across_slices_off", "slice_lf_random",
        "deblock_override", "slice_sao_random", "slice_qp_random", "slice_chroma_qp", "conf_left", "conf_right", "conf_top", "conf_bottom",
        "rext_sps", "log2_max_ts", "cross_component", "chroma_qp_list", "chroma_qp_depth", "sao_scale_y", "sao_scale_c", "big_levels", "no_split")]

# bits of rext_sps (sps_range_extension flags in syntax order)
REXT_TS_ROTATION, REXT_TS_CONTEXT, REXT_IMPLICIT_RDPCM, REXT_EXPLICIT_RDPCM, REXT_EXTENDED_PRECISION = 1, 2, 4, 8, 16
REXT_NO_INTRA_SMOOTHING, REXT_HIGH_PRECISION_OFFSETS, REXT_PERSISTENT_RICE, REXT_BYPASS_ALIGNMENT = 32, 64, 128, 256


DEFAULTS = dict(width=64, height=64, chroma_format=1, bit_depth=8, log2_ctb=5, log2_min_cb=3, log2_min_tb=2,
                log2_max_tb=5, max_th_depth_intra=2, qp=27, cu_qp_delta=1, diff_cu_qp_delta_depth=1, sao=1,
                deblock_disable=0, sign_hiding=1, transform_skip=1, strong_intra=1, cb_qp_offset=0, cr_qp_offset=0,
                beta_offset_div2=0, tc_offset_div2=0, vui=1, full_range=1, matrix=6, primaries=1, density=60, wpp=0, scaling_list=0,
                pcm=0, pcm_bits_y=8, pcm_bits_c=8, pcm_log2_min=3, pcm_log2_max=5, pcm_loop_filter_disable=0, tq_bypass=0,
                slices=0, dependent=0, tile_cols=1, tile_rows=1, tiles_uniform=1, lf_across_tiles=1, pps_lf_across_slices_off=0,
                slice_lf_random=0, deblock_override=0, slice_sao_random=0, slice_qp_random=0, slice_chroma_qp=0,
                conf_left=0, conf_right=0, conf_top=0, conf_bottom=0,
                rext_sps=0, log2_max_ts=0, cross_component=0, chroma_qp_list=0, chroma_qp_depth=0, sao_scale_y=0, sao_scale_c=0, big_levels=0, no_split=0)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            subprocess.run(["make"], cwd=os.path.join(HERE, "synth"), check=True)
        _lib = C.CDLL(SO)
        _lib.hm_synth_picture.argtypes = [C.POINTER(SynthParams), C.c_uint64, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        _lib.hm_synth_free.argtypes = [C.c_void_p]
    return _lib


def picture(seed, **kw):
    """One coded intra picture as [u32 BE len][NAL]... bytes (VPS, SPS, PPS, IDR slice)."""
    cfg = dict(DEFAULTS)
    cfg.update(kw)
    cfg["log2_max_tb"] = min(cfg["log2_max_tb"], cfg["log2_ctb"], 5)
    cfg["pcm_log2_max"] = min(cfg["pcm_log2_max"], cfg["log2_ctb"], 5)
    cfg["pcm_log2_min"] = min(max(cfg["pcm_log2_min"], cfg["log2_min_cb"]), cfg["pcm_log2_max"])
    p = SynthParams(**cfg)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    rc = lib().hm_synth_picture(C.byref(p), seed, C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError(f"synth failed: {rc}")
    data = C.string_at(out, n.value)
    lib().hm_synth_free(out)
    return data
