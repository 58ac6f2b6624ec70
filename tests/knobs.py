"""Measurement / test scripts steer the library's tuning and fault-injection knobs through variables of THEIR environment
(HM_CHAIN_RING=4 tools/bench_classes.py ...).  The library itself reads no such variable (a stray one must not change a
production decode): this helper - test infrastructure - hands them to the exported test hook hm_debug_set
(heif-decoder-lib_amd/csrc/hm_internal.h) once, right after the library is loaded."""
import ctypes
import os

ENV_TO_KNOB = {
    "HM_CHAIN_SPIN_LIMIT": "chain_spin_limit", "HM_CHAIN_TEST_STALL": "chain_test_stall",
    "HM_CHAIN_PAIRS": "chain_pairs", "HM_CHAIN_SHARE": "chain_share", "HM_CHAIN_RING": "chain_ring", "HM_CHAIN_ALT": "chain_alt",
    "HM_CHAIN_NP": "chain_np", "HM_CHAIN_DEBUG": "chain_debug", "HM_RESID_SEGS": "resid_segs", "HM_RECON_WAVES": "recon_waves",
    "HM_QUAD_CLASS": "quad_class", "HM_TAIL_FUSED": "tail_fused", "HM_STREAM_INTERLEAVED": "stream_interleaved", "HM_CHAIN_SPLIT": "chain_split", "HM_TAIL_HDR16": "tail_hdr16", "HM_CHAIN_EARLY": "chain_early", "HM_GRID_SLAB_ROWS": "grid_slab_rows",
}


def set_knob(lib, name, value):
    lib.hm_debug_set.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.hm_debug_set.restype = ctypes.c_int
    if lib.hm_debug_set(name.encode(), int(value)) != 0:
        raise KeyError(f"unknown knob {name}")


def apply_env(lib):
    """-> the knobs that were set, {name: value}"""
    done = {}
    for env, knob in ENV_TO_KNOB.items():
        v = os.environ.get(env)
        if v not in (None, ""):
            set_knob(lib, knob, int(v))
            done[knob] = int(v)
    return done
