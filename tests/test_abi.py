"""CPU: the C-ABI library loads and exports every symbol include/*.h declares."""
import ctypes as C
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    syms = set()
    for hdr in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = "\n".join(l for l in open(hdr).read().splitlines() if not l.lstrip().startswith("#"))
        for m in re.finditer(r"HM_API\s+[^;(]*?\b(\w+)\s*\(", text):
            syms.add(m.group(1))
    return syms


def test_exports(hm):
    syms = _declared_symbols()
    assert len(syms) >= 9
    missing = [s for s in sorted(syms) if not hasattr(hm, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_status_strings(hm):
    assert hm.hm_status_string(0) == b"ok"
    assert b"unsupported" in hm.hm_status_string(-2)
    assert hm.hm_version().startswith(b"heif-mi355x")


def test_plane_stride_matches_reference_observations(hm):
    # strides observable through heif_image_get_plane (pixelimage.cc:139-218)
    assert hm.hm_plane_stride(1280, 3) == 3840
    assert hm.hm_plane_stride(72, 3) == 224
    assert hm.hm_plane_stride(2560, 3) == 7680
    assert hm.hm_plane_stride(1280, 6) == 7680
    assert hm.hm_plane_stride(1, 1) == 64
    assert hm.hm_plane_stride(4032, 3) == 12096


def test_pipeline_selection(pkg, hm):
    D = pkg.capi.ColourDesc
    # grid canvas (no nclx), 8-bit 4:2:0 -> RGB24: integer op
    assert hm.hm_colour_pipeline(C.byref(D(64, 64, 8, 1, 0, 0, 0, 0, 10, 0, 0, 0, 0))) == 1
    # limited range single image -> float chain (row 1)
    assert hm.hm_colour_pipeline(C.byref(D(64, 64, 8, 1, 1, 2, 2, 0, 10, 0, 0, 0, 0))) == 2
    # 4:2:2 8-bit -> float chain (row 5); 10-bit 4:2:2 -> RRGGBB (row 4)
    assert hm.hm_colour_pipeline(C.byref(D(64, 64, 8, 2, 1, 1, 1, 1, 10, 0, 0, 0, 0))) == 2
    assert hm.hm_colour_pipeline(C.byref(D(64, 64, 10, 2, 1, 9, 9, 0, 14, 0, 0, 0, 0))) == 2
    # matrix 0 / 8 never use the integer op
    assert hm.hm_colour_pipeline(C.byref(D(64, 64, 8, 1, 1, 0, 1, 1, 10, 0, 0, 0, 0))) == 2


def test_coefficients_match_oracle(hm, oracle):
    class K(C.Structure):
        _fields_ = [("v", C.c_float * 4)]
    oracle.orc_ycbcr_to_rgb_coeffs.restype = K
    for has in (0, 1):
        for m in (0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 12, 13):
            for p in (1, 2, 9, 12):
                out = (C.c_float * 4)()
                assert hm.hm_ycbcr_coefficients(has, m, p, out) == 0
                exp = oracle.orc_ycbcr_to_rgb_coeffs(has, m, p)
                assert bytes(out) == bytes(exp.v), (has, m, p)
