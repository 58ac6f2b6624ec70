"""The product's choice of colour-conversion chain (hm_colour_pipeline, colour_host.cpp) against the restatement of the
reference's pipeline search (oracle/pipeline_search.py: Dijkstra over ColorStates with the reference's tie-breaks,
colorconversion.cc:266-420) for every combination of sample depth, chroma format, nclx, target format and alpha the C
ABI offers; plus the search's own known answers (chains pinned by the reference fingerprints of BASELINE.md)."""
import ctypes as C
import itertools
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import pipeline_search as ps  # noqa: E402

DROP = "Op_drop_alpha_plane"
SWAP = "Op_RRGGBBaa_swap_endianness"


def pipe_of_chain(capi, chain):
    """HM_PIPE_* whose kernels compute the chain (alpha handling and the final byte swap are outside the pipe id)"""
    if chain is None:
        return None
    c = tuple(n for n in chain if n not in (DROP, SWAP))
    table = {
        ("Op_YCbCr420_to_RGB24",): capi.HM_PIPE_INT420,
        ("Op_YCbCr420_to_RGB32",): capi.HM_PIPE_INT420,
        ("Op_YCbCr_to_RGB<uint8_t>", "Op_RGB_to_RGB24_32"): capi.HM_PIPE_FLOAT,
        ("Op_YCbCr_to_RGB<uint16_t>", "Op_RGB_HDR_to_RRGGBBaa_BE"): capi.HM_PIPE_FLOAT,
        ("Op_YCbCr420_to_RRGGBBaa",): capi.HM_PIPE_FLOAT,  # the direct 4:2:0 op: the float op's arithmetic
        ("Op_to_hdr_planes", "Op_YCbCr420_to_RRGGBBaa"): capi.HM_PIPE_TO_HDR_FLOAT,
        ("Op_to_sdr_planes", "Op_YCbCr420_to_RGB24"): capi.HM_PIPE_SDR_INT420,
        ("Op_to_sdr_planes", "Op_YCbCr420_to_RGB32"): capi.HM_PIPE_SDR_INT420,
        ("Op_YCbCr_to_RGB<uint16_t>", "Op_to_sdr_planes", "Op_RGB_to_RGB24_32"): capi.HM_PIPE_FLOAT_SDR,
        ("Op_YCbCr_to_RGB<uint8_t>", "Op_to_hdr_planes", "Op_RGB_HDR_to_RRGGBBaa_BE"): capi.HM_PIPE_FLOAT_HDR,
        ("Op_YCbCr420_bilinear_to_YCbCr444<uint8_t>", "Op_YCbCr_to_RGB<uint8_t>", "Op_RGB_to_RGB24_32"): capi.HM_PIPE_BILINEAR_FLOAT,
        ("Op_YCbCr422_bilinear_to_YCbCr444<uint8_t>", "Op_YCbCr_to_RGB<uint8_t>", "Op_RGB_to_RGB24_32"): capi.HM_PIPE_BILINEAR_FLOAT,
        ("Op_YCbCr420_bilinear_to_YCbCr444<uint16_t>", "Op_YCbCr_to_RGB<uint16_t>", "Op_RGB_HDR_to_RRGGBBaa_BE"): capi.HM_PIPE_BILINEAR_FLOAT,
        ("Op_YCbCr422_bilinear_to_YCbCr444<uint16_t>", "Op_YCbCr_to_RGB<uint16_t>", "Op_RGB_HDR_to_RRGGBBaa_BE"): capi.HM_PIPE_BILINEAR_FLOAT,
        ("Op_mono_to_RGB24_32",): capi.HM_PIPE_MONO,
    }
    return table.get(c, ("unknown chain", c))


def test_search_known_answers():
    """chains the reference is known to run (their arithmetic reproduces BASELINE.md's fingerprints, tests/test_golden_heic.py)"""
    lim, full = ps.Nclx(1, 1, 1, False), ps.Nclx(6, 1, 13, True)
    assert ps.chain(ps.CS_YCBCR, ps.C_420, False, 8, lim, ps.CS_RGB, ps.C_RGB) == ["Op_YCbCr_to_RGB<uint8_t>", "Op_RGB_to_RGB24_32"]
    assert ps.chain(ps.CS_YCBCR, ps.C_420, False, 8, None, ps.CS_RGB, ps.C_RGB) == ["Op_YCbCr420_to_RGB24"]  # a grid canvas: no nclx
    assert ps.chain(ps.CS_YCBCR, ps.C_420, True, 8, full, ps.CS_RGB, ps.C_RGBA) == ["Op_YCbCr420_to_RGB32"]
    assert ps.chain(ps.CS_YCBCR, ps.C_420, True, 8, full, ps.CS_RGB, ps.C_RGB) == [DROP, "Op_YCbCr420_to_RGB24"]
    assert ps.chain(ps.CS_YCBCR, ps.C_420, False, 8, lim, ps.CS_RGB, ps.C_RRGGBB_LE) == ["Op_to_hdr_planes", "Op_YCbCr420_to_RRGGBBaa"]
    assert ps.chain(ps.CS_YCBCR, ps.C_422, False, 10, lim, ps.CS_RGB, ps.C_RRGGBB_LE) == \
        ["Op_YCbCr_to_RGB<uint16_t>", "Op_RGB_HDR_to_RRGGBBaa_BE", SWAP]
    assert ps.chain(ps.CS_MONO, ps.C_MONO, False, 8, None, ps.CS_RGB, ps.C_RGB) == ["Op_mono_to_RGB24_32"]
    # identical states: nothing to do; matrix 11: every YCbCr -> RGB op refuses
    assert ps.chain(ps.CS_RGB, ps.C_RGB, False, 8, None, ps.CS_RGB, ps.C_RGB) == []
    assert ps.chain(ps.CS_YCBCR, ps.C_444, False, 8, ps.Nclx(11, 1, 1, True), ps.CS_RGB, ps.C_RGB) is None


NCLX = [None, (6, 1, 13, 1), (1, 1, 1, 0), (1, 1, 1, 1), (9, 9, 16, 0), (0, 1, 1, 1), (0, 1, 1, 0), (8, 1, 1, 1), (2, 2, 2, 0), (11, 1, 1, 1), (14, 9, 16, 0)]
TARGETS = [ps.C_RGB, ps.C_RGBA, ps.C_RRGGBB_BE, ps.C_RRGGBB_LE, ps.C_RRGGBBAA_BE, ps.C_RRGGBBAA_LE]


@pytest.mark.parametrize("bpp", [8, 10, 12])
@pytest.mark.parametrize("chroma", [ps.C_MONO, ps.C_420, ps.C_422, ps.C_444])
def test_product_search_equals_oracle_search(pkg, bpp, chroma):
    """the product runs the same search (colour_search.cpp): identical op sequences for every request of the grid; the
    chains it cannot execute are refused loudly and only those"""
    capi, L = pkg.capi, pkg.lib()
    names = [n for n, _ in ps.OPS]
    executed = refused = 0
    for nclx, target, forced_bilinear, has_alpha in itertools.product(NCLX, TARGETS, (False, True), (0, 1)):
        n = ps.Nclx(nclx[0], nclx[1], nclx[2], bool(nclx[3])) if nclx else None
        opts = ps.Options(ps.DOWN_AVERAGE, ps.UP_BILINEAR, forced_bilinear)
        d = capi.ColourDesc(64, 64, bpp, chroma, 1 if nclx else 0, nclx[0] if nclx else 0, nclx[1] if nclx else 0, nclx[3] if nclx else 0,
                            target, 0, 0, 0, 0, 2 if forced_bilinear else 0, has_alpha)
        inp, tgt = ps.conversion_states(ps.CS_MONO if chroma == ps.C_MONO else ps.CS_YCBCR, chroma, bool(has_alpha), bpp, n, ps.CS_RGB, target)
        steps = ps.construct_pipeline(inp, tgt, opts)
        ops = (C.c_int * 8)()
        cnt = L.hm_colour_chain(C.byref(d), ops, 8)
        key = (bpp, chroma, nclx, target, forced_bilinear, has_alpha)
        if steps is None:
            assert cnt == -1, key
            assert L.hm_colour_pipeline(C.byref(d)) < 0, key
            continue
        assert [names[ops[i]] for i in range(cnt)] == [name for name, _ in steps], key
        pipe = L.hm_colour_pipeline(C.byref(d))
        exp = pipe_of_chain(capi, [name for name, _ in steps])
        if pipe < 0:
            refused += 1
            assert not isinstance(exp, int) or exp == capi.HM_PIPE_MONO and bpp != 8, ("refused although its kernels exist", key, exp)
        else:
            executed += 1
            if isinstance(exp, int):
                assert pipe == exp, key
    assert executed >= (0 if chroma == ps.C_MONO and bpp != 8 else 40), (executed, refused)
