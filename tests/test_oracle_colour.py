"""CPU: pin the colour oracle against the reference's own known-answer vectors."""
import json
import os

import numpy as np

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_bilinear_kat(oracle):
    # data of the reference's "Bilinear upsampling" test (tests/conversion.cc:635-670)
    kat = json.load(open(os.path.join(GOLD, "bilinear_kat.json")))
    for case in kat["cases"]:
        src = np.array(case["in"], np.uint8).reshape(2, 2)
        out = np.zeros((4, 4), np.uint8)
        oracle.orc_upsample_bilinear_420(orc.ptr(src), 2, 4, 4, orc.ptr(out), 4)
        assert out.flatten().tolist() == case["out"]


def test_integer_constants(oracle):
    # BT.601 full-range constants 359 / -88 / -183 / 454 (yuv2rgb.cc:336-339)
    w = h = 2
    y = (np.full((64, 64), 100, np.uint8), 64)
    cb = (np.full((64, 64), 129, np.uint8), 64)
    cr = (np.full((64, 64), 127, np.uint8), 64)
    out, os_ = orc.colour_int(y, cb, cr, w, h, 0, 0, 0, 10)
    r = 100 + ((359 * -1 + 128) >> 8)
    g = 100 + ((-88 * 1 + -183 * -1 + 128) >> 8)
    b = 100 + ((454 * 1 + 128) >> 8)
    assert out[0, :3].tolist() == [r, g, b]


def test_paste_rescale_quirk(oracle):
    # context.cc:2504-2528: limited->full rescale with the luma offset for chroma too (Q2)
    tile = np.arange(64 * 64, dtype=np.uint32).astype(np.uint8).reshape(64, 64)
    canvas = np.zeros((64, 128), np.uint8)
    assert oracle.orc_paste_tile_plane(orc.ptr(tile), 64, 64, 64, orc.ptr(canvas), 128, 100, 64, 64, 0, 0, 1, 8, 1, 0, 2) == 0
    f32 = np.float32
    exp = np.floor(((tile[:, :36].astype(f32) - f32(16)) * f32(1.1689)) + f32(0.5)).clip(0, 255).astype(np.uint8)
    # trunc == floor for x+0.5 >= 0; negatives clip to 0 either way
    np.testing.assert_array_equal(canvas[:, 64:100], exp)
    assert not canvas[:, :64].any() and not canvas[:, 100:].any()
    cb = np.full((32, 32), 128, np.uint8)
    cc = np.zeros((32, 64), np.uint8)
    assert oracle.orc_paste_tile_plane(orc.ptr(cb), 32, 32, 32, orc.ptr(cc), 64, 100, 64, 0, 0, 1, 1, 8, 1, 0, 2) == 0
    assert cc[0, 0] == 128  # (128-16)*1.1429 = 128.0048 -> 128
    # origin outside canvas -> error
    assert oracle.orc_paste_tile_plane(orc.ptr(tile), 64, 64, 64, orc.ptr(canvas), 128, 64, 64, 64, 0, 0, 1, 8, 0, 1, 1) == -1


def test_bilinear_variants(oracle):
    """The 16-bit instantiation of the 4:2:0 op equals the 8-bit one on 8-bit data, and the 4:2:2 op
    (chroma_sampling.cc:766-933) follows its 3/4-1/4 rule with copied borders.  (Only the 8-bit 4:2:0 op has a
    reference KAT; these restatements are pinned by it through the shared template.)"""
    rng = np.random.default_rng(5)
    for w, h in ((4, 4), (7, 5), (16, 9), (33, 32), (2, 2), (1, 1), (3, 1), (1, 3)):
        cw, ch = (w + 1) // 2, (h + 1) // 2
        src = rng.integers(0, 256, (ch, cw), dtype=np.uint8)
        o8 = np.zeros((h, w), np.uint8)
        oracle.orc_upsample_bilinear_420(orc.ptr(src), cw, w, h, orc.ptr(o8), w)
        src16 = src.astype(np.uint16)
        o16 = np.zeros((h, w), np.uint16)
        oracle.orc_upsample_bilinear_420_u16(orc.ptr(src16), cw, w, h, orc.ptr(o16), w)
        np.testing.assert_array_equal(o8, o16)
    row = np.array([[10, 20, 30]], np.uint8)
    out = np.zeros((1, 6), np.uint8)
    oracle.orc_upsample_bilinear_422(orc.ptr(row), 3, 6, 1, orc.ptr(out), 6)
    assert out.tolist() == [[10, 13, 18, 23, 28, 30]]
    out5 = np.zeros((1, 5), np.uint8)
    oracle.orc_upsample_bilinear_422(orc.ptr(row), 3, 5, 1, orc.ptr(out5), 5)
    assert out5.tolist() == [[10, 13, 18, 23, 28]]
    row16 = (row.astype(np.uint16) * 4)
    o16 = np.zeros((1, 6), np.uint16)
    oracle.orc_upsample_bilinear_422_u16(orc.ptr(row16), 3, 6, 1, orc.ptr(o16), 6)
    assert o16.tolist() == [[40, 50, 70, 90, 110, 120]]


def test_paste_rescale_integer_form(oracle):
    """The fused tail kernels (filters.hip: pk_rescale) rescale 8-bit samples with
    t = max(v, 16) - 16; out = min(255, t + ((t * A + B) >> S)), (A, B, S) = (173, 507, 10) luma / (73, 256, 9) chroma,
    every intermediate below 2^16: equal to the oracle's float paste (context.cc:2504-2528) for all 256 values of both
    kinds of plane (the kernel source checks the same identity at compile time)."""
    ramp = np.tile(np.arange(256, dtype=np.uint8), (64, 1))
    for c, (A, B, S) in ((0, (173, 507, 10)), (1, (73, 256, 9))):
        canvas = np.zeros((64, 256), np.uint8)
        assert oracle.orc_paste_tile_plane(orc.ptr(ramp), 256, 256, 64, orc.ptr(canvas), 256, 256, 64, 0, 0, c, 3, 8, 1, 0, 2) == 0
        t = np.maximum(np.arange(256), 16) - 16
        assert (t * A + B).max() < 65536
        form = np.minimum(255, t + ((t * A + B) >> S))
        np.testing.assert_array_equal(canvas[0], form)
