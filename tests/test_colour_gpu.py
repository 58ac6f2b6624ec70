"""GPU parity: fused YCbCr->RGB kernels (through the C ABI) vs the oracle restatement.
Bit-exact (integer / byte work; the float op chain is reproduced operation by operation)."""
import ctypes as C

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def _chroma_dims(w, h, chroma):
    if chroma == 1:
        return (w + 1) // 2, (h + 1) // 2
    if chroma == 2:
        return (w + 1) // 2, h
    return w, h


def _run_gpu(pkg, planes, w, h, bit_depth, chroma, nclx, out_fmt, upsampling=0, has_alpha=0):
    import torch
    capi = pkg.capi
    L = pkg.lib()
    (y, ys), (cb, cbs), (cr, crs) = planes
    obpp = orc.OUT_BYTES[out_fmt]
    ostride = L.hm_plane_stride(w, obpp)
    assert ostride == orc.plane_stride(w, obpp)
    dev = torch.device("cuda:0")
    dy, dcb, dcr = (torch.from_numpy(a).to(dev) for a in (y, cb, cr))
    rows = max(64, (h + 1) & ~1)
    dout = torch.zeros((rows, ostride), dtype=torch.uint8, device=dev)
    d = capi.ColourDesc(w, h, bit_depth, chroma, nclx[0], nclx[1], nclx[2], nclx[3], out_fmt, ys, cbs, crs, ostride, upsampling, has_alpha)
    stream = torch.cuda.current_stream().cuda_stream
    capi.check(L.hm_colour_convert(C.byref(d), dy.data_ptr(), dcb.data_ptr(), dcr.data_ptr(), dout.data_ptr(), stream))
    torch.cuda.synchronize()
    return dout.cpu().numpy(), ostride, obpp


SIZES = [(64, 64), (1280, 854), (4032, 3024), (72, 72), (17, 9), (1, 1), (1023, 3), (2, 2), (4030, 31)]


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("out_fmt", [10, 11])
@pytest.mark.parametrize("nclx", [(0, 0, 0, 0), (1, 6, 1, 1), (1, 1, 1, 1), (1, 9, 9, 1), (1, 12, 1, 1)])
def test_int420(pkg, w, h, out_fmt, nclx):
    rng = np.random.default_rng(w * 7919 + h + out_fmt)
    cw, ch = _chroma_dims(w, h, 1)
    planes = [orc.alloc_plane(w, h, 1, rng=rng), orc.alloc_plane(cw, ch, 1, rng=rng), orc.alloc_plane(cw, ch, 1, rng=rng)]
    d = pkg.capi.ColourDesc(w, h, 8, 1, *nclx, out_fmt, 0, 0, 0, 0)
    assert pkg.lib().hm_colour_pipeline(C.byref(d)) == pkg.capi.HM_PIPE_INT420
    got, ostride, obpp = _run_gpu(pkg, planes, w, h, 8, 1, nclx, out_fmt)
    exp, es = orc.colour_int(planes[0], planes[1], planes[2], w, h, nclx[0], nclx[1], nclx[2], out_fmt)
    assert es == ostride
    np.testing.assert_array_equal(got[:h, :w * obpp], exp[:h, :w * obpp])


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("chroma", [1, 2, 3])
@pytest.mark.parametrize("nclx", [(1, 2, 2, 0), (1, 6, 1, 0), (1, 1, 1, 0), (1, 0, 1, 1), (1, 0, 1, 0), (1, 8, 1, 1)])
def test_float_8bit(pkg, w, h, chroma, nclx):
    if chroma == 1 and nclx[3] == 1 and nclx[1] not in (0, 8, 11, 14):
        pytest.skip("reference picks the integer op for this state")
    rng = np.random.default_rng(w * 31 + h * 17 + chroma)
    cw, ch = _chroma_dims(w, h, chroma)
    planes = [orc.alloc_plane(w, h, 1, rng=rng), orc.alloc_plane(cw, ch, 1, rng=rng), orc.alloc_plane(cw, ch, 1, rng=rng)]
    for out_fmt in (10, 11):
        d = pkg.capi.ColourDesc(w, h, 8, chroma, *nclx, out_fmt, 0, 0, 0, 0)
        assert pkg.lib().hm_colour_pipeline(C.byref(d)) == pkg.capi.HM_PIPE_FLOAT
        got, ostride, obpp = _run_gpu(pkg, planes, w, h, 8, chroma, nclx, out_fmt)
        exp, es = orc.colour_float(planes[0], planes[1], planes[2], w, h, 8, chroma, *nclx, out_fmt)
        np.testing.assert_array_equal(got[:h, :w * obpp], exp[:h, :w * obpp])


@pytest.mark.parametrize("w,h", [(2048, 1536), (64, 64), (33, 5), (1, 1), (4032, 3024)])
@pytest.mark.parametrize("chroma", [1, 2, 3])
@pytest.mark.parametrize("bit_depth", [10, 12])
@pytest.mark.parametrize("nclx", [(0, 0, 0, 0), (1, 9, 9, 0), (1, 9, 9, 1), (1, 1, 1, 0), (1, 0, 1, 0), (1, 8, 1, 1)])
@pytest.mark.parametrize("out_fmt", [12, 14])
def test_float_hdr(pkg, w, h, chroma, bit_depth, nclx, out_fmt):
    rng = np.random.default_rng(w + h * 3 + chroma * 5 + bit_depth)
    cw, ch = _chroma_dims(w, h, chroma)
    mv = (1 << bit_depth) - 1
    planes = [orc.alloc_plane(w, h, 2, rng=rng, maxval=mv), orc.alloc_plane(cw, ch, 2, rng=rng, maxval=mv),
              orc.alloc_plane(cw, ch, 2, rng=rng, maxval=mv)]
    got, ostride, obpp = _run_gpu(pkg, planes, w, h, bit_depth, chroma, nclx, out_fmt)
    exp, es = orc.colour_float(planes[0], planes[1], planes[2], w, h, bit_depth, chroma, *nclx, out_fmt)
    np.testing.assert_array_equal(got[:h, :w * obpp], exp[:h, :w * obpp])


def test_no_silent_fallback(pkg):
    """unsupported states fail loudly instead of falling back"""
    d = pkg.capi.ColourDesc(64, 64, 8, 0, 0, 0, 0, 0, 14, 64, 0, 0, 384)  # monochrome -> RRGGBB: offered (test_monochrome_to_16bit_targets)
    assert pkg.lib().hm_colour_pipeline(C.byref(d)) >= 0
    d = pkg.capi.ColourDesc(64, 64, 8, 3, 1, 11, 1, 1, 10, 64, 64, 64, 192)  # matrix 11: every YCbCr -> RGB op of the reference refuses
    assert pkg.lib().hm_colour_pipeline(C.byref(d)) == -2


DEPTH_SIZES = [(64, 64), (1280, 854), (17, 9), (1, 1), (1023, 3), (4030, 31)]
DEPTH_NCLX = [(0, 0, 0, 0), (1, 9, 9, 0), (1, 1, 1, 1), (1, 1, 1, 0), (1, 0, 1, 1), (1, 0, 1, 0), (1, 8, 1, 1)]


def _random_planes(w, h, chroma, bit_depth, seed):
    rng = np.random.default_rng(seed)
    cw, ch = _chroma_dims(w, h, chroma)
    bps = 2 if bit_depth > 8 else 1
    mv = (1 << bit_depth) - 1
    return [orc.alloc_plane(w, h, bps, rng=rng, maxval=mv), orc.alloc_plane(cw, ch, bps, rng=rng, maxval=mv),
            orc.alloc_plane(cw, ch, bps, rng=rng, maxval=mv)]


@pytest.mark.parametrize("w,h", DEPTH_SIZES)
@pytest.mark.parametrize("chroma", [1, 2, 3])
@pytest.mark.parametrize("bit_depth", [10, 12])
@pytest.mark.parametrize("nclx", DEPTH_NCLX)
def test_hdr_image_to_8bit_rgb(pkg, w, h, chroma, bit_depth, nclx):
    """> 8-bit image -> RGB24 / RGBA32 (interleaved RGB targets are 8 bit whatever the image holds,
    colorconversion.cc:566-573): Op_to_sdr_planes + the integer op for full-range 4:2:0, else the float op at the image's
    depth + Op_to_sdr_planes + the interleave - the chains of the pipeline search, op by op in the oracle"""
    planes = _random_planes(w, h, chroma, bit_depth, w * 13 + h * 3 + chroma * 5 + bit_depth)
    matrix = nclx[1] if nclx[0] else 2
    full = nclx[3] if nclx[0] else 1
    int_chain = chroma == 1 and full and matrix not in (0, 8, 11, 14)
    for out_fmt in (10, 11):
        d = pkg.capi.ColourDesc(w, h, bit_depth, chroma, *nclx, out_fmt, 0, 0, 0, 0)
        assert pkg.lib().hm_colour_pipeline(C.byref(d)) == (pkg.capi.HM_PIPE_SDR_INT420 if int_chain else pkg.capi.HM_PIPE_FLOAT_SDR)
        got, ostride, obpp = _run_gpu(pkg, planes, w, h, bit_depth, chroma, nclx, out_fmt)
        exp, es, chain = orc.convert_by_search(planes, w, h, bit_depth, chroma, nclx, out_fmt)
        assert ("Op_to_sdr_planes" in chain) and (chain[0] == "Op_to_sdr_planes") == bool(int_chain)
        assert es == ostride
        np.testing.assert_array_equal(got[:h, :w * obpp], exp[:h, :w * obpp])


@pytest.mark.parametrize("w,h", DEPTH_SIZES)
@pytest.mark.parametrize("chroma", [1, 2, 3])
@pytest.mark.parametrize("nclx", DEPTH_NCLX)
@pytest.mark.parametrize("out_fmt", [12, 13, 14, 15])
@pytest.mark.parametrize("has_alpha", [0, 1])
def test_8bit_image_to_16bit_rgb(pkg, w, h, chroma, nclx, out_fmt, has_alpha):
    """8-bit image -> RRGGBB[AA] (a 16-bit target of an 8-bit image is 10 bit, colorconversion.cc:575-585).  The chain
    depends on chroma format, matrix AND on whether the image has an alpha plane (Op_to_hdr_planes before or after the
    float op: different arithmetic) - whatever the search picks, op by op in the oracle"""
    planes = _random_planes(w, h, chroma, 8, w * 11 + h * 5 + chroma * 3 + out_fmt)
    got, ostride, obpp = _run_gpu(pkg, planes, w, h, 8, chroma, nclx, out_fmt, has_alpha=has_alpha)
    exp, es, chain = orc.convert_by_search(planes, w, h, 8, chroma, nclx, out_fmt, has_alpha=bool(has_alpha))
    assert "Op_to_hdr_planes" in chain
    assert es == ostride
    np.testing.assert_array_equal(got[:h, :w * obpp], exp[:h, :w * obpp])


@pytest.mark.parametrize("w,h", [(64, 64), (33, 5), (1280, 854)])
@pytest.mark.parametrize("chroma", [1, 2, 3])
@pytest.mark.parametrize("bit_depth", [10, 12])
@pytest.mark.parametrize("out_fmt", [13, 15])
@pytest.mark.parametrize("has_alpha", [0, 1])
def test_hdr_image_to_rrggbbaa(pkg, w, h, chroma, bit_depth, out_fmt, has_alpha):
    """> 8-bit image -> RRGGBBAA: the float op (or the direct 4:2:0 op) and the 16-bit interleave; the alpha word of the
    colour kernel's output is (1 << bits) - 1 (an alpha plane is written over it by the caller: tests/test_transforms.py)"""
    planes = _random_planes(w, h, chroma, bit_depth, w + h * 7 + chroma + bit_depth)
    for nclx in [(1, 9, 9, 0), (1, 1, 1, 1), (0, 0, 0, 0)]:
        got, ostride, obpp = _run_gpu(pkg, planes, w, h, bit_depth, chroma, nclx, out_fmt, has_alpha=has_alpha)
        exp, es, chain = orc.convert_by_search(planes, w, h, bit_depth, chroma, nclx, out_fmt, has_alpha=bool(has_alpha))
        np.testing.assert_array_equal(got[:h, :w * obpp], exp[:h, :w * obpp])


@pytest.mark.parametrize("w,h", [(64, 64), (17, 9), (1023, 3)])
@pytest.mark.parametrize("chroma", [1, 2])
@pytest.mark.parametrize("bit_depth,out_fmt", [(8, 12), (8, 15), (10, 10), (12, 11)])
@pytest.mark.parametrize("nclx", [(0, 0, 0, 0), (1, 1, 1, 0), (1, 9, 9, 1)])
def test_forced_bilinear_with_depth_change(pkg, w, h, chroma, bit_depth, out_fmt, nclx):
    """forced bilinear upsampling combined with a change of sample depth: depth change, upsampling, float op and the
    second depth change in whatever order the search decides"""
    planes = _random_planes(w, h, chroma, bit_depth, w * 3 + h + chroma + bit_depth + out_fmt)
    got, ostride, obpp = _run_gpu(pkg, planes, w, h, bit_depth, chroma, nclx, out_fmt, upsampling=2)
    exp, es, chain = orc.convert_by_search(planes, w, h, bit_depth, chroma, nclx, out_fmt, forced_bilinear=True)
    assert any("bilinear" in c for c in chain)
    np.testing.assert_array_equal(got[:h, :w * obpp], exp[:h, :w * obpp])


@pytest.mark.parametrize("w,h", [(64, 64), (1280, 854), (72, 72), (17, 9), (1, 1), (2, 2), (3, 2), (2, 3), (1023, 3), (4030, 31), (5, 1), (1, 6)])
@pytest.mark.parametrize("chroma", [1, 2])
@pytest.mark.parametrize("bit_depth,out_fmt", [(8, 10), (8, 11), (10, 12), (12, 14)])
@pytest.mark.parametrize("nclx", [(0, 0, 0, 0), (1, 6, 1, 1), (1, 1, 1, 0), (1, 9, 9, 1), (1, 8, 1, 1)])
def test_forced_bilinear_chain(pkg, w, h, chroma, bit_depth, out_fmt, nclx):
    """Op_YCbCr420/422_bilinear_to_YCbCr444 (incl. the cx/2, cy/2 border quirk Q8) followed by
    the float op on 4:4:4, as convert_colorspace() chains them when the caller forces bilinear upsampling."""
    rng = np.random.default_rng(w * 131 + h * 7 + chroma + bit_depth)
    cw, ch = _chroma_dims(w, h, chroma)
    bps = 2 if bit_depth > 8 else 1
    mv = (1 << bit_depth) - 1
    planes = [orc.alloc_plane(w, h, bps, rng=rng, maxval=mv), orc.alloc_plane(cw, ch, bps, rng=rng, maxval=mv),
              orc.alloc_plane(cw, ch, bps, rng=rng, maxval=mv)]
    d = pkg.capi.ColourDesc(w, h, bit_depth, chroma, *nclx, out_fmt, 0, 0, 0, 0, 2)
    assert pkg.lib().hm_colour_pipeline(C.byref(d)) == 3
    got, ostride, obpp = _run_gpu(pkg, planes, w, h, bit_depth, chroma, nclx, out_fmt, upsampling=2)
    up = [orc.upsample_bilinear(planes[c], w, h, bit_depth, chroma) for c in (1, 2)]
    # the float op is the second step: it sees the intermediate state's profile (undefined -> sRGB defaults, cf. pipeline.cpu_decode)
    m2, p2 = (nclx[1], nclx[2]) if nclx[0] else (2, 2)
    step2 = (1, 6 if m2 == 2 else m2, 1 if p2 == 2 else p2, nclx[3] if nclx[0] else 1)
    exp, es = orc.colour_float(planes[0], up[0], up[1], w, h, bit_depth, 3, *step2, out_fmt)
    np.testing.assert_array_equal(got[:h, :w * obpp], exp[:h, :w * obpp])


def test_forced_bilinear_selection(pkg):
    L = pkg.lib()
    # 4:4:4 input needs no upsampling: the normal chain
    d = pkg.capi.ColourDesc(64, 64, 8, 3, 1, 6, 1, 1, 10, 0, 0, 0, 0, 2)
    assert L.hm_colour_pipeline(C.byref(d)) == pkg.capi.HM_PIPE_FLOAT
    # matrix 0: neither the bilinear op nor (forced) the nearest-neighbour ops accept -> no chain, loud failure
    d = pkg.capi.ColourDesc(64, 64, 8, 1, 1, 0, 1, 1, 10, 0, 0, 0, 0, 2)
    assert L.hm_colour_pipeline(C.byref(d)) == -2
    # not forced: the cheaper integer op
    d = pkg.capi.ColourDesc(64, 64, 8, 1, 0, 0, 0, 0, 10, 0, 0, 0, 0, 1)
    assert L.hm_colour_pipeline(C.byref(d)) == pkg.capi.HM_PIPE_INT420


@pytest.mark.parametrize("n", [1, 3, 33, 70])
@pytest.mark.parametrize("out_fmt", [10, 11])
def test_batched_conversion_equals_single(pkg, n, out_fmt):
    """hm_colour_convert_batch (one launch per 32 images of equal geometry) == n calls of hm_colour_convert"""
    import torch
    L = pkg.lib()
    w, h = 200, 74
    rng = np.random.default_rng(n * 5 + out_fmt)
    dev = torch.device("cuda:0")
    obpp = 3 if out_fmt == 10 else 4
    ostride = L.hm_plane_stride(w, obpp)
    imgs = []
    for _ in range(n):
        planes = [orc.alloc_plane(w, h, 1, rng=rng), orc.alloc_plane((w + 1) // 2, (h + 1) // 2, 1, rng=rng), orc.alloc_plane((w + 1) // 2, (h + 1) // 2, 1, rng=rng)]
        t = [torch.from_numpy(p[0]).to(dev) for p in planes]
        imgs.append((planes, t, torch.zeros((max(64, (h + 1) & ~1), ostride), dtype=torch.uint8, device=dev)))
    d = pkg.capi.ColourDesc(w, h, 8, 1, 0, 0, 0, 0, out_fmt, imgs[0][0][0][1], imgs[0][0][1][1], imgs[0][0][2][1], ostride)
    Arr = C.c_void_p * n
    ptrs = [Arr(*[im[1][k].data_ptr() for im in imgs]) for k in range(3)] + [Arr(*[im[2].data_ptr() for im in imgs])]
    L.hm_colour_convert_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    pkg.capi.check(L.hm_colour_convert_batch(C.byref(d), n, ptrs[0], ptrs[1], ptrs[2], ptrs[3], torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for planes, _, out in imgs:
        exp, _ = orc.colour_int(planes[0], planes[1], planes[2], w, h, 0, 0, 0, out_fmt)
        np.testing.assert_array_equal(out.cpu().numpy()[:h, :w * obpp], exp[:h, :w * obpp])


@pytest.mark.parametrize("w,h", [(64, 64), (1280, 854), (17, 9), (2, 2), (1023, 3)])
@pytest.mark.parametrize("nclx", [(0, 0, 0, 0), (1, 2, 2, 0), (1, 6, 1, 1), (1, 1, 1, 0), (1, 9, 9, 1)])
@pytest.mark.parametrize("out_fmt", [12, 14])
def test_8bit_to_rrggbb_chain(pkg, w, h, nclx, out_fmt):
    """8-bit 4:2:0 -> RRGGBB: the target becomes 10 bit (colorconversion.cc:575-585), reached by Op_to_hdr_planes +
    Op_YCbCr420_to_RRGGBBaa; the second op reads the intermediate state's profile (matrix 2 -> 6).  Pinned end to end by
    BASELINE.md's RRGGBB_LE fingerprints of example.heic (tests/test_golden_heic.py)."""
    rng = np.random.default_rng(w * 3 + h + out_fmt)
    cw, ch = _chroma_dims(w, h, 1)
    planes = [orc.alloc_plane(w, h, 1, rng=rng), orc.alloc_plane(cw, ch, 1, rng=rng), orc.alloc_plane(cw, ch, 1, rng=rng)]
    d = pkg.capi.ColourDesc(w, h, 8, 1, *nclx, out_fmt, 0, 0, 0, 0)
    assert pkg.lib().hm_colour_pipeline(C.byref(d)) == 4
    got, ostride, obpp = _run_gpu(pkg, planes, w, h, 8, 1, nclx, out_fmt)
    o = orc.load()
    hi = []
    for (buf, st), (pw, ph) in zip(planes, ((w, h), (cw, ch), (cw, ch))):
        b2, s2 = orc.alloc_plane(pw, ph, 2)
        o.orc_to_hdr_plane(orc.ptr(buf), st, pw, ph, 10, orc.ptr(b2), s2)
        hi.append((b2, s2))
    m2, p2 = (nclx[1], nclx[2]) if nclx[0] else (2, 2)
    step2 = (1, 6 if m2 == 2 else m2, 1 if p2 == 2 else p2, nclx[3] if nclx[0] else 1)
    exp, es = orc.colour_float(hi[0], hi[1], hi[2], w, h, 10, 1, *step2, out_fmt)
    np.testing.assert_array_equal(got[:h, :w * obpp], exp[:h, :w * obpp])


@pytest.mark.parametrize("w,h", [(64, 64), (33, 5), (1, 1), (1280, 854), (1023, 3)])
@pytest.mark.parametrize("bit_depth", [8, 10, 12])
@pytest.mark.parametrize("out_fmt", [12, 13, 14, 15])
@pytest.mark.parametrize("has_alpha", [0, 1])
def test_monochrome_to_16bit_targets(pkg, w, h, bit_depth, out_fmt, has_alpha):
    """monochrome image -> RRGGBB[AA]: the reference's chain starts with Op_mono_to_YCbCr420 (neutral chroma planes,
    monochrome.cc:26-155) and continues as for a 4:2:0 image whose profile is the sRGB default set - whatever the
    monochrome image itself declared (limited range, BT.709 ...)"""
    import torch
    rng = np.random.default_rng(w * 7 + h + bit_depth + out_fmt)
    bps = 2 if bit_depth > 8 else 1
    y = orc.alloc_plane(w, h, bps, rng=rng, maxval=(1 << bit_depth) - 1)
    capi, L = pkg.capi, pkg.lib()
    obpp = orc.OUT_BYTES[out_fmt]
    ostride = L.hm_plane_stride(w, obpp)
    dev = torch.device("cuda:0")
    for nclx in [(0, 0, 0, 0), (1, 1, 1, 0), (1, 9, 9, 1)]:
        dy = torch.from_numpy(y[0]).to(dev)
        dout = torch.zeros((max(64, (h + 1) & ~1), ostride), dtype=torch.uint8, device=dev)
        d = capi.ColourDesc(w, h, bit_depth, 0, nclx[0], nclx[1], nclx[2], nclx[3], out_fmt, y[1], 0, 0, ostride, 0, has_alpha)
        capi.check(L.hm_colour_convert(C.byref(d), dy.data_ptr(), None, None, dout.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        got = dout.cpu().numpy()
        exp, es, chain = orc.convert_by_search([y, None, None], w, h, bit_depth, 0, nclx, out_fmt, has_alpha=bool(has_alpha))
        assert "Op_mono_to_YCbCr420" in chain and es == ostride
        np.testing.assert_array_equal(got[:h, :w * obpp], exp[:h, :w * obpp])
