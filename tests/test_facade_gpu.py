"""GPU: the libheif-compatible facade (libheif_mi355x_api.so) driven exactly like the reference's
README sample / examples/heif_dec.cc:412-554, and the decoder plugin driven like
HeifContext::decode_image_planar drives a plugin (context.cc:1787-1835)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import heifwriter
import orc
import pipeline

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = json.load(open(os.path.join(HERE, "golden", "heic.json")))


class Err(C.Structure):
    _fields_ = [("code", C.c_int), ("subcode", C.c_int), ("message", C.c_char_p)]


@pytest.fixture(scope="module")
def api(pkg):
    pkg.lib()  # loads torch's HIP runtime + the core library first
    a = C.CDLL(os.path.join(ROOT, "heif-decoder-lib_amd", "libheif_mi355x_api.so"))
    a.heif_context_alloc.restype = C.c_void_p
    a.heif_context_free.argtypes = [C.c_void_p]
    a.heif_context_read_from_memory.restype = Err
    a.heif_context_read_from_memory.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p]
    a.heif_context_get_primary_image_handle.restype = Err
    a.heif_context_get_primary_image_handle.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    a.heif_context_get_image_handle.restype = Err
    a.heif_context_get_image_handle.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    a.heif_context_set_threads.restype = None  # heif.h:1015: void f(ctx, const handle*, int)
    a.heif_context_set_threads.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    a.heif_image_get_decoding_warnings.argtypes = [C.c_void_p, C.c_int, C.POINTER(Err), C.c_int]
    a.heif_image_get_colorspace.argtypes = [C.c_void_p]
    a.heif_image_get_chroma_format.argtypes = [C.c_void_p]
    a.heif_image_has_channel.argtypes = [C.c_void_p, C.c_int]
    a.heif_image_handle_get_color_profile_type.argtypes = [C.c_void_p]
    a.heif_image_handle_get_raw_color_profile_size.restype = C.c_size_t
    a.heif_image_handle_get_raw_color_profile_size.argtypes = [C.c_void_p]
    a.heif_image_handle_get_raw_color_profile.restype = Err
    a.heif_image_handle_get_raw_color_profile.argtypes = [C.c_void_p, C.c_char_p]
    a.heif_image_get_color_profile_type.argtypes = [C.c_void_p]
    a.heif_image_get_raw_color_profile_size.restype = C.c_size_t
    a.heif_image_get_raw_color_profile_size.argtypes = [C.c_void_p]
    a.heif_image_get_raw_color_profile.restype = Err
    a.heif_image_get_raw_color_profile.argtypes = [C.c_void_p, C.c_char_p]
    a.heif_image_handle_release.argtypes = [C.c_void_p]
    a.heif_image_handle_get_width.argtypes = [C.c_void_p]
    a.heif_decode_image.restype = Err
    a.heif_decode_image.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]
    a.heif_image_get_plane_readonly.restype = C.POINTER(C.c_uint8)
    a.heif_image_get_plane_readonly.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    a.heif_image_get_width.argtypes = [C.c_void_p, C.c_int]
    a.heif_image_get_height.argtypes = [C.c_void_p, C.c_int]
    a.heif_image_get_bits_per_pixel_range.argtypes = [C.c_void_p, C.c_int]
    a.heif_image_release.argtypes = [C.c_void_p]
    a.heif_decoding_options_alloc.restype = C.c_void_p
    a.heif_decoding_options_free.argtypes = [C.c_void_p]
    a.heif_decoding_options_add_external_dest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    return a


def _decode(api, data, item, colorspace, chroma, threads=0, options=None):
    ctx = api.heif_context_alloc()
    e = api.heif_context_read_from_memory(ctx, data, len(data), None)
    assert e.code == 0, e.message
    h = C.c_void_p()
    e = api.heif_context_get_image_handle(ctx, item, C.byref(h)) if item else api.heif_context_get_primary_image_handle(ctx, C.byref(h))
    assert e.code == 0, e.message
    if threads:
        api.heif_context_set_threads(ctx, h, threads)
    img = C.c_void_p()
    e = api.heif_decode_image(h, C.byref(img), colorspace, chroma, options)
    return ctx, h, img, e


@pytest.mark.parametrize("case", GOLD["cases"][:4], ids=lambda c: f"{c['file']}-{c['item']}-{c['fmt']}")
def test_heif_decode_image_like_reference_sample(api, case):
    data = open(os.path.join(HERE, "data", case["file"]), "rb").read()
    ctx, h, img, e = _decode(api, data, case["item"], 1, case["fmt"])  # heif_colorspace_RGB
    assert e.code == 0, e.message
    stride = C.c_int()
    p = api.heif_image_get_plane_readonly(img, 10, C.byref(stride))  # heif_channel_interleaved
    w, hh = api.heif_image_get_width(img, 10), api.heif_image_get_height(img, 10)
    assert (w, hh) == (case["w"], case["h"])
    bpp = {10: 3, 11: 4, 12: 6, 14: 6}[case["fmt"]]
    buf = np.ctypeslib.as_array(p, shape=(hh, stride.value))
    assert pipeline.survey_fnv(np.ascontiguousarray(buf), stride.value, w * bpp, hh) == case["fnv"]
    api.heif_image_release(img)
    api.heif_image_handle_release(h)
    api.heif_context_free(ctx)


def test_grid_with_threads_and_ext_dst(api, hm):
    g = GOLD["grid_1x2"]
    f = pipeline.HeifFile(hm, open(os.path.join(HERE, "data", "example.heic"), "rb").read())
    tiles = [f.hevc_data(i) for i in g["tiles"]]
    f.close()
    data = heifwriter.write_heic(tiles, (1280, 854), grid=(1, 2, 2560, 854))
    # RGB24 with 4 host threads (heif_context_set_threads: tile fan-out)
    ctx, h, img, e = _decode(api, data, 0, 1, 10, threads=4)
    assert e.code == 0, e.message
    stride = C.c_int()
    p = api.heif_image_get_plane_readonly(img, 10, C.byref(stride))
    assert stride.value == g["stride"]
    buf = np.ascontiguousarray(np.ctypeslib.as_array(p, shape=(g["h"], stride.value)))
    assert pipeline.survey_fnv(buf, stride.value, g["w"] * 3, g["h"]) == g["fnv"]
    api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx)
    # RGBA into a caller buffer (fork API heif_decoding_options_add_external_dest, android_jni heif_jni.cpp:80-97)
    ext = np.zeros((g["h"], g["w"] * 4), np.uint8)
    opt = api.heif_decoding_options_alloc()
    api.heif_decoding_options_add_external_dest(opt, ext.ctypes.data_as(C.c_void_p), ext.size, g["w"] * 4)
    ctx, h, img, e = _decode(api, data, 0, 1, 11, options=opt)
    assert e.code == 0, e.message
    exp, es, _ = pipeline.cpu_decode(hm, tiles, 1280, 854, g["w"], g["h"], 2, True, 11)
    np.testing.assert_array_equal(ext, exp[:g["h"], :g["w"] * 4])
    api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx); api.heif_decoding_options_free(opt)


def test_unsupported_requests_fail_loudly(api):
    data = open(os.path.join(HERE, "data", "colors-no-alpha.heic"), "rb").read()
    ctx, h, img, e = _decode(api, data, 0, 1, 3)  # planar RGB 4:4:4 target: only the interleaved targets are on the GPU path
    assert e.code == 4 and not img  # heif_error_Unsupported_feature
    api.heif_image_handle_release(h); api.heif_context_free(ctx)
    # 8-bit -> RRGGBB_LE is served (Op_to_hdr_planes + the 4:2:0 HDR op): 10-bit values in 16-bit words
    ctx, h, img, e = _decode(api, data, 0, 1, 14)
    assert e.code == 0 and img, e.message
    assert api.heif_image_get_bits_per_pixel_range(img, 10) == 10
    api.heif_image_release(img)
    api.heif_image_handle_release(h); api.heif_context_free(ctx)


def test_decoder_plugin_call_sequence(api, hm):
    """new_decoder -> set_strict_decoding -> push_data -> decode_image -> free_decoder, planes == oracle"""
    import corpus
    import hevcutil

    class Plugin(C.Structure):
        _fields_ = [("plugin_api_version", C.c_int), ("get_plugin_name", C.c_void_p), ("init_plugin", C.c_void_p),
                    ("deinit_plugin", C.c_void_p), ("does_support_format", C.CFUNCTYPE(C.c_int, C.c_int)),
                    ("new_decoder", C.CFUNCTYPE(Err, C.POINTER(C.c_void_p), C.c_int)), ("free_decoder", C.CFUNCTYPE(None, C.c_void_p)),
                    ("push_data", C.CFUNCTYPE(Err, C.c_void_p, C.c_char_p, C.c_size_t)),
                    ("decode_image", C.CFUNCTYPE(Err, C.c_void_p, C.POINTER(C.c_void_p))),
                    ("set_strict_decoding", C.CFUNCTYPE(None, C.c_void_p, C.c_int)), ("id_name", C.c_char_p)]
    api.hm_get_decoder_plugin.restype = C.POINTER(Plugin)
    api.heif_register_decoder_plugin.restype = Err
    api.heif_register_decoder_plugin.argtypes = [C.POINTER(Plugin)]
    pl = api.hm_get_decoder_plugin()
    assert api.heif_register_decoder_plugin(pl).code == 0
    p = pl.contents
    assert p.does_support_format(1) == 150
    # 4:2:0, 4:2:2, 4:0:0 (monochrome colourspace, one plane) and 4:4:4 (full-size chroma), 8 and 10 bit, a picture
    # with a conformance window (ragged: 72x40 in 32-pixel CTBs): decoder_libde265.cc:88-157
    for name in ("tile512_novui", "hi422_10", "mono8", "mono10", "yuv444_8", "yuv444_10_ctb64_wpp", "ragged", "hi420_10", "conf_window", "conf_window_422_10", "slices_mono_ctb16"):
        data = corpus.stream(name)
        dec = C.c_void_p()
        assert p.new_decoder(C.byref(dec), 0).code == 0
        p.set_strict_decoding(dec, 0)
        half = len(data) // 2
        assert p.push_data(dec, data[:half], half).code == 0          # data may arrive in several pieces
        assert p.push_data(dec, data[half:], len(data) - half).code == 0
        img = C.c_void_p()
        e = p.decode_image(dec, C.byref(img))
        assert e.code == 0, e.message
        p.free_decoder(dec)
        exp, info = orc.oracle_decode(hevcutil.parse(hm, data), 3, crop=True)
        wide = info["bit_depth"] > 8
        cf = info["chroma"]
        assert api.heif_image_get_colorspace(img) == (2 if cf == 0 else 0)  # heif_colorspace_monochrome / YCbCr
        assert api.heif_image_get_chroma_format(img) == cf
        assert api.heif_image_has_channel(img, 1) == (0 if cf == 0 else 1)
        W, H = info["width"], info["height"]
        for c in range(1 if cf == 0 else 3):
            ew = W if (c == 0 or cf == 3) else W // 2
            eh = H if (c == 0 or cf != 1) else H // 2
            assert (api.heif_image_get_width(img, c), api.heif_image_get_height(img, c)) == (ew, eh)
            stride = C.c_int()
            ptr = api.heif_image_get_plane_readonly(img, c, C.byref(stride))
            w, hgt = api.heif_image_get_width(img, c), api.heif_image_get_height(img, c)
            assert api.heif_image_get_bits_per_pixel_range(img, c) == info["bit_depth"]
            raw = np.ctypeslib.as_array(ptr, shape=(hgt, stride.value))
            got = raw[:, :w * 2].copy().view(np.uint16).reshape(hgt, w) if wide else raw[:, :w].astype(np.uint16)
            np.testing.assert_array_equal(got, exp[c][:hgt, :w])
        api.heif_image_release(img)


def test_plugin_driven_like_the_reference_drives_a_grid(api, hm):
    """context.cc:2361-2401: the tiles of a grid, one decoder instance each, from a window of concurrent C++ tasks
    (tests/synth/plugin_driver.cpp).  Behind decode_image the concurrent calls meet in the device's shared worker and
    run as shared batches on several executor streams (picture.cpp): every tile of every round must still be the
    oracle's, whatever was batched with what - pictures of different classes (8 / 10 bit, 4:2:0 / 4:2:2 / 4:0:0, ragged
    sizes) mixed on purpose."""
    import corpus
    import hevcutil
    import pluginapi
    names = ["tile512_a", "hi422_10", "mono8", "ragged", "tile512_novui", "hi420_10", "conf_window", "ctb64_wpp"]
    datas = [bytes(corpus.stream(n)) for n in names]
    expected = []
    for d in datas:
        exp, info = orc.oracle_decode(hevcutil.parse(hm, d), 3, crop=True)
        expected.append((exp, info))
    saved = api.hm_get_decoder_plugin.restype
    api.hm_get_decoder_plugin.restype = C.c_void_p
    plugin_ptr = api.hm_get_decoder_plugin()
    api.hm_get_decoder_plugin.restype = saved
    tiles = [datas[i % len(datas)] for i in range(40)]
    for threads in (8, 3, 1):
        imgs = pluginapi.drive_grid(plugin_ptr, tiles, threads)
        try:
            for i, img in enumerate(imgs):
                exp, info = expected[i % len(datas)]
                wide = info["bit_depth"] > 8
                for c in range(1 if info["chroma"] == 0 else 3):
                    stride = C.c_int()
                    ptr = api.heif_image_get_plane_readonly(img, c, C.byref(stride))
                    w, hgt = api.heif_image_get_width(img, c), api.heif_image_get_height(img, c)
                    raw = np.ctypeslib.as_array(ptr, shape=(hgt, stride.value))
                    got = raw[:, :w * 2].copy().view(np.uint16).reshape(hgt, w) if wide else raw[:, :w].astype(np.uint16)
                    np.testing.assert_array_equal(got, exp[c][:hgt, :w], err_msg=f"tile {i} ({names[i % len(datas)]}) plane {c}, {threads} threads")
        finally:
            for img in imgs:
                api.heif_image_release(img)


def test_a_failing_picture_does_not_fail_its_batch_neighbours(hm, hm_hooks):
    """Concurrent decode_image calls share batches behind the device worker (picture.cpp).  A picture whose batch fails
    at execute (stood in for by the library's test hook: batches holding a picture 192 samples wide are refused) must
    fail alone: the valid tiles queued with it - from unrelated decoder instances - come back decoded.
    (r06: the hook is not in the libraries that ship; this test drives the facade linked against libheif_mi355x_test.so -
    the same objects + csrc/test_hooks.cpp.)"""
    import ctypes
    import corpus
    import hevcutil
    import pluginapi
    import synthutil
    api = C.CDLL(os.path.join(ROOT, "heif-decoder-lib_amd", "libheif_mi355x_api_test.so"))
    api.heif_image_release.argtypes = [C.c_void_p]
    api.heif_image_get_plane_readonly.restype = C.POINTER(C.c_uint8)
    api.heif_image_get_plane_readonly.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    api.heif_image_get_width.argtypes = [C.c_void_p, C.c_int]
    api.heif_image_get_height.argtypes = [C.c_void_p, C.c_int]
    hm_dbg = hm_hooks
    saved = api.hm_get_decoder_plugin.restype
    api.hm_get_decoder_plugin.restype = C.c_void_p
    pl = api.hm_get_decoder_plugin()
    api.hm_get_decoder_plugin.restype = saved
    good = bytes(corpus.stream("tile512_a"))
    bad = synthutil.picture(616161, width=192, height=64)
    assert hm_dbg.hm_debug_set(b"batch_fail_width", 192) == 0
    exp, info = orc.oracle_decode(hevcutil.parse(hm, good), 3, crop=True)
    for img in pluginapi.drive_grid(pl, [good], 1):  # (loads the driver)
        api.heif_image_release(img)
    drv = pluginapi._driver
    tiles = [good] * 5 + [bad] + [good] * 6
    n = len(tiles)
    try:
        _check_isolation(api, drv, pl, tiles, n, exp)
    finally:
        hm_dbg.hm_debug_set(b"batch_fail_width", 0)


def _check_isolation(api, drv, pl, tiles, n, exp):
    import ctypes
    for _ in range(3):
        data = (ctypes.c_char_p * n)(*tiles)
        size = (ctypes.c_size_t * n)(*[len(t) for t in tiles])
        out = (ctypes.c_void_p * n)()
        rc = drv.hm_test_drive_grid(ctypes.cast(pl, ctypes.c_void_p), data, size, n, n, out)
        assert rc != 0  # the marked picture is refused (heif_error code of the failed tile) ...
        for i in range(n):
            if i == 5:
                assert not out[i]
                continue
            assert out[i], f"tile {i} failed with its neighbour"  # ... and only it
            img = ctypes.c_void_p(out[i])
            stride = ctypes.c_int()
            ptr = api.heif_image_get_plane_readonly(img, 0, ctypes.byref(stride))
            w, h = api.heif_image_get_width(img, 0), api.heif_image_get_height(img, 0)
            got = np.ctypeslib.as_array(ptr, shape=(h, stride.value))[:, :w]
            np.testing.assert_array_equal(got, exp[0][:h, :w])
            api.heif_image_release(img)


@pytest.mark.parametrize("strict", [0, 1])
def test_damaged_streams_through_the_plugin(api, hm, strict):
    """The contract at the plugin boundary for damaged slice data ("Damaged streams"), over the stream
    classes of tools/fuzz_ref.py with 1-2 flipped bits: decode_image either hands out a picture - then it is the oracle's
    picture of the same bytes - or fails the way the reference's caller sees a failed libde265 decode: heif_error_
    Decoder_plugin_error / heif_suberror_Unspecified (context.cc:1826-1830), End_of_data for broken [length][NAL] framing
    (decoder_libde265.cc:276-292), Unsupported_feature for syntax outside the decoder; never a crash, never another code.
    r05: without strict decoding a picture with damaged SLICE DATA is handed out like the reference's plugin hands it out
    (decoder_libde265.cc:311-336) - decoded up to the damage, concealed behind it (HM_PARSE_CONCEAL) - and equals the oracle's
    picture of the same concealing parse; with strict decoding (an extension of what the reference's flag covers) it is refused."""
    import random
    import corpus
    import hevcutil
    import pluginapi
    saved = api.hm_get_decoder_plugin.restype
    api.hm_get_decoder_plugin.restype = C.POINTER(pluginapi.Plugin)
    p = api.hm_get_decoder_plugin().contents
    api.hm_get_decoder_plugin.restype = saved
    rng = random.Random(4242 + strict)
    seen = {"ok": 0, "failed": 0, "concealed": 0}

    def mutate(data):
        b = bytearray(data)
        for _ in range(rng.randrange(1, 3)):
            b[rng.randrange(len(b) // 3, len(b))] ^= 1 << rng.randrange(8)
        return bytes(b)

    for name in ("ragged", "ctb64_wpp", "hi422_10", "ctb16_nosao", "mono10", "slices_headers", "tiles_3x2_nolf", "dense_lowqp"):
        data = bytes(corpus.stream(name))
        # ten mutations as they come (nearly all of them are refused), one that the parser still accepts (searched on the
        # host: a few in a hundred are), one with a truncated last NAL record
        cases = [mutate(data) for _ in range(10)]
        for _ in range(400):
            b = mutate(data)
            try:
                hevcutil.parse(hm, b)
            except RuntimeError:
                continue
            cases.append(b)
            break
        cases.append(data[:len(data) - 7])
        for t, b in enumerate(cases):
            truncated = t == len(cases) - 1
            b = bytes(b)
            dec = C.c_void_p()
            assert p.new_decoder(C.byref(dec), 0).code == 0
            p.set_strict_decoding(dec, strict)
            assert p.push_data(dec, b, len(b)).code == 0
            img = C.c_void_p()
            e = p.decode_image(dec, C.byref(img))
            p.free_decoder(dec)
            if e.code == 0:
                assert img
                # (strict: the parser takes the stream as it is or not at all; otherwise damaged slice data is concealed, HM_PARSE_CONCEAL)
                blob = hevcutil.parse(hm, b) if strict else hevcutil.parse_concealing(hm, b)[0]
                seen["concealed"] += 0 if strict else hevcutil.parse_concealing(hm, b)[1] > 0
                exp, info = orc.oracle_decode(blob, 3, crop=True)
                stride = C.c_int()
                ptr = api.heif_image_get_plane_readonly(img, 0, C.byref(stride))
                w, h = api.heif_image_get_width(img, 0), api.heif_image_get_height(img, 0)
                raw = np.ctypeslib.as_array(ptr, shape=(h, stride.value))
                got = raw[:, :w * 2].copy().view(np.uint16).reshape(h, w) if info["bit_depth"] > 8 else raw[:, :w].astype(np.uint16)
                np.testing.assert_array_equal(got, exp[0][:h, :w], err_msg=f"{name} mutation {t}")
                api.heif_image_release(img)
                seen["ok"] += 1
            else:
                assert not img
                assert (e.code, e.subcode) in ((7, 0), (7, 100), (4, 3000)) or e.code == 4, (name, t, e.code, e.subcode, e.message)
                if truncated:
                    assert (e.code, e.subcode) == (7, 100), (name, e.code, e.subcode, e.message)  # Decoder_plugin_error / End_of_data
                seen["failed"] += 1
    assert seen["ok"] > 0 and seen["failed"] > 0, seen
    assert (seen["concealed"] > 20) == (strict == 0), seen  # (without strict decoding most damaged pictures come back, concealed)


def test_strict_decoding_and_warnings(api, hm):
    """unknown VUI colour codes: a decoding warning + 'unspecified' without strict decoding, an error with it
    (HEIF_WARN_OR_FAIL, heif_plugin.h:290-301; decoder_libde265.cc:339-357; heif.cc:1223-1245, 1811-1905)"""
    import synthutil
    pic = synthutil.picture(77, width=64, height=64, vui=1, matrix=3, primaries=3, full_range=1)  # 3 = reserved code points
    data = heifwriter.write_heic([pic], (64, 64))
    ctx, h, img, e = _decode(api, data, 0, 0, 99)  # native planar
    assert e.code == 0, e.message
    assert api.heif_image_get_decoding_warnings(img, 0, None, 0) == 2
    w = (Err * 4)()
    assert api.heif_image_get_decoding_warnings(img, 0, w, 4) == 2
    assert [(x.code, x.subcode) for x in w[:2]] == [(2, 133), (2, 135)]  # Invalid_input / Unknown_NCLX_color_primaries, _matrix_coefficients
    api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx)
    opt = api.heif_decoding_options_alloc()
    C.cast(opt, C.POINTER(_DecodingOptions)).contents.strict_decoding = 1
    ctx, h, img, e = _decode(api, data, 0, 0, 99, options=opt)
    assert e.code == 2 and not img, (e.code, e.message)
    api.heif_image_handle_release(h); api.heif_context_free(ctx); api.heif_decoding_options_free(opt)
    # a well-formed file has no warnings
    ctx, h, img, e = _decode(api, open(os.path.join(HERE, "data", "colors-no-alpha.heic"), "rb").read(), 0, 0, 99)
    assert e.code == 0 and api.heif_image_get_decoding_warnings(img, 0, None, 0) == 0
    api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx)


class _ColourConvOptions(C.Structure):  # heif.h:1546-1562
    _fields_ = [("version", C.c_uint8), ("preferred_chroma_downsampling_algorithm", C.c_int),
                ("preferred_chroma_upsampling_algorithm", C.c_int), ("only_use_preferred_chroma_algorithm", C.c_uint8)]


class _DecodingOptions(C.Structure):  # heif.h:1565-1611 (fork layout with the trailing ext_dst fields)
    _fields_ = [("version", C.c_uint8), ("ignore_transformations", C.c_uint8), ("start_progress", C.c_void_p),
                ("on_progress", C.c_void_p), ("end_progress", C.c_void_p), ("progress_user_data", C.c_void_p),
                ("convert_hdr_to_8bit", C.c_uint8), ("strict_decoding", C.c_uint8), ("decoder_id", C.c_char_p),
                ("color_conversion_options", _ColourConvOptions), ("ext_dst_enable", C.c_bool), ("ext_dst", C.c_void_p),
                ("ext_dst_len", C.c_uint32), ("ext_dst_stride", C.c_uint32)]


def test_forced_bilinear_upsampling_option(api, hm):
    """only_use_preferred_chroma_algorithm = 1 with the (default) bilinear preference switches the chain to
    Op_YCbCr420_bilinear_to_YCbCr444 -> float op; without it the nearest-neighbour ops stay."""
    case = GOLD["cases"][0]
    data = open(os.path.join(HERE, "data", case["file"]), "rb").read()
    opt = api.heif_decoding_options_alloc()
    o = C.cast(opt, C.POINTER(_DecodingOptions)).contents
    assert o.color_conversion_options.preferred_chroma_upsampling_algorithm == 2  # heif_chroma_upsampling_bilinear (heif.cc:1085)
    assert o.color_conversion_options.only_use_preferred_chroma_algorithm == 0
    o.color_conversion_options.only_use_preferred_chroma_algorithm = 1
    ctx, h, img, e = _decode(api, data, case["item"], 1, 10, options=opt)
    assert e.code == 0, e.message
    stride = C.c_int()
    p = api.heif_image_get_plane_readonly(img, 10, C.byref(stride))
    got = np.ascontiguousarray(np.ctypeslib.as_array(p, shape=(case["h"], stride.value)))
    f = pipeline.HeifFile(hm, data)
    hevc = f.hevc_data(case["item"] or f.primary())
    f.close()
    exp, es, _ = pipeline.cpu_decode(hm, [hevc], case["w"], case["h"], case["w"], case["h"], 1, False, 10, bilinear=True)
    assert es == stride.value
    np.testing.assert_array_equal(got[:, :case["w"] * 3], exp[:case["h"], :case["w"] * 3])
    api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx); api.heif_decoding_options_free(opt)


def test_hdr_image_to_8bit_targets_and_convert_hdr_to_8bit(api, hm):
    """heif_decode_image of a 10-bit image: RGB24 / RGBA32 targets are 8 bit whatever the image holds
    (colorconversion.cc:566-573), RRGGBBAA carries (1 << 10) - 1 as alpha; convert_hdr_to_8bit (heif.h:1585, context.cc:1550)
    changes none of these targets (RRGGBB stays "> 8 bit", a native target is not converted at all) and is accepted"""
    import synthutil
    W, H = 128, 72
    pic = synthutil.picture(9100, width=W, height=H, bit_depth=10, chroma_format=2, vui=1, full_range=0, matrix=9, primaries=9)
    data = heifwriter.write_heic([pic], (W, H), chroma_format=2, bit_depth=10)
    for hdr_flag in (0, 1):
        opt = api.heif_decoding_options_alloc()
        C.cast(opt, C.POINTER(_DecodingOptions)).contents.convert_hdr_to_8bit = hdr_flag
        for chroma, bpp, bits in ((10, 3, 8), (11, 4, 8), (14, 6, 10), (13, 8, 10), (15, 8, 10)):
            ctx, h, img, e = _decode(api, data, 0, 1, chroma, options=opt)
            assert e.code == 0, (chroma, e.message)
            stride = C.c_int()
            p = api.heif_image_get_plane_readonly(img, 10, C.byref(stride))
            got = np.ascontiguousarray(np.ctypeslib.as_array(p, shape=(H, stride.value)))
            exp, es, _ = pipeline.cpu_decode(hm, [pic], W, H, W, H, 1, False, chroma)
            assert es == stride.value and api.heif_image_get_bits_per_pixel_range(img, 10) == bits
            np.testing.assert_array_equal(got[:, :W * bpp], exp[:H, :W * bpp])
            api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx)
        # native target: no conversion runs, the planes stay 10 bit
        ctx, h, img, e = _decode(api, data, 0, 0, 99, options=opt)
        assert e.code == 0 and api.heif_image_get_bits_per_pixel_range(img, 0) == 10
        api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx)
        api.heif_decoding_options_free(opt)


def test_icc_profile_on_handles_and_images(api, hm):
    """heif_image_handle_get_raw_color_profile / heif_image_get_raw_color_profile (heif.cc:1768-1793, 1931-2003): the
    item's ICC profile on the handle and on the decoded image (converted or not); a grid handle inherits its tile's, the
    decoded grid carries none"""
    import synthutil
    profile = bytes((7 * i) & 0xFF for i in range(1000))
    tiles = [synthutil.picture(90 + i, width=64, height=64) for i in range(2)]
    single = heifwriter.write_heic(tiles[:1], (64, 64), icc=(b"prof", profile))
    for colorspace, chroma in ((0, 99), (1, 10)):  # native planar, RGB24
        ctx, h, img, e = _decode(api, single, 0, colorspace, chroma)
        assert e.code == 0, e.message
        assert api.heif_image_handle_get_color_profile_type(h) == 0x70726F66 and api.heif_image_handle_get_raw_color_profile_size(h) == len(profile)
        buf = C.create_string_buffer(len(profile))
        assert api.heif_image_handle_get_raw_color_profile(h, buf).code == 0 and buf.raw == profile
        assert api.heif_image_get_color_profile_type(img) == 0x70726F66 and api.heif_image_get_raw_color_profile_size(img) == len(profile)
        buf2 = C.create_string_buffer(len(profile))
        assert api.heif_image_get_raw_color_profile(img, buf2).code == 0 and buf2.raw == profile
        api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx)
    grid = heifwriter.write_heic(tiles, (64, 64), grid=(1, 2, 128, 64), icc=(b"rICC", profile))
    ctx, h, img, e = _decode(api, grid, 0, 1, 10)
    assert e.code == 0, e.message
    assert api.heif_image_handle_get_color_profile_type(h) == 0x72494343
    assert api.heif_image_get_raw_color_profile_size(img) == 0 and api.heif_image_get_color_profile_type(img) == 0x6E636C78  # the converted canvas: nclx only
    api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx)


def test_range_extension_images_through_heif_decode_image(api, hm):
    """HEIC files whose coded pictures use the range-extension tools (4:4:4 with cross-component prediction, implicit
    RDPCM + rotation + large transform-skip blocks in a 10-bit 4:2:2 grid, persistent Rice adaptation + CU chroma QP
    offsets in an 8-bit 4:2:0 grid): heif_decode_image == the CPU restatement of the whole path (entropy decode by the
    product's parser, reconstruction / filters / paste / colour by the oracle - equal to libde265 on these tools:
    test_oracle_decode.py::test_range_extension_sweep_matches_reference_decoder_live)"""
    import synthutil
    cases = [
        (dict(chroma_format=3, cross_component=1, rext_sps=1 | 2 | 4 | 32, log2_max_ts=4, tq_bypass=150, matrix=0), 3, 8, (1, 1, 128, 72), 10),
        (dict(chroma_format=3, bit_depth=10, cross_component=1, rext_sps=128, big_levels=200, matrix=1, full_range=0), 3, 10, (1, 1, 128, 72), 14),
        (dict(chroma_format=2, bit_depth=10, rext_sps=1 | 4, log2_max_ts=5, matrix=9, full_range=0, primaries=9), 2, 10, (2, 2, 250, 140), 14),
        (dict(chroma_format=1, rext_sps=128 | 2, chroma_qp_list=3, chroma_qp_depth=1, big_levels=300, matrix=6, full_range=1), 1, 8, (2, 3, 380, 140), 10),
    ]
    for k, (kw, cf, bd, (rows, cols, ow, oh), chroma) in enumerate(cases):
        W, H = 128, 72
        pics = [synthutil.picture(9300 + 10 * k + t, width=W, height=H, vui=1, **kw) for t in range(rows * cols)]
        grid = None if rows * cols == 1 else (rows, cols, ow, oh)
        data = heifwriter.write_heic(pics, (W, H), grid=grid, chroma_format=cf, bit_depth=bd)
        ctx, h, img, e = _decode(api, data, 0, 1, chroma, threads=3)
        assert e.code == 0, (k, e.message)
        stride = C.c_int()
        p = api.heif_image_get_plane_readonly(img, 10, C.byref(stride))
        got = np.ascontiguousarray(np.ctypeslib.as_array(p, shape=(oh, stride.value)))
        exp, es, _ = pipeline.cpu_decode(hm, pics, W, H, ow, oh, cols, grid is not None, chroma)
        bpp = {10: 3, 14: 6}[chroma]
        assert es == stride.value
        np.testing.assert_array_equal(got[:, :ow * bpp], exp[:oh, :ow * bpp], err_msg=f"case {k}")
        api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx)


def test_monochrome_image_to_16bit_targets(api, hm):
    """heif_decode_image of a 4:0:0 item to RRGGBB_LE / RRGGBBAA_BE: the chain through Op_mono_to_YCbCr420 (rank 3 of SURVEY 8f)"""
    import hevcutil
    import orc
    import synthutil
    W, H = 96, 72
    for bd in (8, 10):
        pic = synthutil.picture(9400 + bd, width=W, height=H, chroma_format=0, bit_depth=bd, log2_ctb=4, vui=1, full_range=0, matrix=1)
        data = heifwriter.write_heic([pic], (W, H), chroma_format=0, bit_depth=bd)
        planes, info = orc.oracle_decode(hevcutil.parse(hm, pic), 3)
        bps = 2 if bd > 8 else 1
        y = orc.alloc_plane(W, H, bps)
        y[0][:H, :W * bps] = np.ascontiguousarray(planes[0][:H, :W].astype(np.uint8 if bps == 1 else np.uint16)).view(np.uint8).reshape(H, W * bps)
        for chroma, bpp in ((14, 6), (13, 8)):
            ctx, h, img, e = _decode(api, data, 0, 1, chroma)
            assert e.code == 0, (bd, chroma, e.message)
            stride = C.c_int()
            p = api.heif_image_get_plane_readonly(img, 10, C.byref(stride))
            got = np.ascontiguousarray(np.ctypeslib.as_array(p, shape=(H, stride.value)))
            exp, es, chain = orc.convert_by_search([y, None, None], W, H, bd, 0, (1, info["matrix"], info["primaries"], info["full_range"]), chroma)
            assert chain[0] == "Op_mono_to_YCbCr420" and es == stride.value
            np.testing.assert_array_equal(got[:, :W * bpp], exp[:H, :W * bpp])
            api.heif_image_release(img); api.heif_image_handle_release(h); api.heif_context_free(ctx)
