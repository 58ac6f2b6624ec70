"""Pin the image-level flow (HEIF parsing -> tile decode -> crop -> paste/rescale -> colour) against the
fingerprints of the REAL reference recorded in BASELINE.md (tests/golden/heic.json).
CPU part: product host parsing + oracle pieces.  GPU part (-m gpu): hm_decode_item through the C ABI."""
import json
import os

import numpy as np
import pytest

import heifwriter
import pipeline

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "heic.json")))


def _load(name):
    return open(os.path.join(HERE, "data", name), "rb").read()


def _grid_file(hm):
    f = pipeline.HeifFile(hm, _load("example.heic"))
    tiles = [f.hevc_data(i) for i in GOLD["grid_1x2"]["tiles"]]
    f.close()
    return heifwriter.write_heic(tiles, (1280, 854), grid=(1, 2, 2560, 854)), tiles


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f"{c['file']}-{c['item']}-{c['fmt']}")
def test_cpu_flow_matches_reference_fingerprint(hm, case):
    f = pipeline.HeifFile(hm, _load(case["file"]))
    iid = case["item"] or f.primary()
    info = f.info(iid)
    assert (info.width, info.height) == (case["w"], case["h"])
    out, stride, _ = pipeline.cpu_decode(hm, [f.hevc_data(iid)], case["w"], case["h"], case["w"], case["h"], 1, False, case["fmt"])
    assert bool(info.has_alpha) == (f.alpha_item(iid) != 0)
    if case.get("alpha"):  # RGBA of an image with an alpha auxiliary image (fuzz-corpus colors-with-alpha*)
        aid = f.alpha_item(iid)
        ai = f.info(aid)
        pipeline.attach_alpha(hm, out, stride, case["w"], case["h"], f.hevc_data(aid), ai.width, ai.height)
    f.close()
    bpp = {10: 3, 11: 4, 12: 6, 14: 6}[case["fmt"]]
    if "stride" in case:
        assert stride == case["stride"]
    if "first" in case:
        assert out[0, :3].tolist() == case["first"]
    assert pipeline.survey_fnv(out, stride, case["w"] * bpp, case["h"]) == case["fnv"]


def test_cpu_grid_flow_matches_reference_fingerprint(hm):
    g = GOLD["grid_1x2"]
    data, tiles = _grid_file(hm)
    f = pipeline.HeifFile(hm, data)
    info = f.info(f.primary())
    assert (info.is_grid, info.grid_rows, info.grid_cols, info.width, info.height) == (1, 1, 2, g["w"], g["h"])
    f.close()
    out, stride, canv = pipeline.cpu_decode(hm, tiles, 1280, 854, g["w"], g["h"], 2, True, 10)
    assert stride == g["stride"] and out[0, :3].tolist() == g["first"]
    assert pipeline.survey_fnv(out, stride, g["w"] * 3, g["h"]) == g["fnv"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f"{c['file']}-{c['item']}-{c['fmt']}")
def test_gpu_decode_item_matches_reference_fingerprint(hm, case):
    f = pipeline.HeifFile(hm, _load(case["file"]))
    iid = case["item"] or f.primary()
    planes, meta = f.decode(iid, case["fmt"])
    f.close()
    bpp = {10: 3, 11: 4, 12: 6, 14: 6}[case["fmt"]]
    assert (meta["width"], meta["height"]) == (case["w"], case["h"])
    if "stride" in case:
        assert meta["stride"][0] == case["stride"]
    assert pipeline.survey_fnv(planes[0], meta["stride"][0], case["w"] * bpp, case["h"]) == case["fnv"]


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 4])
def test_gpu_grid_matches_reference_fingerprint(hm, threads):
    g = GOLD["grid_1x2"]
    data, tiles = _grid_file(hm)
    f = pipeline.HeifFile(hm, data)
    planes, meta = f.decode(f.primary(), 10, threads=threads)
    native, nmeta = f.decode(f.primary(), 0)
    f.close()
    assert nmeta["has_nclx"] == 0  # a grid canvas carries no nclx
    # the converted image carries the conversion's output state: the default profile with the sRGB replacements
    assert (meta["has_nclx"], meta["primaries"], meta["transfer"], meta["matrix"], meta["full_range"]) == (1, 1, 13, 6, 1)
    assert pipeline.survey_fnv(planes[0], meta["stride"][0], g["w"] * 3, g["h"]) == g["fnv"]
    # native canvas planes == CPU flow (limited->full rescale of the paste, context.cc:2504-2528)
    _, _, canv = pipeline.cpu_decode(hm, tiles, 1280, 854, g["w"], g["h"], 2, True, 10)
    np.testing.assert_array_equal(native[0][:g["h"], :g["w"]], canv[0][0][:g["h"], :g["w"]])
    np.testing.assert_array_equal(native[1][:g["h"] // 2, :g["w"] // 2], canv[1][0][:g["h"] // 2, :g["w"] // 2])


@pytest.mark.gpu
def test_gpu_decode_item_forced_bilinear(hm):
    """heif_color_conversion_options.only_use_preferred_chroma_algorithm + bilinear through
    hm_decode_item: example.heic (limited-range 4:2:0 with nclx) and a grid (canvas without nclx)."""
    case = GOLD["cases"][0]
    f = pipeline.HeifFile(hm, _load(case["file"]))
    iid = case["item"] or f.primary()
    planes, meta = f.decode(iid, 10, upsampling=2)
    exp, stride, _ = pipeline.cpu_decode(hm, [f.hevc_data(iid)], case["w"], case["h"], case["w"], case["h"], 1, False, 10, bilinear=True)
    f.close()
    assert meta["stride"][0] == stride
    np.testing.assert_array_equal(planes[0][:case["h"], :case["w"] * 3], exp[:case["h"], :case["w"] * 3])
    # differs from the default (nearest-neighbour) result, i.e. the option is not ignored
    ref, _, _ = pipeline.cpu_decode(hm, [_hevc_of(hm, case)], case["w"], case["h"], case["w"], case["h"], 1, False, 10)
    assert not np.array_equal(ref[:case["h"], :case["w"] * 3], exp[:case["h"], :case["w"] * 3])

    g = GOLD["grid_1x2"]
    data, tiles = _grid_file(hm)
    f = pipeline.HeifFile(hm, data)
    planes, meta = f.decode(f.primary(), 11, upsampling=2)
    f.close()
    exp, stride, _ = pipeline.cpu_decode(hm, tiles, 1280, 854, g["w"], g["h"], 2, True, 11, bilinear=True)
    np.testing.assert_array_equal(planes[0][:g["h"], :g["w"] * 4], exp[:g["h"], :g["w"] * 4])


def _hevc_of(hm, case):
    f = pipeline.HeifFile(hm, _load(case["file"]))
    try:
        return f.hevc_data(case["item"] or f.primary())
    finally:
        f.close()


@pytest.mark.gpu
def test_concurrent_decode_calls(hm):
    """hm_decode_item from several application threads at once (libheif callers do that): shared worker crew, memory
    pools and the default stream must give every caller its own correct picture."""
    import threading
    g = GOLD["grid_1x2"]
    data, tiles = _grid_file(hm)
    exp, stride, _ = pipeline.cpu_decode(hm, tiles, 1280, 854, g["w"], g["h"], 2, True, 10)
    case = GOLD["cases"][0]
    single = _load(case["file"])
    results, errors = {}, []

    def work(k):
        try:
            for it in range(3):
                if k % 2 == 0:
                    f = pipeline.HeifFile(hm, data)
                    planes, meta = f.decode(f.primary(), 10, threads=1 + k)
                    f.close()
                    ok = np.array_equal(planes[0][:g["h"], :g["w"] * 3], exp[:g["h"], :g["w"] * 3])
                else:
                    f = pipeline.HeifFile(hm, single)
                    planes, meta = f.decode(case["item"] or f.primary(), case["fmt"], threads=2)
                    f.close()
                    bpp = {10: 3, 11: 4, 12: 6, 14: 6}[case["fmt"]]
                    ok = pipeline.survey_fnv(planes[0], meta["stride"][0], case["w"] * bpp, case["h"]) == case["fnv"]
                results[(k, it)] = ok
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(results) == 18 and all(results.values())
