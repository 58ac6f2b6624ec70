"""GPU: the pipelined end-to-end entry (hm_pipeline_*): many HEIF files in flight, results in submission order, every
result identical to the image-at-a-time path (hm_decode_item) - which the golden tests pin to the reference."""
import json
import os

import numpy as np
import pytest

import heifwriter
import pipeline
import synthutil

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "heic.json")))


def _files():
    files = []
    for i in range(24):  # grids of different shapes / contents, single images, 10-bit 4:2:2, monochrome
        kind = i % 4
        if kind == 0:
            tiles = [synthutil.picture(8000 + 10 * i + t, width=64, height=64) for t in range(6)]
            files.append(heifwriter.write_heic(tiles, (64, 64), grid=(2, 3, 180, 120)))
        elif kind == 1:
            tiles = [synthutil.picture(8000 + 10 * i + t, width=128, height=64, vui=0) for t in range(4)]
            files.append(heifwriter.write_heic(tiles, (128, 64), grid=(2, 2, 250, 128)))
        elif kind == 2:
            files.append(heifwriter.write_heic([synthutil.picture(8000 + 10 * i, width=200, height=136, slices=80, dependent=300)], (200, 136)))
        else:
            files.append(heifwriter.write_heic([synthutil.picture(8000 + 10 * i, width=96, height=72, log2_ctb=4, chroma_format=0)], (96, 72)))
    for name in ("colors-no-alpha.heic", "colors-with-alpha.heic", "example.heic"):
        files.append(open(os.path.join(HERE, "data", name), "rb").read())
    return files


@pytest.mark.parametrize("out_format,threads,depth", [(10, 4, 8), (11, 1, 2), (10, 16, 32)])
def test_pipeline_results_equal_image_at_a_time(hm, out_format, threads, depth):
    files = _files()
    expected = []
    for data in files:
        f = pipeline.HeifFile(hm, data)
        planes, meta = f.decode(f.primary(), out_format, threads=2)
        expected.append((planes[0], meta))
        f.close()
    pl = pipeline.Pipeline(hm, out_format, host_threads=threads, max_in_flight=depth)
    got = {}
    nxt = 0
    try:
        for round_ in range(2):  # the second round runs on recycled slots / pool blocks
            for i, data in enumerate(files):
                tag = 1000 * round_ + i
                while not pl.submit(data, tag):
                    t, status, arr, meta = pl.next()
                    assert status == 0, meta
                    got[t] = (arr, meta)
            while pl.pending():
                t, status, arr, meta = pl.next()
                assert status == 0, meta
                assert t not in got
                got[t] = (arr, meta)
    finally:
        pl.close()
    assert sorted(got) == [1000 * r + i for r in range(2) for i in range(len(files))]
    for t, (arr, meta) in got.items():
        exp, emeta = expected[t % 1000]
        assert (meta["width"], meta["height"], meta["stride"]) == (emeta["width"], emeta["height"], emeta["stride"][0])
        bpp = 3 if out_format == 10 else 4
        w, h = meta["width"], meta["height"]
        np.testing.assert_array_equal(arr[:h, :w * bpp], exp[:h, :w * bpp])  # (row padding is not part of the image)


def test_pipeline_order_and_reference_fingerprints(hm):
    """results come back in submission order; example.heic's items reproduce the reference fingerprints"""
    data = open(os.path.join(HERE, "data", "example.heic"), "rb").read()
    cases = [c for c in GOLD["cases"] if c["file"] == "example.heic" and c["fmt"] == 10]
    pl = pipeline.Pipeline(hm, 10, host_threads=3, max_in_flight=4)
    try:
        tags = []
        for k in range(3):
            for j, c in enumerate(cases):
                tag = 10 * k + j
                while not pl.submit(data, tag, item=c["item"]):
                    t, status, arr, meta = pl.next()
                    assert status == 0 and t == tags.pop(0)
                    cc = cases[t % 10]
                    assert pipeline.survey_fnv(arr, meta["stride"], cc["w"] * 3, cc["h"]) == cc["fnv"]
                tags.append(tag)
        while pl.pending():
            t, status, arr, meta = pl.next()
            assert status == 0 and t == tags.pop(0)
            cc = cases[t % 10]
            assert pipeline.survey_fnv(arr, meta["stride"], cc["w"] * 3, cc["h"]) == cc["fnv"]
        assert not tags
    finally:
        pl.close()


def test_pipeline_reports_bad_files_and_keeps_going(hm):
    good = heifwriter.write_heic([synthutil.picture(8800 + t, width=64, height=64) for t in range(4)], (64, 64), grid=(2, 2, 128, 128))
    # a tile without a coded picture (its slice NALs are gone): the container parses, the entropy decode of that tile fails
    # (a tile whose slice data is merely cut short is concealed since r05 - HM_PARSE_CONCEAL - and comes back as a picture: below)
    import hevcutil
    whole = synthutil.picture(8811, width=64, height=64)
    broken_tile = hevcutil.join_nals([n for n in hevcutil.split_nals(whole) if ((n[0] >> 1) & 0x3F) > 21])
    bad = heifwriter.write_heic([synthutil.picture(8810, width=64, height=64), broken_tile], (64, 64), grid=(1, 2, 128, 64))
    cut = heifwriter.write_heic([synthutil.picture(8810, width=64, height=64), whole[:len(whole) - 40]], (64, 64), grid=(1, 2, 128, 64))
    pl = pipeline.Pipeline(hm, 10, host_threads=2, max_in_flight=4)
    try:
        with pytest.raises(RuntimeError):
            pl.submit(b"\x00\x00\x00\x10ftypheic\x00\x00\x00\x00", 1)  # no meta box: refused at submit, nothing queued
        assert pl.pending() == 0
        assert pl.submit(good, 2) and pl.submit(bad, 3) and pl.submit(good, 4)
        res = [pl.next() for _ in range(3)]
        assert [r[0] for r in res] == [2, 3, 4]
        assert res[0][1] == 0 and res[2][1] == 0 and res[1][1] < 0 and "tile 1" in res[1][3]["error"]
        np.testing.assert_array_equal(res[0][2][:128, :128 * 3], res[2][2][:128, :128 * 3])
        assert pl.submit(cut, 7)
        t7 = pl.next()
        assert t7[0] == 7 and t7[1] == 0, t7[3]  # damaged slice data: a picture (its damaged part concealed)
        # closing with images still pending is fine
        assert pl.submit(good, 5) and pl.submit(good, 6)
    finally:
        pl.close()


def test_two_pipelines_with_their_own_cpu_sets(hm):
    """one pipeline per GPU, each with a CPU set of its own for its entropy-decode crew (here both on
    device 0 - the box has one GPU -, CPUs split in halves).  Results equal the image-at-a-time path; a device index
    beyond the visible devices and a CPU set outside the machine are refused."""
    import os
    data = open(os.path.join(HERE, "data", "example.heic"), "rb").read()
    f = pipeline.HeifFile(hm, data)
    exp, _ = f.decode(f.primary(), 10)
    f.close()
    cpus = sorted(os.sched_getaffinity(0))
    half = max(1, len(cpus) // 2)
    a = pipeline.Pipeline(hm, 10, host_threads=2, max_in_flight=4, device=0, cpus=(cpus[0], half))
    b = pipeline.Pipeline(hm, 10, host_threads=2, max_in_flight=4, device=0, cpus=(cpus[0] + half, max(1, len(cpus) - half)))
    for p in (a, b):
        for k in range(4):
            assert p.submit(data, k)
    for p in (a, b):
        for k in range(4):
            tag, status, arr, meta = p.next()
            assert (tag, status) == (k, 0)
            np.testing.assert_array_equal(arr, exp[0])
    a.close()
    b.close()
    with pytest.raises(RuntimeError):
        pipeline.Pipeline(hm, 10, device=64)
    with pytest.raises(RuntimeError):
        pipeline.Pipeline(hm, 10, cpus=(0, 100000))
