"""CPU: host parser (product, C ABI) + oracle scalar executors reproduce the fingerprints the REAL
reference decoder (libde265 built from /root/reference by oracle/Makefile) produced for the
committed fixtures (tests/golden/decode.json, written by tools/make_fixtures.py), stage by stage."""
import json
import os

import numpy as np
import pytest

import hevcutil
import orc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "decode.json")))


def _fp(planes):
    h = 0
    for p in planes:
        a = p if p.max() > 255 else p.astype(np.uint8)
        buf = a.tobytes()
        h = orc.load().orc_fnv1a64(buf, len(buf), h)
    return f"{h:016x}"


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_matches_reference_fingerprints(hm, name):
    data = open(os.path.join(HERE, "data", name + ".hevc"), "rb").read()
    blob = hevcutil.parse(hm, data)
    for stage, bits in (("recon", 0), ("deblock", 1), ("full", 3)):
        planes, info = orc.oracle_decode(blob, bits)
        assert _fp(planes) == GOLD[name][stage], f"{name}: stage {stage}"
    assert info["width"] == GOLD[name]["width"] and info["height"] == GOLD[name]["height"]
    assert info["full_range"] == GOLD[name]["info"]["full_range"]
    assert info["matrix"] == GOLD[name]["info"]["matrix"]


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_matches_reference_decoder_live(hm, name):
    """when oracle/_ref is present (build container and GPU box) compare planes directly"""
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built")
    data = open(os.path.join(HERE, "data", name + ".hevc"), "rb").read()
    ref, _ = orc.ref_decode(data, 0)
    mine, _ = orc.oracle_decode(hevcutil.parse(hm, data), 3)
    for c in range(3):
        np.testing.assert_array_equal(mine[c], ref[c])


def test_parser_rejects_garbage(hm):
    with pytest.raises(RuntimeError):
        hevcutil.parse(hm, b"\x00\x00\x00\x05hello")
    with pytest.raises(RuntimeError):
        hevcutil.parse(hm, b"\x00\x00\x00\xffshort")


# ---- synthetic corpus (tests/corpus.py), blessed by the reference decoder -------------------
import corpus  # noqa: E402

SYNTH = json.load(open(os.path.join(HERE, "golden", "synth.json")))


@pytest.mark.parametrize("name", sorted(corpus.CASES))
def test_synth_corpus_matches_reference_fingerprints(hm, name):
    data = corpus.stream(name)
    assert f"{orc.load().orc_fnv1a64(data, len(data), 0):016x}" == SYNTH[name]["stream_fnv"], "synthesiser is not deterministic"
    blob = hevcutil.parse(hm, data)
    for stage, bits in (("recon", 0), ("deblock", 1), ("full", 3)):
        planes, info = orc.oracle_decode(blob, bits, crop=True)
        assert _fp(planes) == SYNTH[name][stage], f"{name}: stage {stage}"
    assert info["full_range"] == SYNTH[name]["info"]["full_range"]
    assert info["matrix"] == SYNTH[name]["info"]["matrix"]
    assert info["bit_depth"] == SYNTH[name]["info"]["bit_depth"]


def test_rare_syntax_sweep_matches_reference_decoder_live(hm):
    """PCM / transquant-bypass / scaling-list streams: host parser + oracle == libde265 at every stage (needs oracle/_ref)"""
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built")
    import synthutil
    for seed, kw in corpus.rare_syntax_sweep(40):
        data = synthutil.picture(seed, **kw)
        blob = hevcutil.parse(hm, data)
        for stage, rf, bits in (("recon", orc.REF_F_NO_DEBLOCK | orc.REF_F_NO_SAO, 0), ("deblock", orc.REF_F_NO_SAO, 1), ("full", 0, 3)):
            ref, _ = orc.ref_decode(data, rf)
            mine, _ = orc.oracle_decode(blob, bits)
            for c in range(len(ref)):
                assert np.array_equal(mine[c], ref[c]), f"seed {seed} {kw}: stage {stage} plane {c}"


def test_structure_sweep_matches_reference_decoder_live(hm):
    """several slices, dependent slice segments, tiles, WPP, loop filters stopped at slice / tile borders, per-slice
    headers, conformance windows: host parser + oracle == libde265 at every stage (needs oracle/_ref)"""
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built")
    import synthutil
    multi = 0
    for seed, kw in corpus.structure_sweep(100):
        data = synthutil.picture(seed, **kw)
        blob = hevcutil.parse(hm, data)
        multi += int.from_bytes(blob[0x2C:0x30], "little") > 1
        for stage, rf, bits in (("recon", orc.REF_F_NO_DEBLOCK | orc.REF_F_NO_SAO, 0), ("deblock", orc.REF_F_NO_SAO, 1), ("full", 0, 3)):
            ref, _ = orc.ref_decode(data, rf)
            mine, _ = orc.oracle_decode(blob, bits, crop=True)
            for c in range(len(ref)):
                assert np.array_equal(mine[c], ref[c]), f"seed {seed} {kw}: stage {stage} plane {c}"
    assert multi > 40  # the sweep really holds multi-slice pictures


def test_range_extension_sweep_matches_reference_decoder_live(hm):
    """range-extension tools (transform-skip rotation / context / block sizes, implicit RDPCM, intra smoothing off,
    persistent Rice adaptation, CU chroma QP offsets, cross-component prediction, SAO offset scaling, and the flags the
    reference reads and ignores): host parser + oracle == libde265 at every stage (needs oracle/_ref)"""
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built")
    import synthutil
    used = {"ccp": 0, "rot": 0, "rdpcm": 0}
    for seed, kw in corpus.rext_sweep(160) + corpus.rext_large():
        data = synthutil.picture(seed, **kw)
        blob = hevcutil.parse(hm, data)
        flags = int.from_bytes(blob[36:40], "little")
        used["ccp"] += bool(flags & 0x10000); used["rot"] += bool(flags & 0x2000); used["rdpcm"] += bool(flags & 0x4000)
        for stage, rf, bits in (("recon", orc.REF_F_NO_DEBLOCK | orc.REF_F_NO_SAO, 0), ("deblock", orc.REF_F_NO_SAO, 1), ("full", 0, 3)):
            ref, _ = orc.ref_decode(data, rf)
            mine, _ = orc.oracle_decode(blob, bits, crop=True)
            for c in range(len(ref)):
                assert np.array_equal(mine[c], ref[c]), f"seed {seed} {kw}: stage {stage} plane {c}"
    assert min(used.values()) > 20, used


def test_range_extension_refusals(hm):
    """combinations whose result in the reference is undefined or mis-sized are refused, not guessed"""
    import synthutil
    with pytest.raises(RuntimeError, match="outside 4:4:4"):
        hevcutil.parse(hm, synthutil.picture(1, cross_component=1, chroma_format=1))
    # (the synthesiser drives the product's own syntax walker, hevc_syntax.h, which raises the refusal while encoding)
    with pytest.raises(RuntimeError, match="synth failed: -3"):
        synthutil.picture(2, width=128, height=128, rext_sps=128, slices=200, dependent=1000)
    assert len(synthutil.picture(2, width=128, height=128, rext_sps=128, slices=200, dependent=0)) > 100


def test_row_parallel_parse_equals_serial(pkg, hm):
    """hm_hevc_parse_mt (WPP rows entropy-decoded in parallel, decctx.cc:1004-1116 of the reference) must produce the
    serial parser's command stream byte for byte: the three real 1080p WPP streams, the corpus, and a sweep of WPP
    pictures with slices / dependent segments / PCM (segments that do not qualify run serially inside the same call)."""
    import os
    capi = pkg.capi
    here = os.path.dirname(__file__)
    n_parallel = 0
    for name in ("basketball_1080p_qp32", "basketball_1080p_qp25", "basketball_1080p_qp1"):
        data = open(os.path.join(here, "data", name + ".hevc"), "rb").read()
        assert capi.parse_hevc(data, threads=6) == capi.parse_hevc(data), name
        n_parallel += 1
    import synthutil
    for name in corpus.CASES:
        data = corpus.stream(name)
        assert capi.parse_hevc(data, threads=4) == capi.parse_hevc(data), name
    for seed, kw in corpus.structure_sweep(60, first_seed=9000):
        kw = dict(kw, wpp=1, tile_cols=1, tile_rows=1)
        data = synthutil.picture(seed, **kw)

        def outcome(threads):
            try:
                return capi.parse_hevc(data, threads=threads)
            except capi.HmError as e:  # (a forced combination the parser refuses like the reference: the same refusal)
                return str(e)
        assert outcome(3 + seed % 4) == outcome(1), (seed, kw)
    # errors come out of the serial pass: same status as the serial parser
    data = bytearray(open(os.path.join(here, "data", "basketball_1080p_qp32.hevc"), "rb").read())
    data[len(data) // 2] ^= 0x5A
    for threads in (1, 4):
        try:
            a = capi.parse_hevc(bytes(data), threads=threads)
        except capi.HmError as e:
            a = str(e)
        if threads == 1:
            ref = a
        else:
            assert a == ref


def test_tile_parallel_parse_equals_serial(pkg, hm):
    """slice segments with an entry point per HEVC tile (no WPP): the rows of tiles are entropy-decoded side by side
    (hevc_parse.cpp: parse_tiles_parallel) and must give the serial parser's command stream byte for byte - in both record
    orders (split chains share a CTB row's record lists between the tiles of a row of tiles; decode order rebases the level
    indices) -, and the parallel path must actually have been taken"""
    import ctypes as C
    import synthutil
    capi = pkg.capi
    hm.hm_parse_parallel_segments.restype = C.c_long
    hm.hm_parse_parallel_segments.argtypes = [C.c_int]
    before = hm.hm_parse_parallel_segments(1)
    shapes = [dict(width=256, height=192, tile_cols=3, tile_rows=2), dict(width=320, height=256, tile_cols=4, tile_rows=3, tiles_uniform=0),
              dict(width=256, height=256, log2_ctb=4, tile_cols=2, tile_rows=4, bit_depth=10), dict(width=384, height=256, log2_ctb=6, tile_cols=2, tile_rows=2, chroma_format=2, bit_depth=10),
              dict(width=1024, height=1024, tile_cols=2, tile_rows=3, density=20), dict(width=256, height=192, tile_cols=1, tile_rows=3, chroma_format=0),
              dict(width=256, height=192, tile_cols=3, tile_rows=2, pcm=200, tq_bypass=100), dict(width=256, height=192, tile_cols=2, tile_rows=2, lf_across_tiles=0, slices=0)]
    for i, kw in enumerate(shapes):
        for seed in (8100 + i, 8200 + i):
            data = synthutil.picture(seed, **kw)
            for order in (0, 1, 2):  # HM_RECORDS_AUTO / SPLIT / DECODE_ORDER
                assert capi.parse_hevc(data, threads=3, record_order=order) == capi.parse_hevc(data, record_order=order), (seed, kw, order)
    assert hm.hm_parse_parallel_segments(1) >= before + len(shapes) * 2 * 3
    # several slices, each a whole number of tiles or not: whatever does not qualify runs serially inside the same call
    for seed, kw in corpus.structure_sweep(40, first_seed=9500):
        kw = dict(kw, wpp=0, tile_cols=2, tile_rows=3)
        data = synthutil.picture(seed, **kw)

        def outcome(threads):
            try:
                return capi.parse_hevc(data, threads=threads)
            except capi.HmError as e:
                return str(e)
        assert outcome(4) == outcome(1), (seed, kw)


def test_record_order_follows_the_picture_class(hm):
    """which reconstruction kernels a picture is written for (hevc_syntax.h: quad_class): split chains (HM_PIC_SPLIT_CHAINS:
    k_residual + k_chain) for every class since r04 - 8 to 12 bit, CTBs of 16 / 32 / 64, small and large pictures -; records in
    decode order (k_recon) only for rare syntax and 4:4:4"""
    import synthutil
    SPLIT = 0x1000

    def split(**kw):
        blob = hevcutil.parse(hm, synthutil.picture(77, **kw))
        return bool(int.from_bytes(blob[36:40], "little") & SPLIT)
    assert split(width=128, height=128, log2_ctb=5) and split(width=128, height=128, log2_ctb=6)
    assert split(width=128, height=128, log2_ctb=4)
    assert split(width=128, height=128, log2_ctb=5, bit_depth=10) and split(width=128, height=128, log2_ctb=6, bit_depth=12, chroma_format=2)
    assert split(width=1024, height=1024, log2_ctb=5, bit_depth=10, density=5) and split(width=1024, height=1024, log2_ctb=4, density=5)
    assert not split(width=1024, height=1024, log2_ctb=5, density=5, scaling_list=1) and not split(width=128, height=128, chroma_format=3)


def test_q9_class_equals_the_reference_scalar_build(hm):
    """Quirk Q9: 8-bit pictures with CTBs of 16 and sub-sampled chroma, SAO on.  The fork's AVX2 SAO kernels write
    8 columns past a chroma CTB (x86_new/x86_sao.cc:271,320-369); its scalar / SSE4 code does not.  Parser + oracle must equal the reference's
    SCALAR build on this class, sample for sample (tools/q9_count.py counts how often the default build differs: profiles/r06_q9_count.txt)."""
    import random
    import synthutil
    if not orc.have_ref():
        pytest.skip("reference decoder not built")
    rng = random.Random(99)
    differ_from_default = 0
    for i in range(12):
        cf = 1 + (i & 1)
        data = synthutil.picture(881000 + i, width=8 * rng.randrange(4, 30), height=8 * rng.randrange(4, 24), chroma_format=cf, bit_depth=8, log2_ctb=4,
                                 qp=rng.randrange(18, 42), sao=1, density=rng.randrange(30, 90), cu_qp_delta=rng.randrange(2))
        mine, _ = orc.oracle_decode(hevcutil.parse(hm, data), 3, crop=True)
        scalar, _ = orc.ref_decode(data, orc.REF_F_SCALAR)
        assert len(mine) == len(scalar) and all(np.array_equal(m, r) for m, r in zip(mine, scalar)), f"picture {i}: differs from the reference's scalar build"
        default, _ = orc.ref_decode(data, 0)
        assert np.array_equal(default[0], scalar[0])  # (the quirk only ever touches chroma)
        differ_from_default += any(not np.array_equal(a, b) for a, b in zip(default[1:], scalar[1:]))
    print(f"Q9: the default build differs from the scalar build in {differ_from_default} of 12 pictures on this host")
