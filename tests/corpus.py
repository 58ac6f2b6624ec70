"""The synthetic test corpus: named (seed, parameter) cases for tests/synth.  Every case is
blessed by the real reference decoder in the build container (tools/make_fixtures.py writes its
fingerprints to tests/golden/synth.json); streams are regenerated deterministically from the
seed wherever the tests run."""

# BASELINE configs: 512x512 tiles, CTB 32, QP 27 +- cu_qp_delta, SAO, deblock, SDH
TILE = dict(width=512, height=512, log2_ctb=5, qp=27, cu_qp_delta=1, sao=1, sign_hiding=1, density=60)

CASES = {
    # config 2 tiles: VUI full_range=1 matrix=6, and the no-VUI variant (limited-range paste rescale)
    "tile512_a": dict(seed=1200000, vui=1, full_range=1, matrix=6, **TILE),
    "tile512_b": dict(seed=1200001, vui=1, full_range=1, matrix=6, **TILE),
    "tile512_novui": dict(seed=1200002, vui=0, **TILE),
    # config 4: 10-bit 4:2:2 (small variant; the full 2048x1536 one is generated in the bench / slow tests)
    "hi422_10": dict(seed=4220010, width=256, height=192, chroma_format=2, bit_depth=10, log2_ctb=5, qp=30,
                     vui=1, full_range=0, matrix=9, primaries=9),
    "hi422_10_full": dict(seed=4220011, width=192, height=128, chroma_format=2, bit_depth=10, log2_ctb=5, qp=24,
                          vui=1, full_range=1, matrix=1, primaries=1, cb_qp_offset=3, cr_qp_offset=-2),
    "hi420_10": dict(seed=4200010, width=128, height=128, chroma_format=1, bit_depth=10, log2_ctb=4, qp=33),
    "hi422_12": dict(seed=4220012, width=128, height=64, chroma_format=2, bit_depth=12, log2_ctb=5, qp=35),
    # structure / tool coverage
    "ctb64": dict(seed=64001, width=192, height=136, log2_ctb=6, qp=25),
    "ctb64_wpp": dict(seed=64002, width=256, height=192, log2_ctb=6, wpp=1, qp=31),
    "ctb16_nosao": dict(seed=16001, width=128, height=72, log2_ctb=4, sao=0, qp=29),
    "ctb32_wpp_422": dict(seed=32002, width=160, height=96, log2_ctb=5, wpp=1, chroma_format=2, qp=28),
    "no_deblock": dict(seed=7001, width=128, height=128, deblock_disable=1),
    "no_sao_no_sdh": dict(seed=7002, width=128, height=96, sao=0, sign_hiding=0, transform_skip=0, strong_intra=0),
    "offsets": dict(seed=7003, width=128, height=128, beta_offset_div2=3, tc_offset_div2=-2, cb_qp_offset=-4, cr_qp_offset=5, qp=36),
    "dense_lowqp": dict(seed=7004, width=96, height=96, density=95, qp=12),
    "sparse_highqp": dict(seed=7005, width=96, height=96, density=25, qp=47),
    "tiny": dict(seed=7006, width=8, height=8, log2_ctb=4),
    "ragged": dict(seed=7007, width=72, height=40, log2_ctb=5),
    "min_cb16": dict(seed=7008, width=128, height=64, log2_ctb=5, log2_min_cb=4, log2_min_tb=3, max_th_depth_intra=1),
    "flat_qp": dict(seed=7009, width=128, height=128, cu_qp_delta=0, max_th_depth_intra=0),
    # monochrome (4:0:0): alpha planes written by libheif-style encoders, depth / gain maps
    "mono8": dict(seed=4000001, width=160, height=104, chroma_format=0, log2_ctb=5, qp=29),
    "mono8_ctb64_wpp": dict(seed=4000002, width=192, height=136, chroma_format=0, log2_ctb=6, wpp=1, qp=24),
    "mono10": dict(seed=4000010, width=128, height=96, chroma_format=0, bit_depth=10, log2_ctb=4, qp=33),
    # scaling lists (transform.cc:507-545): default lists, random lists in the SPS, lists in the PPS overriding the SPS
    "sl_default": dict(seed=5100001, width=128, height=96, log2_ctb=5, qp=30, scaling_list=1),
    "sl_sps": dict(seed=5100002, width=192, height=128, log2_ctb=5, qp=26, scaling_list=2),
    "sl_pps_422_10": dict(seed=5100003, width=128, height=128, log2_ctb=5, chroma_format=2, bit_depth=10, qp=34, scaling_list=3),
    "sl_sps_ctb64_lowqp": dict(seed=5100004, width=192, height=136, log2_ctb=6, qp=8, density=90, scaling_list=2),
    "sl_sps_12bit_highqp": dict(seed=5100005, width=64, height=64, log2_ctb=4, bit_depth=12, qp=50, scaling_list=2),
    # PCM and transquant-bypass coding units (slice.cc:4462-4536, transform.cc:431-449) and the reference's "pcmf"
    # loop-filter branches (deblock.cc:723-786, 1635-1756; sao.cc:356-363)
    "pcm_nofilter": dict(seed=5200001, width=128, height=96, log2_ctb=5, qp=30, pcm=300, pcm_bits_y=7, pcm_bits_c=5, pcm_loop_filter_disable=1),
    "pcm_filtered": dict(seed=5200002, width=128, height=96, log2_ctb=5, qp=30, pcm=300, pcm_bits_y=8, pcm_bits_c=6, pcm_loop_filter_disable=0),
    "pcm_422_10": dict(seed=5200003, width=128, height=64, log2_ctb=5, chroma_format=2, bit_depth=10, qp=32, pcm=250, pcm_bits_y=10, pcm_bits_c=8,
                       pcm_loop_filter_disable=1, pcm_log2_max=4),
    "tq_bypass": dict(seed=5200004, width=128, height=96, log2_ctb=5, qp=28, tq_bypass=400),
    "lossless_all": dict(seed=5200005, width=96, height=64, log2_ctb=4, bit_depth=10, qp=22, tq_bypass=1000),
    "pcm_bypass_mono12": dict(seed=5200006, width=96, height=72, log2_ctb=5, chroma_format=0, bit_depth=12, qp=36, pcm=200, pcm_bits_y=11,
                              pcm_bits_c=12, pcm_loop_filter_disable=1, tq_bypass=250),
    "pcm_bypass_sl_wpp": dict(seed=5200007, width=192, height=128, log2_ctb=6, bit_depth=10, qp=29, wpp=1, scaling_list=2, pcm=200,
                              pcm_bits_y=9, pcm_bits_c=9, pcm_loop_filter_disable=0, tq_bypass=200),
    # 4:4:4 (chroma blocks as large as luma ones, chroma reference samples smoothed like luma: intrapred.cc:307-311)
    "yuv444_8": dict(seed=5300001, width=160, height=96, chroma_format=3, log2_ctb=5, qp=28, matrix=0),
    "yuv444_10_ctb64_wpp": dict(seed=5300002, width=192, height=136, chroma_format=3, bit_depth=10, log2_ctb=6, wpp=1, qp=31),
    "yuv444_12_ctb16": dict(seed=5300003, width=96, height=64, chroma_format=3, bit_depth=12, log2_ctb=4, qp=36),
    "yuv444_rare": dict(seed=5300004, width=128, height=96, chroma_format=3, log2_ctb=5, qp=30, scaling_list=2, pcm=200, pcm_bits_y=8,
                        pcm_bits_c=7, pcm_loop_filter_disable=1, tq_bypass=200),
}

# slice / tile structure (slice.cc:5004-5083, 5350-5406; deblock.cc:160-196; sao.cc:336-424): several slices, dependent
# slice segments, tiles (uniform / explicit), loop filters stopped at slice / tile borders, per-slice deblocking
# override, SAO flags, QP and chroma QP offsets, WPP together with slices / tiles, conformance windows
CASES.update({
    "slices": dict(seed=6100001, width=256, height=192, slices=60),
    "slices_dependent": dict(seed=6100002, width=256, height=192, slices=60, dependent=500),
    "slices_nolf": dict(seed=6100003, width=256, height=192, slices=80, pps_lf_across_slices_off=1),
    "slices_headers": dict(seed=6100004, width=256, height=192, slices=80, slice_lf_random=1, deblock_override=1, slice_sao_random=1,
                           slice_qp_random=1, slice_chroma_qp=1, dependent=300),
    "tiles_3x2": dict(seed=6100005, width=256, height=192, tile_cols=3, tile_rows=2),
    "tiles_3x2_nolf": dict(seed=6100006, width=256, height=192, tile_cols=3, tile_rows=2, lf_across_tiles=0),
    "tiles_explicit_slices": dict(seed=6100007, width=320, height=256, tile_cols=4, tile_rows=3, tiles_uniform=0, lf_across_tiles=0, slices=100,
                                  dependent=400, slice_lf_random=1),
    "wpp_slices_dependent": dict(seed=6100008, width=256, height=192, wpp=1, slices=100, dependent=600),
    "wpp_tiles_slices": dict(seed=6100009, width=256, height=192, wpp=1, tile_cols=2, tile_rows=2, slices=50, dependent=300, lf_across_tiles=0),
    "tiles_422_10_ctb64": dict(seed=6100010, width=256, height=192, log2_ctb=6, tile_cols=2, tile_rows=2, slices=300, chroma_format=2, bit_depth=10,
                               lf_across_tiles=0, slice_lf_random=1),
    "slices_444_pcm": dict(seed=6100011, width=192, height=128, chroma_format=3, slices=120, dependent=300, slice_lf_random=1, pcm=200,
                           pcm_loop_filter_disable=1, tq_bypass=150, slice_sao_random=1),
    "slices_mono_ctb16": dict(seed=6100012, width=160, height=96, chroma_format=0, log2_ctb=4, slices=60, dependent=400, pps_lf_across_slices_off=1),
    "conf_window": dict(seed=6100013, width=200, height=136, conf_left=2, conf_right=6, conf_top=4, conf_bottom=2),
    "conf_window_422_10": dict(seed=6100014, width=200, height=136, conf_right=8, conf_bottom=6, chroma_format=2, bit_depth=10),
})

# range-extension coding tools (sps.cc:1375-1390, pps.cc:47-142; slice.cc:3143-3177, 3425-3432, 3565-3655, 3774-3805,
# 3809-3864, 3928-3957; transform.cc:251-285, 427-466, 566-643; intrapred.cc:307-326 of the reference).  rext_sps bits:
# 1 transform_skip_rotation, 2 transform_skip_context, 4 implicit_rdpcm, 8 explicit_rdpcm, 16 extended_precision,
# 32 intra_smoothing_disabled, 64 high_precision_offsets, 128 persistent_rice, 256 cabac_bypass_alignment
CASES.update({
    "rext_ts_tools": dict(seed=6200001, width=128, height=96, rext_sps=1 | 2 | 4, log2_max_ts=4, qp=26),
    "rext_ts_bypass_422_10": dict(seed=6200002, width=128, height=96, chroma_format=2, bit_depth=10, rext_sps=1 | 2 | 4, log2_max_ts=5, tq_bypass=250, qp=30),
    "rext_nosmooth_rice": dict(seed=6200003, width=160, height=96, log2_ctb=6, rext_sps=32 | 128, big_levels=300, qp=22),
    "rext_chroma_qp_list": dict(seed=6200004, width=128, height=128, chroma_qp_list=2, chroma_qp_depth=1, cb_qp_offset=2, slices=80),
    "rext_chroma_qp_list6_422": dict(seed=6200005, width=128, height=64, chroma_format=2, bit_depth=10, chroma_qp_list=6, chroma_qp_depth=0, wpp=1),
    "rext_cross_444": dict(seed=6200006, width=128, height=96, chroma_format=3, cross_component=1, qp=27),
    "rext_cross_444_all": dict(seed=6200007, width=128, height=96, chroma_format=3, bit_depth=10, cross_component=1, rext_sps=1 | 2 | 4 | 32 | 128,
                               log2_max_ts=5, tq_bypass=150, chroma_qp_list=3, big_levels=200, qp=24, wpp=1),
    "rext_ignored_flags": dict(seed=6200008, width=96, height=64, rext_sps=8 | 16 | 64 | 256, qp=30),
    "rext_sao_scale_12": dict(seed=6200009, width=96, height=64, bit_depth=12, sao_scale_y=2, sao_scale_c=1, qp=34),
    "rext_mono_rice_rdpcm": dict(seed=6200010, width=96, height=72, chroma_format=0, bit_depth=12, rext_sps=4 | 128 | 1, big_levels=250, tq_bypass=200, qp=33),
})

# 8-bit pictures in which the reference takes its "pcmf" deblocking branch: its SIMD build (the configuration of
# oracle/_ref, and what x86 / ARM users run) filters luma edges between ordinary units with the SSE / NEON kernel, its
# scalar build leaves them unfiltered (fallback-postfilter.h:85-124 reads the flags with the opposite polarity).  The
# fixtures and the product follow the SIMD build; tools/make_fixtures.py does not require the scalar build to agree.
SIMD_BUILD_ONLY = {"pcm_nofilter", "tq_bypass", "yuv444_rare", "slices_444_pcm"}


def stream(name):
    import synthutil
    kw = dict(CASES[name])
    seed = kw.pop("seed")
    return synthutil.picture(seed, **kw)


def rare_syntax_sweep(n, first_seed=2000):
    """(seed, parameters) of a seeded sweep over the rarely used syntax: PCM and transquant-bypass units with every
    loop-filter flag combination, scaling lists, WPP, 4:0:0 / 4:2:0 / 4:2:2 / 4:4:4, 8-12 bit, every CTB size."""
    out = []
    for seed in range(first_seed, first_seed + n):
        kw = dict(width=[64, 96, 72, 128][seed % 4], height=[64, 40, 72][seed % 3], log2_ctb=[5, 4, 6, 5][seed % 4] if seed % 5 else 5,
                  chroma_format=[1, 2, 3, 0][seed % 4], bit_depth=[8, 10, 8, 12, 9][seed % 5], pcm=[200, 0, 300][seed % 3],
                  tq_bypass=[0, 300, 150, 1][seed % 4], pcm_loop_filter_disable=seed % 2, wpp=int(seed % 7 == 0), cu_qp_delta=1,
                  scaling_list=[0, 0, 2][seed % 3])
        if kw["log2_ctb"] == 4 and kw["bit_depth"] == 8 and kw["chroma_format"] in (1, 2):
            kw["log2_ctb"] = 5  # 8-bit SAO on 8-sample-wide chroma CTBs: the reference's SIMD quirk Q9, not a corpus subject
        kw["pcm_bits_y"] = max(1, kw["bit_depth"] - seed % 3)
        kw["pcm_bits_c"] = max(1, kw["bit_depth"] - seed % 4)
        out.append((seed, kw))
    return out


def structure_sweep(n, first_seed=7000):
    """(seed, parameters) of a seeded sweep over slice / tile structures: slices, dependent segments, uniform / explicit
    tiles, loop filters stopped at slice / tile borders, per-slice headers, WPP, all chroma formats, CTB sizes, depths."""
    out = []
    for seed in range(first_seed, first_seed + n):
        r = seed * 2654435761 % (1 << 32)
        pick = lambda k, opts: opts[(r >> k) % len(opts)]
        kw = dict(width=pick(0, [128, 192, 256, 160]), height=pick(2, [128, 96, 192, 64]), log2_ctb=pick(4, [5, 5, 4, 6]),
                  chroma_format=pick(6, [1, 1, 2, 3, 0]), bit_depth=pick(9, [8, 8, 10, 12]),
                  slices=pick(11, [0, 40, 100, 300]), dependent=pick(13, [0, 300, 1000]),
                  tile_cols=pick(15, [1, 1, 2, 3]), tile_rows=pick(17, [1, 2, 3]), tiles_uniform=pick(19, [1, 0]),
                  lf_across_tiles=pick(20, [1, 0]), pps_lf_across_slices_off=pick(21, [0, 0, 1]), slice_lf_random=pick(23, [0, 1]),
                  deblock_override=pick(24, [0, 1]), slice_sao_random=pick(25, [0, 1]), slice_qp_random=pick(26, [0, 1]),
                  slice_chroma_qp=pick(27, [0, 1]), wpp=pick(28, [0, 0, 1]), cu_qp_delta=1, diff_cu_qp_delta_depth=pick(30, [1, 0, 2]))
        if seed % 9 == 0:
            kw.update(pcm=200, pcm_loop_filter_disable=seed % 2, tq_bypass=150)
            kw["pcm_bits_y"] = kw["pcm_bits_c"] = kw["bit_depth"]
        if kw["log2_ctb"] == 4 and kw["bit_depth"] == 8 and kw["chroma_format"] in (1, 2):
            kw["log2_ctb"] = 5  # quirk Q9 (see rare_syntax_sweep)
        if kw["wpp"] and (kw["tile_cols"] > 1 or kw["tile_rows"] > 1):
            # WPP together with tiles: the reference accepts at most one entry point per tile and per remaining CTB row
            # (slice.cc:813-829) and takes its row tables from picture column 1 - only short slices in full-width tiles
            kw["tile_cols"] = 1
            kw["slices"] = max(kw["slices"], 300)
        kw["diff_cu_qp_delta_depth"] = min(kw["diff_cu_qp_delta_depth"], kw["log2_ctb"] - 3)
        out.append((seed, kw))
    return out


def rext_sweep(n, first_seed=11000):
    """(seed, parameters) of a seeded sweep over the range-extension tools, alone and combined with each other and with
    transquant bypass, scaling lists, WPP, slices, tiles, every chroma format / bit depth / CTB size."""
    out = []
    for seed in range(first_seed, first_seed + n):
        r = seed * 2654435761 % (1 << 32)
        pick = lambda k, opts: opts[(r >> k) % len(opts)]
        kw = dict(width=pick(0, [64, 96, 128, 72]), height=pick(2, [64, 40, 96, 72]), log2_ctb=pick(4, [5, 5, 4, 6]),
                  chroma_format=pick(6, [1, 3, 2, 3, 0]), bit_depth=pick(9, [8, 8, 10, 12]), cu_qp_delta=1,
                  rext_sps=pick(11, [0, 1, 2, 4, 32, 128, 7, 135, 167, 511, 39, 5]), log2_max_ts=pick(15, [0, 0, 3, 4, 5]),
                  tq_bypass=pick(18, [0, 0, 200, 500]), chroma_qp_list=pick(20, [0, 0, 1, 2, 6]), chroma_qp_depth=pick(23, [0, 1, 2]),
                  big_levels=pick(25, [0, 200, 400]), wpp=pick(27, [0, 0, 1]), slices=pick(29, [0, 0, 100]),
                  scaling_list=pick(30, [0, 0, 0, 2]), qp=pick(12, [27, 22, 33, 38]))
        kw["cross_component"] = int(kw["chroma_format"] == 3 and seed % 3 != 0)
        kw["chroma_qp_depth"] = min(kw["chroma_qp_depth"], kw["log2_ctb"] - 3)
        if kw["chroma_format"] == 0:
            kw["chroma_qp_list"] = 0
        if seed % 11 == 0:
            kw.update(tile_cols=2, tile_rows=2, wpp=0)
        if seed % 13 == 0 and not (kw["rext_sps"] & 128):
            kw.update(dependent=500, slices=max(kw["slices"], 100))  # (persistent_rice + dependent segments: refused, see below)
        if kw["bit_depth"] == 12 and seed % 2:
            kw.update(sao_scale_y=seed % 3, sao_scale_c=(seed // 3) % 3)
        if kw["log2_ctb"] == 4 and kw["bit_depth"] == 8 and kw["chroma_format"] in (1, 2):
            kw["log2_ctb"] = 5  # quirk Q9 (see rare_syntax_sweep)
        if kw["tq_bypass"] and kw["bit_depth"] == 8:
            pass  # (the "pcmf" deblocking branch: the SIMD build of the reference is the oracle, as for SIMD_BUILD_ONLY)
        out.append((seed, kw))
    return out


def rext_large():
    """larger pictures with 32x32 transform-skip / bypass blocks and CTB 64: RDPCM runs of 32 samples, cross-component
    prediction of 32x32 blocks"""
    return [(12000 + i, dict(width=192, height=128, log2_ctb=[6, 5][i % 2], chroma_format=[3, 1, 3, 2][i % 4], bit_depth=[8, 10, 12][i % 3],
                             rext_sps=[7, 135, 167, 39][i % 4], log2_max_ts=5, tq_bypass=[0, 300][i % 2], cross_component=int(i % 4 in (0, 2)),
                             chroma_qp_list=[0, 3][i % 2], big_levels=200, wpp=i % 2, qp=[24, 30, 36][i % 3], max_th_depth_intra=[0, 1, 2][i % 3]))
            for i in range(12)]
