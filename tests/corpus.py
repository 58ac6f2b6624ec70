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

# 8-bit pictures in which the reference takes its "pcmf" deblocking branch: its SIMD build (the configuration of
# oracle/_ref, and what x86 / ARM users run) filters luma edges between ordinary units with the SSE / NEON kernel, its
# scalar build leaves them unfiltered (fallback-postfilter.h:85-124 reads the flags with the opposite polarity).  The
# fixtures and the product follow the SIMD build; tools/make_fixtures.py does not require the scalar build to agree.
SIMD_BUILD_ONLY = {"pcm_nofilter", "tq_bypass", "yuv444_rare"}


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
