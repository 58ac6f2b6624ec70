"""GPU box: 500 seeded random parameter combinations (rare syntax + ordinary syntax over wide ranges): HIP == oracle at the
three stages.  python tools/big_sweep.py  (r01: 500 cases, 0 mismatches)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, random
import corpus, synthutil, orc, gpudecode
import __graft_entry__ as g
pkg = g.load_package(test_knobs=True)
rng = random.Random(20261002)
cases = corpus.rare_syntax_sweep(240, first_seed=4000)
for i in range(260):   # ordinary syntax, wide parameter ranges
    cf = rng.choice([0, 1, 1, 1, 2, 3]); bd = rng.choice([8, 8, 8, 10, 12, 9, 11]); l2 = rng.choice([4, 5, 5, 6])
    if l2 == 4 and bd == 8 and cf in (1, 2): l2 = 5
    kw = dict(width=8 * rng.randrange(1, 40), height=8 * rng.randrange(1, 30), chroma_format=cf, bit_depth=bd, log2_ctb=l2,
              qp=rng.randrange(10, 48), cu_qp_delta=rng.randrange(2), sao=rng.randrange(2), deblock_disable=int(rng.random() < 0.15),
              sign_hiding=rng.randrange(2), transform_skip=rng.randrange(2), strong_intra=rng.randrange(2), cb_qp_offset=rng.randrange(-6, 7),
              cr_qp_offset=rng.randrange(-6, 7), beta_offset_div2=rng.randrange(-4, 5), tc_offset_div2=rng.randrange(-4, 5),
              density=rng.randrange(20, 100), wpp=rng.randrange(2), log2_min_cb=rng.choice([3, 3, 4]) if l2 > 4 else 3)
    if kw["log2_min_cb"] == 4: kw.update(log2_min_tb=rng.choice([2, 3]), width=(kw["width"] + 15) // 16 * 16, height=(kw["height"] + 15) // 16 * 16)
    cases.append((7000 + i, kw))
blobs = []
kept = []
for seed, kw in cases:
    try:
        blobs.append(pkg.capi.parse_hevc(synthutil.picture(seed, **kw))); kept.append((seed, kw))
    except Exception as e:
        print("skip", seed, kw, str(e)[:80])
bad = 0
for bits in (0, 1, 3):
    for s in range(0, len(blobs), 125):
        got = gpudecode.decode_pictures(pkg, blobs[s:s + 125], bits)
        for (seed, kw), blob, gp in zip(kept[s:s + 125], blobs[s:s + 125], got):
            exp, _ = orc.oracle_decode(blob, bits)
            for c in range(len(exp)):
                if not np.array_equal(gp[c], exp[c]):
                    bad += 1; print("MISMATCH", seed, kw, "stages", bits, "plane", c, int((gp[c] != exp[c]).sum())); break
print("cases", len(kept), "bad", bad)
