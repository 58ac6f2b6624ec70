"""Build container (needs oracle/_ref): the same 500 combinations, host parser + oracle == the reference libde265 at the
three stages.  python tools/big_sweep_cpu.py  (r01: 500 cases, 0 mismatches)"""
import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, corpus, synthutil, orc, hevcutil
import __graft_entry__ as g
hm = g.load_package().lib()
rng = random.Random(20261002)
cases = corpus.rare_syntax_sweep(240, first_seed=4000)
for i in range(260):
    cf = rng.choice([0, 1, 1, 1, 2, 3]); bd = rng.choice([8, 8, 8, 10, 12, 9, 11]); l2 = rng.choice([4, 5, 5, 6])
    if l2 == 4 and bd == 8 and cf in (1, 2): l2 = 5
    kw = dict(width=8 * rng.randrange(1, 40), height=8 * rng.randrange(1, 30), chroma_format=cf, bit_depth=bd, log2_ctb=l2,
              qp=rng.randrange(10, 48), cu_qp_delta=rng.randrange(2), sao=rng.randrange(2), deblock_disable=int(rng.random() < 0.15),
              sign_hiding=rng.randrange(2), transform_skip=rng.randrange(2), strong_intra=rng.randrange(2), cb_qp_offset=rng.randrange(-6, 7),
              cr_qp_offset=rng.randrange(-6, 7), beta_offset_div2=rng.randrange(-4, 5), tc_offset_div2=rng.randrange(-4, 5),
              density=rng.randrange(20, 100), wpp=rng.randrange(2), log2_min_cb=rng.choice([3, 3, 4]) if l2 > 4 else 3)
    if kw["log2_min_cb"] == 4: kw.update(log2_min_tb=rng.choice([2, 3]), width=(kw["width"] + 15) // 16 * 16, height=(kw["height"] + 15) // 16 * 16)
    cases.append((7000 + i, kw))
bad = skipped = 0
for seed, kw in cases:
    try:
        data = synthutil.picture(seed, **kw)
        blob = hevcutil.parse(hm, data)
    except Exception as e:
        skipped += 1; continue
    for stage, rf, bits in (("recon", orc.REF_F_NO_DEBLOCK | orc.REF_F_NO_SAO, 0), ("deblock", orc.REF_F_NO_SAO, 1), ("full", 0, 3)):
        try:
            ref, _ = orc.ref_decode(data, rf)
        except RuntimeError as e:
            print("REF FAIL", seed, kw, e); bad += 1; break
        mine, _ = orc.oracle_decode(blob, bits)
        if not all(np.array_equal(a, b) for a, b in zip(ref, mine)):
            bad += 1; print("MISMATCH", seed, kw, stage, [int((a != b).sum()) for a, b in zip(ref, mine)]); break
print("cases", len(cases), "skipped", skipped, "bad", bad)
