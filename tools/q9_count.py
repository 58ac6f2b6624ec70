#!/usr/bin/env python3
"""Quirk Q9, counted (CPU only).  The fork's AVX2 SAO (third-party/libde265/libde265/x86_new/x86_sao.cc:271,320-369)
walks a chroma CTB of an 8-bit picture in steps of 16 columns: where chroma CTBs are 8 samples wide (CTBs of 16, 4:2:0 / 4:2:2) it also
filters 8 columns of the CTB to the right with THIS CTB's parameters; its SSE4 and scalar code do not.  Product and oracle follow the
scalar code (the standard's).  How many pictures of that class does a caller on an AVX2 host get different pixels for?

Over seeded pictures of the class with tools/big_sweep.py's parameter ranges (ordinary syntax), SAO on: the real libde265 of /root/reference
(oracle/_ref) decodes every stream twice - default acceleration (AVX2 here) and de265_acceleration_SCALAR - and the planes are compared.
The same for the neighbouring classes (CTB 32, 4:4:4, 4:0:0, 10-bit) as a control: they must never differ.
usage: python3 tools/q9_count.py [pictures per class]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc, synthutil, hevcutil
import __graft_entry__ as g

hm = g.load_package().lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(20261004)
flags = open("/proc/cpuinfo").read()
print("host:", "avx2" if " avx2" in flags else "no avx2", "| reference build: oracle/_ref (bundled libde265 1.0.8 of /root/reference, its own x86 dispatch)")


def picture(seed, cf, bd, l2, sao):
    kw = dict(width=8 * rng.randrange(2, 40), height=8 * rng.randrange(2, 30), chroma_format=cf, bit_depth=bd, log2_ctb=l2,
              qp=rng.randrange(10, 48), cu_qp_delta=rng.randrange(2), sao=sao, deblock_disable=int(rng.random() < 0.15),
              sign_hiding=rng.randrange(2), transform_skip=rng.randrange(2), strong_intra=rng.randrange(2), cb_qp_offset=rng.randrange(-6, 7),
              cr_qp_offset=rng.randrange(-6, 7), beta_offset_div2=rng.randrange(-4, 5), tc_offset_div2=rng.randrange(-4, 5),
              density=rng.randrange(20, 100), wpp=rng.randrange(2), log2_min_cb=3)
    return synthutil.picture(seed, **kw), kw


def count(name, cf, bd, l2, sao=1, n=N):
    differ = chroma_only = product_eq_scalar = 0
    samples = diff_samples = 0
    wide = 0
    for i in range(n):
        data, kw = picture(880000 + i, cf, bd, l2, sao)
        a, _ = orc.ref_decode(data, 0)
        b, _ = orc.ref_decode(data, orc.REF_F_SCALAR)
        d = [int((x != y).sum()) for x, y in zip(a, b)]
        if sum(d):
            differ += 1
            chroma_only += d[0] == 0
            diff_samples += sum(d)
        samples += sum(x.size for x in a)
        wide += kw["width"] > 16
        mine, _ = orc.oracle_decode(hevcutil.parse(hm, data), 3, crop=True)
        product_eq_scalar += all(np.array_equal(m, r) for m, r in zip(mine, b))
    print(f"{name:34s} pictures {n:4d}  default != scalar: {differ:4d} ({100.0 * differ / n:5.1f} %)  only in chroma: {chroma_only:4d}  "
          f"samples that differ: {diff_samples} of {samples} ({100.0 * diff_samples / max(samples, 1):.3f} %)  product parser + oracle == scalar build: {product_eq_scalar} of {n}")
    return differ


q9 = count("8-bit 4:2:0 CTB 16, SAO on", 1, 8, 4) + count("8-bit 4:2:2 CTB 16, SAO on", 2, 8, 4)
count("8-bit 4:2:0 CTB 16, SAO off", 1, 8, 4, sao=0, n=N // 4)
for name, cf, bd, l2 in (("control: 8-bit 4:2:0 CTB 32", 1, 8, 5), ("control: 8-bit 4:4:4 CTB 16", 3, 8, 4), ("control: 8-bit 4:0:0 CTB 16", 0, 8, 4),
                         ("control: 10-bit 4:2:0 CTB 16", 1, 10, 4)):
    count(name, cf, bd, l2, n=N // 4)
