"""corrupted streams (1-2 bit flips in the slice data): the product parser + oracle against the real reference decoder (oracle/_ref, CPU)"""
import importlib, os, random, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import corpus, hevcutil, orc
pkg = importlib.import_module("heif-decoder-lib_amd")
hm = pkg.lib()
rng = random.Random(5)
names = ["ragged", "ctb64_wpp", "hi422_10", "hi420_10", "ctb16_nosao", "pcm_bypass_sl_wpp", "yuv444_rare", "rext_cross_444_all", "rext_ts_bypass_422_10", "rext_nosmooth_rice", "mono10", "slices_headers", "tiles_3x2_nolf", "dense_lowqp", "sl_sps_12bit_highqp"]
stat = dict(both_ok_equal=0, both_ok_differ=0, mine_only=0, ref_only=0, both_fail=0)
diffs = []
for name in names:
    data = corpus.stream(name)
    lo = len(data) // 3
    for t in range(60):
        b = bytearray(data)
        for _ in range(rng.randrange(1, 3)):
            b[rng.randrange(lo, len(b))] ^= 1 << rng.randrange(8)
        b = bytes(b)
        try:
            mine, _ = orc.oracle_decode(hevcutil.parse(hm, b), 3, crop=True)
        except RuntimeError:
            mine = None
        try:
            ref, _ = orc.ref_decode(b, 0)
        except Exception:
            ref = None
        if mine is None and ref is None: stat["both_fail"] += 1
        elif mine is None: stat["ref_only"] += 1
        elif ref is None: stat["mine_only"] += 1
        else:
            same = len(mine) == len(ref) and all(m.shape == r.shape and np.array_equal(m, r) for m, r in zip(mine, ref))
            stat["both_ok_equal" if same else "both_ok_differ"] += 1
            if not same: diffs.append((name, t))
print(stat); print(diffs[:20])
