"""corrupted streams (1-2 bit flips in the slice data): the product parser + oracle against the real reference decoder (oracle/_ref, CPU)

--determinism: every stream the reference accepts and the product refuses is decoded twice more by the reference, in child
processes whose allocator fills fresh memory with different bytes (MALLOC_PERTURB_): equal pictures = the reference's output does
not depend on memory it never wrote, i.e. there is something to match; different = its picture holds uninitialised samples."""
import hashlib, importlib, os, random, subprocess, sys, tempfile
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import corpus, hevcutil, orc
if len(sys.argv) > 2 and sys.argv[1] == "--child":  # md5 of the reference's planes of one stream (or "fail")
    try:
        planes, _ = orc.ref_decode(open(sys.argv[2], "rb").read(), 0)
        print(hashlib.md5(b"".join(p.tobytes() for p in planes)).hexdigest())
    except Exception:
        print("fail")
    sys.exit(0)
pkg = importlib.import_module("heif-decoder-lib_amd")
hm = pkg.lib()
rng = random.Random(5)
names = ["ragged", "ctb64_wpp", "hi422_10", "hi420_10", "ctb16_nosao", "pcm_bypass_sl_wpp", "yuv444_rare", "rext_cross_444_all", "rext_ts_bypass_422_10", "rext_nosmooth_rice", "mono10", "slices_headers", "tiles_3x2_nolf", "dense_lowqp", "sl_sps_12bit_highqp"]
stat = dict(both_ok_equal=0, both_ok_differ=0, mine_only=0, ref_only=0, both_fail=0)
diffs, ref_only = [], []
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
            why = hm.hm_last_error().decode() if hasattr(hm, "hm_last_error") else "?"
        try:
            ref, _ = orc.ref_decode(b, 0)
        except Exception:
            ref = None
        if mine is None and ref is None: stat["both_fail"] += 1
        elif mine is None:
            stat["ref_only"] += 1
            ref_only.append((name, t, b, why))
        elif ref is None: stat["mine_only"] += 1
        else:
            same = len(mine) == len(ref) and all(m.shape == r.shape and np.array_equal(m, r) for m, r in zip(mine, ref))
            stat["both_ok_equal" if same else "both_ok_differ"] += 1
            if not same: diffs.append((name, t, b))
print(stat); print([d[:2] for d in diffs[:20]])
if "--determinism" in sys.argv:
    det = dict(same=0, differ=0, fail=0)
    same_cases = []
    with tempfile.TemporaryDirectory() as d:
        for name, t, b, why in ref_only:
            f = os.path.join(d, "s.bin")
            open(f, "wb").write(b)
            got = [subprocess.run([sys.executable, __file__, "--child", f], env=dict(os.environ, MALLOC_PERTURB_=str(v)), capture_output=True, text=True).stdout.strip().split("\n")[-1]
                   for v in (85, 170)]
            if "fail" in got or "" in got: det["fail"] += 1
            elif got[0] == got[1]:
                det["same"] += 1
                same_cases.append((name, t, why))
            else: det["differ"] += 1
    print("reference accepts, product refuses:", det)
    dd = dict(reference_deterministic=0, reference_not=0)
    real = []
    with tempfile.TemporaryDirectory() as d:
        for name, t, b in diffs:
            f = os.path.join(d, "s.bin")
            open(f, "wb").write(b)
            got = [subprocess.run([sys.executable, __file__, "--child", f], env=dict(os.environ, MALLOC_PERTURB_=str(v)), capture_output=True, text=True).stdout.strip().split("\n")[-1]
                   for v in (85, 170)]
            if got[0] == got[1] and got[0] != "fail":
                dd["reference_deterministic"] += 1
                real.append((name, t))
            else: dd["reference_not"] += 1
    print("both accept, pictures differ:", dd, real[:30])
    import collections
    print("the product's reasons for the deterministic ones:", collections.Counter(w[:60] for _, _, w in same_cases).most_common(20))
