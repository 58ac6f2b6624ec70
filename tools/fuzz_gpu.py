"""differential fuzz: corrupted-but-parsable streams, GPU (all stages) vs oracle"""
import importlib, os, random, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import corpus, gpudecode, hevcutil, orc, synthutil
pkg = importlib.import_module("heif-decoder-lib_amd")
hm = pkg.lib()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 99)
names = ["rext_ts_tools", "rext_chroma_qp_list6_422", "rext_mono_rice_rdpcm", "pcm_422_10", "lossless_all", "sl_pps_422_10", "yuv444_12_ctb16", "hi422_12", "wpp_tiles_slices", "conf_window_422_10", "ragged", "ctb64_wpp", "hi422_10", "hi420_10", "ctb16_nosao", "pcm_bypass_sl_wpp", "yuv444_rare", "rext_cross_444_all", "rext_ts_bypass_422_10", "rext_nosmooth_rice", "mono10", "slices_headers", "tiles_3x2_nolf", "dense_lowqp", "sl_sps_12bit_highqp"]
blobs, tags = [], []
for name in names:
    data = corpus.stream(name)
    lo = len(data) // 3
    tries = 0
    got = 0
    while got < 30 and tries < 1500:
        tries += 1
        b = bytearray(data)
        for _ in range(rng.randrange(1, int(os.environ.get("HM_FUZZ_FLIPS", "4")))):
            i = rng.randrange(lo, len(b))
            if rng.random() < 0.2: b[i] = rng.randrange(256)
            else: b[i] ^= 1 << rng.randrange(8)
        try:
            blob = hevcutil.parse(hm, bytes(b))
        except RuntimeError:
            continue
        blobs.append(blob); tags.append((name, tries)); got += 1
print(len(blobs), "corrupted streams parse", flush=True)
bad = 0
for bits in (0, 3):
    out = gpudecode.decode_pictures(pkg, blobs, bits)
    for (name, t), blob, g in zip(tags, blobs, out):
        exp, _ = orc.oracle_decode(blob, bits, crop=True)
        for c in range(len(exp)):
            if not np.array_equal(g[c], exp[c]):
                d = np.argwhere(g[c] != exp[c])
                print("MISMATCH", name, t, "stage", bits, "plane", c, len(d), d[0].tolist(), int(g[c][tuple(d[0])]), int(exp[c][tuple(d[0])]), flush=True)
                bad += 1
                break
print("mismatching pictures:", bad)
