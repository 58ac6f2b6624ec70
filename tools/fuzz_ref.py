"""corrupted streams (1-2 bit flips in the slice data): the product parser + oracle against the real reference decoder (oracle/_ref, CPU)

--determinism: every stream the reference accepts and the product refuses is decoded twice more by the reference, in child
processes whose allocator fills fresh memory with different bytes (MALLOC_PERTURB_): equal pictures = the reference's output does
not depend on memory it never wrote, i.e. there is something to match; different = its picture holds uninitialised samples."""
import hashlib, importlib, os, random, subprocess, sys, tempfile
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import corpus, hevcutil, orc
if len(sys.argv) > 3 and sys.argv[1] == "--child-planes":  # the reference's luma plane of one stream -> .npy (or nothing)
    try:
        planes, _ = orc.ref_decode(open(sys.argv[2], "rb").read(), 0)
        np.save(sys.argv[3], planes[0])
    except Exception:
        pass
    sys.exit(0)
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

def tile_scan(stream, ctb_w, ctb_h):
    """raster address -> tile-scan address (CtbAddrRsToTs, 6.5.1) from the stream's PPS (7.3.2.3.1): [u32 length][NAL] records"""
    import hevcutil
    pps = next(n for n in hevcutil.split_nals(stream) if ((n[0] >> 1) & 63) == 34)
    rbsp = bytearray()
    z = 0
    for x in pps[2:]:  # without the emulation prevention bytes
        if z >= 2 and x == 3:
            z = 0
            continue
        rbsp.append(x)
        z = z + 1 if x == 0 else 0
    pos = [0]

    def u(n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((rbsp[pos[0] >> 3] >> (7 - (pos[0] & 7))) & 1)
            pos[0] += 1
        return v

    def ue():
        k = 0
        while u(1) == 0:
            k += 1
        return (1 << k) - 1 + u(k)

    def se():
        v = ue()
        return (v + 1) // 2 if v & 1 else -(v // 2)
    ue(); ue(); u(1); u(1); u(3); u(1); u(1); ue(); ue(); se(); u(1); u(1)
    if u(1):
        ue()  # diff_cu_qp_delta_depth
    se(); se(); u(1); u(1); u(1); u(1)
    tiles = u(1)
    u(1)  # entropy_coding_sync_enabled_flag
    cols, rows = [ctb_w], [ctb_h]
    if tiles:
        nc, nr = ue() + 1, ue() + 1
        if u(1):  # uniform_spacing_flag (6-3, 6-4)
            cols = [(i + 1) * ctb_w // nc - i * ctb_w // nc for i in range(nc)]
            rows = [(i + 1) * ctb_h // nr - i * ctb_h // nr for i in range(nr)]
        else:
            cols = [ue() + 1 for _ in range(nc - 1)]
            cols.append(ctb_w - sum(cols))
            rows = [ue() + 1 for _ in range(nr - 1)]
            rows.append(ctb_h - sum(rows))
    ts_of = [0] * (ctb_w * ctb_h)
    ts = 0
    y0 = 0
    for rh in rows:
        x0 = 0
        for cw_ in cols:
            for y in range(y0, y0 + rh):
                for x in range(x0, x0 + cw_):
                    ts_of[y * ctb_w + x] = ts
                    ts += 1
            x0 += cw_
        y0 += rh
    return ts_of


if "--conceal" in sys.argv:
    # r05: the same 900 streams with HM_PARSE_CONCEAL.  A stream the strict parser refuses comes back as a picture: every CTB in front of
    # the damage decoded from the data, the rest concealed.  Against the reference: the CTBs up to one CTB row + 2 in front of the first
    # concealed one must be the reference's - unless the reference itself gave up earlier (its picture is then memory it never wrote
    # from there on: two runs under different MALLOC_PERTURB_ differ) - that is checked for every stream that does not match.
    import struct
    rng = random.Random(5)
    st = dict(intact=0, concealed=0, refused=0, reference_fails=0, front_equal=0, front_differs_reference_undefined=0, front_differs_REAL=0, tiles_not_compared=0)
    real = []
    with tempfile.TemporaryDirectory() as d:
        for name in names:
            data = corpus.stream(name)
            lo = len(data) // 3
            for t in range(60):
                b = bytearray(data)
                for _ in range(rng.randrange(1, 3)):
                    b[rng.randrange(lo, len(b))] ^= 1 << rng.randrange(8)
                b = bytes(b)
                try:
                    blob, n_conc, first = hevcutil.parse_concealing(hm, b)
                except RuntimeError:
                    st["refused"] += 1
                    continue
                if n_conc == 0:
                    st["intact"] += 1
                    # a stream the strict parser refuses that needs no concealed CTB: a damaged segment ran on into CTBs the segments
                    # behind it then took over - the whole picture must be the reference's
                    try:
                        hevcutil.parse(hm, b)
                    except RuntimeError:
                        mine, _ = orc.oracle_decode(blob, 3, crop=True)
                        try:
                            ref, _ = orc.ref_decode(b, 0)
                            same = len(mine) == len(ref) and all(m.shape == r.shape and np.array_equal(m, r) for m, r in zip(mine, ref))
                        except Exception:
                            same = None
                        key = "taken_over_equal" if same else ("taken_over_reference_fails" if same is None else "taken_over_DIFFERS")
                        st[key] = st.get(key, 0) + 1
                    continue
                st["concealed"] += 1
                mine, _ = orc.oracle_decode(blob, 3)
                try:
                    ref, _ = orc.ref_decode(b, 0)
                except Exception:
                    st["reference_fails"] += 1
                    continue
                w, h = struct.unpack_from("<HH", blob, 8)
                cl, cr, ct, cb = struct.unpack_from("<4H", blob, 12)
                ctb = 1 << blob[23]
                ctb_w = (w + ctb - 1) // ctb
                flags = struct.unpack_from("<I", blob, 36)[0]
                def ctb_equal(a, r, c):
                    x0, y0 = max((c % ctb_w) * ctb, cl), max((c // ctb_w) * ctb, ct)
                    x1, y1 = min((c % ctb_w) * ctb + ctb, w - cr), min((c // ctb_w) * ctb + ctb, h - cb)
                    return x1 <= x0 or y1 <= y0 or np.array_equal(a[y0:y1, x0:x1], r[y0 - ct:y1 - ct, x0 - cl:x1 - cl])

                if flags & 0x40:
                    # HEVC tiles (r06): "in front of" is the DECODING order - tile by tile, raster inside a tile (6.5.1).  A CTB can be
                    # compared when it and its eight neighbours (whose samples the in-loop filters of the CTB read) all come before the
                    # first concealed CTB in that order - for a picture without tiles that is the "one CTB row + 2" rule below.
                    ctb_h = (h + ctb - 1) // ctb
                    ts_of = tile_scan(b, ctb_w, ctb_h)
                    ts_first = ts_of[first]
                    front = [c for c in range(ctb_w * ctb_h)
                             if all(ts_of[(c // ctb_w + dy) * ctb_w + c % ctb_w + dx] < ts_first
                                    for dy in (-1, 0, 1) for dx in (-1, 0, 1) if 0 <= c // ctb_w + dy < ctb_h and 0 <= c % ctb_w + dx < ctb_w)]
                    st["tiles_compared_in_decoding_order"] = st.get("tiles_compared_in_decoding_order", 0) + 1
                    st["tile_ctbs_compared"] = st.get("tile_ctbs_compared", 0) + len(front)
                else:
                    front = range(max(0, first - ctb_w - 2))
                bad = [c for c in front if not ctb_equal(mine[0], ref[0], c)]
                if not bad:
                    st["front_equal"] += 1
                    continue
                f = os.path.join(d, "s.bin")
                open(f, "wb").write(b)
                runs = []
                for v in (85, 170):
                    o = os.path.join(d, f"p{v}.npy")
                    if os.path.exists(o):
                        os.remove(o)
                    subprocess.run([sys.executable, __file__, "--child-planes", f, o], env=dict(os.environ, MALLOC_PERTURB_=str(v)), capture_output=True)
                    runs.append(np.load(o) if os.path.exists(o) else None)
                undefined_from = None
                if runs[0] is not None and runs[1] is not None:
                    n_ctbs = ctb_w * ((h + ctb - 1) // ctb)
                    full = lambda p: np.pad(p, ((ct, cb), (cl, cr)))
                    for c in range(n_ctbs):
                        x0, y0 = (c % ctb_w) * ctb, (c // ctb_w) * ctb
                        if not np.array_equal(full(runs[0])[y0:y0 + ctb, x0:x0 + ctb], full(runs[1])[y0:y0 + ctb, x0:x0 + ctb]):
                            undefined_from = c
                            break
                if undefined_from is not None and bad[0] >= undefined_from - ctb_w - 2:
                    st["front_differs_reference_undefined"] += 1
                else:
                    st["front_differs_REAL"] += 1
                    real.append((name, t, first, bad[0], undefined_from))
    print("with HM_PARSE_CONCEAL:", st)
    print("front differs although the reference is defined there:", real[:20])
