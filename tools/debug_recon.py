#!/usr/bin/env python3
"""GPU-box debugging aid: reconstruct one stream (stage 0) on the GPU, compare with the oracle, and list the transform
blocks (records of the command stream) that contain mismatching samples.
usage: python3 tools/debug_recon.py <corpus case | path.hevc> [key=value synth overrides]"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
import corpus  # noqa: E402
import gpudecode  # noqa: E402
import orc  # noqa: E402
import synthutil  # noqa: E402

pkg = g.load_package()
name = sys.argv[1]
if os.path.exists(name):
    data = open(name, "rb").read()
elif name in corpus.CASES:
    data = corpus.stream(name)
else:
    kw = dict(a.split("=") for a in sys.argv[2:])
    data = synthutil.picture(int(name), **{k: int(v) for k, v in kw.items()})
blob = pkg.capi.parse_hevc(data)
got = gpudecode.decode_pictures(pkg, [blob], 0)[0]
exp, info = orc.oracle_decode(blob, 0, crop=True)
w, h, cf = info["width"], info["height"], info["chroma"]
log2_ctb = blob[23]
ctb = 1 << log2_ctb
n_ctbs, n_tus = struct.unpack_from("<II", blob, 0x30)
off_ctbs, off_tus = struct.unpack_from("<II", blob, 0x40)
ctb_w = struct.unpack_from("<H", blob, 0x1C)[0]
print("picture", w, h, "chroma", cf, "ctb", ctb, "ctb_w", ctb_w, "tus", n_tus)
total = 0
for c in range(len(exp)):
    d = got[c] != exp[c]
    total += int(d.sum())
    print("plane", c, "mismatches", int(d.sum()))
if not total:
    print("OK")
    sys.exit(0)
sw = 1 if cf == 3 else 2
shh = 2 if cf == 1 else 1
shown = 0
kinds = {}
for cidx_ctb in range(n_ctbs):
    first, cnt = struct.unpack_from("<IH", blob, off_ctbs + 44 * cidx_ctb)
    first_c, cnt_c = struct.unpack_from("<IH", blob, off_ctbs + 44 * cidx_ctb + 36)
    cx, cy = cidx_ctb % ctb_w, cidx_ctb // ctb_w
    for t in list(range(first, first + cnt)) + list(range(first_c, first_c + cnt_c)):
        x, y, inf, mode, qp, qpy, nc, cfirst, aL, aBL, aT, aTR = struct.unpack_from("<BBBBBbHIBBBB", blob, off_tus + 16 * t)
        c = (inf >> 3) & 3
        s = 1 << (inf & 7)
        X = cx * (ctb if c == 0 else ctb // sw) + x
        Y = cy * (ctb if c == 0 else ctb // shh) + y
        pl_g, pl_e = got[c], exp[c]
        blk = (pl_g[Y:Y + s, X:X + s] != pl_e[Y:Y + s, X:X + s])
        bad = int(blk.sum())
        interior = (aL >= s and aT >= s and (inf & 0x80))
        key = (s, c, "int" if interior else "bord", "cbf" if inf & 0x20 else "nocbf", "ts" if inf & 0x40 else "")
        k = kinds.setdefault(key, [0, 0])
        k[0] += 1
        k[1] += bad > 0
        if bad and shown < 12:
            shown += 1
            print(f"ctb ({cx},{cy}) rec {t - first}/{cnt}: c={c} size={s} at ({X},{Y}) mode={mode} cbf={bool(inf & 0x20)} tskip={bool(inf & 0x40)} qp={qp} ncoef={nc} "
                  f"avail L{aL} BL{aBL} T{aT} TR{aTR} TL{bool(inf & 0x80)} bad={bad}")
            print("  got", pl_g[Y:Y + min(s, 4), X:X + min(s, 8)].tolist())
            print("  exp", pl_e[Y:Y + min(s, 4), X:X + min(s, 8)].tolist())
print("blocks by kind: (size, cidx, interior, cbf, tskip): [count, with mismatches]")
for k in sorted(kinds):
    print(" ", k, kinds[k])
