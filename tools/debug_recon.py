#!/usr/bin/env python3
"""GPU-box debugging aid: reconstruct one stream (stage 0) on the GPU, compare with the oracle, and list the transform
blocks (records of the command stream, either record format) that contain mismatching samples.
usage: python3 tools/debug_recon.py <corpus case | path.hevc | seed> [key=value synth overrides]"""
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

HM_PIC_SPLIT_CHAINS = 0x1000
CTB_BYTES = 52


def records(blob):
    """yields (ctb x, ctb y, index in chain, chain length, dict of record fields) for every record, either format"""
    b = bytes(blob)
    ctb_w = struct.unpack_from("<H", b, 28)[0]
    flags = struct.unpack_from("<I", b, 36)[0]
    n_slices, n_ctbs, n_tus, n_coeffs, off_slices, off_ctbs, off_tus, off_coeffs = struct.unpack_from("<8I", b, 44)
    split = bool(flags & HM_PIC_SPLIT_CHAINS)
    for i in range(n_ctbs):
        o = off_ctbs + CTB_BYTES * i
        first, cnt = struct.unpack_from("<IH", b, o)
        first_c, cnt_c = struct.unpack_from("<IH", b, o + 36)
        cx, cy = i % ctb_w, i // ctb_w
        for (f, n) in ((first, cnt), (first_c, cnt_c)):
            for k in range(n):
                t = f + k
                if split:
                    pos, info, mode, qp, qpy, avail, count = struct.unpack_from("<BBBBbBH", b, off_tus + 8 * t)
                    s = 1 << (info & 7)
                    r = dict(x=(pos & 15) * 4, y=(pos >> 4) * 4, info=info, mode=mode, qp=qp, n=count & 0x7FF,
                             aL=s if count & 0x800 else 0, aT=s if count & 0x1000 else 0, aBL=(avail & 15) * 4, aTR=(avail >> 4) * 4)
                else:
                    x, y, info, mode, qp, qpy, nc, cfirst, aL, aBL, aT, aTR = struct.unpack_from("<BBBBBbHIBBBB", b, off_tus + 16 * t)
                    r = dict(x=x, y=y, info=info, mode=mode & 63, qp=qp, n=nc, aL=aL, aT=aT, aBL=aBL, aTR=aTR)
                yield cx, cy, k, n, r


def main():
    pkg = g.load_package(test_knobs=True)
    name = sys.argv[1]
    if os.path.exists(name):
        data = open(name, "rb").read()
    elif name in corpus.CASES:
        data = corpus.stream(name)
    else:
        kw = dict(a.split("=") for a in sys.argv[2:])
        base = dict(corpus.TILE) if "tile" in kw else {}
        kw.pop("tile", None)
        base.update({k: int(v) for k, v in kw.items()})
        data = synthutil.picture(int(name), **base)
    blob = pkg.capi.parse_hevc(data)
    got = gpudecode.decode_pictures(pkg, [blob], 0)[0]
    exp, info = orc.oracle_decode(blob, 0, crop=True)
    w, h, cf = info["width"], info["height"], info["chroma"]
    ctb = 1 << bytes(blob)[23]
    print("picture", w, h, "chroma", cf, "ctb", ctb, "split chains", bool(struct.unpack_from("<I", bytes(blob), 36)[0] & HM_PIC_SPLIT_CHAINS))
    total = 0
    for c in range(len(exp)):
        d = got[c] != exp[c]
        total += int(d.sum())
        print("plane", c, "mismatches", int(d.sum()))
    if not total:
        print("OK")
        return 0
    sw = 1 if cf == 3 else 2
    shh = 2 if cf == 1 else 1
    shown = 0
    kinds = {}
    for cx, cy, k, n, r in records(blob):
        inf = r["info"]
        c = (inf >> 3) & 3
        s = 1 << (inf & 7)
        X = cx * (ctb if c == 0 else ctb // sw) + r["x"]
        Y = cy * (ctb if c == 0 else ctb // shh) + r["y"]
        pl_g, pl_e = got[c], exp[c]
        blk = (pl_g[Y:Y + s, X:X + s] != pl_e[Y:Y + s, X:X + s])
        bad = int(blk.sum())
        interior = (r["aL"] >= s and r["aT"] >= s and (inf & 0x80))
        key = (s, c, "int" if interior else "bord", "cbf" if inf & 0x20 else "nocbf", "ts" if inf & 0x40 else "")
        kk = kinds.setdefault(key, [0, 0])
        kk[0] += 1
        kk[1] += bad > 0
        if bad and shown < 12:
            shown += 1
            print(f"ctb ({cx},{cy}) rec {k}/{n}: c={c} size={s} at ({X},{Y}) mode={r['mode']} cbf={bool(inf & 0x20)} tskip={bool(inf & 0x40)} qp={r['qp']} ncoef={r['n']} "
                  f"avail L{r['aL']} BL{r['aBL']} T{r['aT']} TR{r['aTR']} TL{bool(inf & 0x80)} bad={bad}")
            print("  got", pl_g[Y:Y + min(s, 4), X:X + min(s, 8)].tolist())
            print("  exp", pl_e[Y:Y + min(s, 4), X:X + min(s, 8)].tolist())
    print("blocks by kind: (size, cidx, interior, cbf, tskip): [count, with mismatches]")
    for k in sorted(kinds):
        print(" ", k, kinds[k])
    return 1


if __name__ == "__main__":
    sys.exit(main())
