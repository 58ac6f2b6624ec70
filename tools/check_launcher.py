#!/usr/bin/env python3
"""How far is the chain launcher's own choice of a cut (chain.hip: hm_launch_chain, ChainTuning) from the best cut it could have
been forced into?  For N = 24 ... 6144 copies of one 512x512 tile (class HM_CLASS_ONLY, default 8bit_420_ctb32): reconstruction
ms (k_residual + k_chain, HIP events, best of 3) with the launcher left alone and with every cut forced through the test hook
(hm_debug_set: chain_pairs / chain_ring / chain_share) - one process, one batch per N.  The table is the regression check of the
launcher's calibration: a threshold that has drifted shows up as a row whose ratio is well above 1.
usage (repo root, GPU box): [HM_CHECK_TILE=256|1024] python3 tools/check_launcher.py [N ...]   (HM_CHECK_TILE: another tile size than 512 - the calibration
was read from 512 x 512 tiles)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bench_classes import CLASSES, MORE  # noqa: E402

# (pairs, ring, share): -1 = the launcher's own; pairs 0 a wave per picture, 1 per pair of CTU rows, 2 per row, 3 per chain of a row
CUTS = {"auto": (-1, -1, -1), "picture": (0, 0, -1), "row_pairs": (1, 0, -1), "rows": (2, 0, -1), "chains": (3, 0, -1),
        "share2": (-1, -1, 2), "share3": (-1, -1, 3), "share4": (-1, -1, 4)}
for w in (2, 3, 4, 8):
    CUTS[f"ring{w}_pairs"] = (1, w, -1)
for w in (2, 4, 8, 16):
    CUTS[f"ring{w}_rows"] = (2, w, -1)
    CUTS[f"ring{w}_chains"] = (3, w, -1)


def main():
    import torch
    import __graft_entry__ as g
    import knobs
    import synthutil
    pkg = g.load_package(test_knobs="always")
    capi, L = pkg.capi, pkg.lib()
    dev = torch.device("cuda:0")
    counts = [int(a) for a in sys.argv[1:]] or [24, 48, 96, 192, 384, 768, 1280, 1536, 2048, 3072, 5120, 6144]
    name = os.environ.get("HM_CLASS_ONLY", "8bit_420_ctb32")
    T = int(os.environ.get("HM_CHECK_TILE", "512"))
    cfg = dict(width=T, height=T, qp=27, cu_qp_delta=1, sao=1, sign_hiding=1, density=60)
    cfg.update({**CLASSES, **MORE}[name])
    blobs = [capi.parse_hevc(synthutil.picture(7700000 + i, **cfg)) for i in range(8)]
    bps = 2 if cfg.get("bit_depth", 8) > 8 else 1
    ys, cs = L.hm_plane_stride(T, bps), L.hm_plane_stride(T // 2, bps)
    ch = T // 2 if cfg.get("chroma_format", 1) == 1 else T
    y = torch.zeros((max(64, T), ys), dtype=torch.uint8, device=dev)
    cb = torch.zeros((max(64, ch), cs), dtype=torch.uint8, device=dev)
    cr = torch.zeros((max(64, ch), cs), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for n in counts:
        batch = capi.Batch()
        for i in range(n):
            d = capi.TileDest()
            d.plane[0], d.plane[1], d.plane[2] = y.data_ptr(), cb.data_ptr(), cr.data_ptr()
            d.pitch[0], d.pitch[1], d.pitch[2] = ys, cs, cs
            d.canvas_width, d.canvas_height, d.x0, d.y0 = T, T, 0, 0
            batch.add(blobs[i % 8], d)
        batch.upload(st)
        ms = {}
        for cut, (pairs, ring, share) in CUTS.items():
            knobs.set_knob(L, "chain_pairs", pairs)
            knobs.set_knob(L, "chain_ring", ring)
            knobs.set_knob(L, "chain_share", share)
            batch.execute(1, st)  # (reconstruction only)
            torch.cuda.synchronize()
            batch.set_profiling(3)
            for _ in range(3):
                batch.execute(1, st)
            torch.cuda.synchronize()
            ms[cut] = round(min(batch.timings_ms(s)[0] for s in range(3)), 3)
            batch.check()
        for k in ("chain_pairs", "chain_ring", "chain_share"):
            knobs.set_knob(L, k, -1)
        batch.close()
        forced = {k: v for k, v in ms.items() if k != "auto"}
        best = min(forced, key=forced.get)
        rows.append(dict(tiles=n, auto_ms=ms["auto"], best_forced=best, best_ms=forced[best], ratio=round(ms["auto"] / forced[best], 3), all=ms))
        print(f"{n:5d} tiles: launcher {ms['auto']:7.3f} ms, best forced cut {best:13s} {forced[best]:7.3f} ms, ratio {ms['auto'] / forced[best]:.3f}", flush=True)
    print(json.dumps({"class": name, "tile": T, "rows": rows}))


if __name__ == "__main__":
    main()
