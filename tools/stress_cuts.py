#!/usr/bin/env python3
"""Race hunt for the hand-overs between waves of k_chain (r06: the early CTU start shortens the distance between a row and the row above it):
N copies of distinct 512x512 tiles are reconstructed in the wave-per-picture cut (no hand-over between waves: the reference), then REPS times in
every cut that hands rows over - through LDS, through HBM, in rings of 2 ... 16 bands, alternating or not - and every output plane is compared
each time.  A hazard that needs an unlucky schedule shows up as a rare mismatch.  usage (repo root, GPU box): python3 tools/stress_cuts.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as g
import knobs, bench

pkg = g.load_package(test_knobs="always")
capi, L = pkg.capi, pkg.lib()
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
# (pairs, ring, alt, np)
CUTS = {"chains np2": (3, 0, 1, 2), "chains np6": (3, 0, 1, 6), "chains np16": (3, 0, 1, 16), "rows np3": (2, 0, 1, 3), "rows np8": (2, 0, 1, 8),
        "ring2 alt": (3, 2, 1, 0), "ring4 alt": (3, 4, 1, 0), "ring8 alt": (3, 8, 1, 0), "ring2 keep": (3, 2, 0, 0), "ring3 keep": (3, 3, 0, 0), "ring4 keep": (3, 4, 0, 0),
        "ring8 keep": (3, 8, 0, 0), "ring4 rows": (2, 4, 1, 0), "ring5 rows": (2, 5, 1, 0), "ring8 rows": (2, 8, 1, 0)}
bad_total = 0
for bit_depth, n_tiles in ((8, 6), (8, 48), (8, 160), (10, 48)):
    blobs = [capi.parse_hevc(bench.tile_stream(9300000 + 13 * k, bit_depth=bit_depth)) for k in range(min(n_tiles, 24))]
    bps = 2 if bit_depth > 8 else 1
    ys, cs = L.hm_plane_stride(512, bps), L.hm_plane_stride(256, bps)
    planes = [(torch.zeros((512, ys), dtype=torch.uint8, device=dev), torch.zeros((256, cs), dtype=torch.uint8, device=dev), torch.zeros((256, cs), dtype=torch.uint8, device=dev)) for _ in range(n_tiles)]
    batch = capi.Batch()
    for i in range(n_tiles):
        d = capi.TileDest()
        d.plane[0], d.plane[1], d.plane[2] = (p.data_ptr() for p in planes[i])
        d.pitch[0], d.pitch[1], d.pitch[2] = ys, cs, cs
        d.canvas_width, d.canvas_height, d.x0, d.y0 = 512, 512, 0, 0
        batch.add(blobs[i % len(blobs)], d)
    batch.upload(st)
    def run():
        for p in planes:
            for q in p: q.zero_()
        batch.execute(1, st)  # (reconstruction only)
        torch.cuda.synchronize()
        batch.check()
        return [torch.cat([q.flatten() for q in p]) for p in planes]
    knobs.set_knob(L, "chain_pairs", 0); knobs.set_knob(L, "chain_ring", 0)
    want = run()
    for name, (pairs, ring, alt, np_) in CUTS.items():
        knobs.set_knob(L, "chain_pairs", pairs); knobs.set_knob(L, "chain_ring", ring); knobs.set_knob(L, "chain_alt", alt); knobs.set_knob(L, "chain_np", np_)
        bad = 0
        for _ in range(reps):
            got = run()
            bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(got, want))
        bad_total += bad
        print(f"{bit_depth}-bit, {n_tiles:4d} tiles, {name:12s}: {reps} runs, pictures that differ from the wave-per-picture cut: {bad}", flush=True)
    for k, v in (("chain_pairs", -1), ("chain_ring", -1), ("chain_alt", 1), ("chain_np", 0)):
        knobs.set_knob(L, k, v)
    batch.close()
print("TOTAL mismatching pictures:", bad_total)
