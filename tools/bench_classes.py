#!/usr/bin/env python3
"""Kernel timings for the picture classes outside the headline workload (10/12-bit, 4:2:2, CTB 16/64), to catch
performance cliffs: N copies of one synthetic 512x512 tile per class, HIP-event times per launch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CLASSES = {
    "8bit_420_ctb32": dict(log2_ctb=5),
    "8bit_mono_ctb32": dict(log2_ctb=5, chroma_format=0),
    "8bit_420_ctb16": dict(log2_ctb=4),
    "8bit_420_ctb64": dict(log2_ctb=6),
    "10bit_420_ctb32": dict(log2_ctb=5, bit_depth=10),
    "10bit_422_ctb32": dict(log2_ctb=5, bit_depth=10, chroma_format=2),
    "12bit_422_ctb64": dict(log2_ctb=6, bit_depth=12, chroma_format=2),
}
# (only on request, HM_CLASS_ONLY=...: what lies between the classes above)
MORE = {
    "8bit_422_ctb32": dict(log2_ctb=5, chroma_format=2),
    "10bit_420_ctb16": dict(log2_ctb=4, bit_depth=10),
    "10bit_420_ctb64": dict(log2_ctb=6, bit_depth=10),
    "10bit_mono_ctb32": dict(log2_ctb=5, bit_depth=10, chroma_format=0),
    "8bit_444_ctb32": dict(log2_ctb=5, chroma_format=3),  # (the rare-syntax kernel k_recon: records in decode order)
    "10bit_444_ctb32": dict(log2_ctb=5, chroma_format=3, bit_depth=10),
}


def main():
    import torch
    import __graft_entry__ as g
    import synthutil
    pkg = g.load_package(test_knobs=True)
    capi, L = pkg.capi, pkg.lib()
    dev = torch.device("cuda:0")
    n = int(os.environ.get("HM_CLASS_TILES", "1536"))
    out = {}
    only = os.environ.get("HM_CLASS_ONLY", "").split(",")
    for name, kw in list(CLASSES.items()) + [(k, v) for k, v in MORE.items() if k in only]:
        cfg = dict(width=512, height=512, qp=27, cu_qp_delta=1, sao=1, sign_hiding=1, density=60)
        cfg.update(kw)
        blobs = [capi.parse_hevc(synthutil.picture(7700000 + i, **cfg)) for i in range(8)]
        bps = 2 if cfg.get("bit_depth", 8) > 8 else 1
        cf = cfg.get("chroma_format", 1)
        if os.environ.get("HM_CLASS_ONLY") and name not in os.environ["HM_CLASS_ONLY"].split(","):
            continue
        ys, cs = L.hm_plane_stride(512, bps), L.hm_plane_stride(512 if cf == 3 else 256, bps)
        ch = 256 if cf == 1 else 512
        y = torch.zeros((512, ys), dtype=torch.uint8, device=dev)
        cb = torch.zeros((ch, cs), dtype=torch.uint8, device=dev)
        cr = torch.zeros((ch, cs), dtype=torch.uint8, device=dev)
        batch = capi.Batch()
        for i in range(n):
            d = capi.TileDest()
            d.plane[0], d.plane[1], d.plane[2] = y.data_ptr(), cb.data_ptr(), cr.data_ptr()
            d.pitch[0], d.pitch[1], d.pitch[2] = ys, cs, cs
            d.canvas_width, d.canvas_height, d.x0, d.y0 = 512, 512, 0, 0
            batch.add(blobs[i % 8], d)
        st = torch.cuda.current_stream().cuda_stream
        batch.upload(st)
        batch.execute(3, st)
        torch.cuda.synchronize()
        batch.set_profiling(3)
        for _ in range(3):
            batch.execute(3, st)
        torch.cuda.synchronize()
        ms = [sum(batch.timings_ms(s)[q] for s in range(3)) / 3 for q in range(3)]
        mp = n * 0.262144
        out[name] = {"tiles": n, "k_recon_ms": round(ms[0], 3), "k_deblock_ms": round(ms[1], 3), "k_sao_paste_ms": round(ms[2], 3),
                     "GP_per_s_kernels": round(mp / sum(ms), 1)}
        batch.check()  # (also prints the phase sums of an instrumented k_chain: HM_CHAIN_TIMING_PRINT=1)
        batch.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
