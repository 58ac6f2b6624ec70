#!/usr/bin/env python3
"""bench.py — megapixels/s of the HEIC grid -> RGB24 hot path on MI355X.

One "step" = one pass of the GPU hot path (HEVC-intra reconstruction -> deblocking -> SAO +
grid paste -> fused YCbCr->RGB24) over a batch of synthetic 12 MP HEIC grids (4032x3024 output,
8x6 grid of 512x512 tiles, 8-bit 4:2:0, CTB 32, QP 27 +- cu_qp_delta, SAO + deblocking + sign
hiding) whose command streams (host CABAC output) are already resident in
HBM.  Images are independent, so with N GPUs every rank decodes its own images (weak scaling,
no data-path collective); rank 0 prints ONE JSON line.

Parity gate: before timing, image 0 of rank 0 is checked bit-exactly against the CPU oracle
(oracle/liboracle.so; and against the real reference decoder oracle/_ref when it is present).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GRID_COLS, GRID_ROWS, TILE = 8, 6, 512
OUT_W, OUT_H = 4032, 3024
MP_PER_IMAGE = OUT_W * OUT_H / 1e6
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def tile_stream(image_index, tile_index):
    import synthutil
    from corpus import TILE as TILE_CFG
    kw = dict(TILE_CFG)
    kw.update(vui=1, full_range=1, matrix=6)
    return synthutil.picture(1200000 + 48 * image_index + tile_index, **kw)


def build_images(pkg, n_images, first_image, dev):
    """synthesise + entropy-decode (host) the tiles, allocate canvases, queue + upload the batch"""
    import torch
    capi = pkg.capi
    L = pkg.lib()
    ys, cs = L.hm_plane_stride(OUT_W, 1), L.hm_plane_stride(OUT_W // 2, 1)
    os_ = L.hm_plane_stride(OUT_W, 3)
    batch = capi.Batch()
    images = []
    NT = GRID_COLS * GRID_ROWS
    # synthesis + host entropy decode of all tiles on a small thread pool (both are C calls that release the GIL);
    # the single-thread parse rate reported as host_entropy_decode is timed separately on the tiles of image 0
    from concurrent.futures import ThreadPoolExecutor

    def make(k):
        data = tile_stream(first_image + k // NT, k % NT)
        return (data if k < NT else None), capi.parse_hevc(data)

    pool = ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1))
    made = pool.map(make, range(n_images * NT))  # yields in order; a tile is handed to the batch (which copies it) and dropped
    host_parse_s = 0.0
    for j in range(n_images):
        y = torch.zeros((OUT_H, ys), dtype=torch.uint8, device=dev)
        cb = torch.zeros((OUT_H // 2, cs), dtype=torch.uint8, device=dev)
        cr = torch.zeros((OUT_H // 2, cs), dtype=torch.uint8, device=dev)
        rgb = torch.zeros((OUT_H, os_), dtype=torch.uint8, device=dev)
        streams, blobs = [], []
        for t in range(NT):
            data, blob = next(made)
            d = capi.TileDest()
            d.plane[0], d.plane[1], d.plane[2] = y.data_ptr(), cb.data_ptr(), cr.data_ptr()
            d.pitch[0], d.pitch[1], d.pitch[2] = ys, cs, cs
            d.canvas_width, d.canvas_height = OUT_W, OUT_H
            d.x0, d.y0 = (t % GRID_COLS) * TILE, (t // GRID_COLS) * TILE
            d.tile_has_nclx, d.tile_full_range, d.tile_matrix = 1, 1, 6
            batch.add(blob, d)
            if j == 0:
                streams.append(data)
                blobs.append(blob)
        if j == 0:  # single-thread parse rate, on the tiles of image 0
            for data in streams:
                t0 = time.perf_counter()
                capi.parse_hevc(data)
                host_parse_s += time.perf_counter() - t0
        desc = capi.ColourDesc(OUT_W, OUT_H, 8, 1, 0, 0, 0, 0, capi.HM_OUT_RGB, ys, cs, cs, os_)
        images.append(dict(y=y, cb=cb, cr=cr, rgb=rgb, desc=desc, streams=streams, blobs=blobs))
    pool.shutdown()
    return batch, images, (ys, cs, os_), host_parse_s


def cpu_image(images0, strides, use_ref):
    """CPU restatement of the same path for one image: decode 48 tiles, paste, convert.  Returns RGB array."""
    import numpy as np
    import orc
    ys, cs, os_ = strides
    o = orc.load()
    y = np.zeros((max(64, OUT_H), ys), np.uint8)
    cb = np.zeros((max(64, OUT_H // 2), cs), np.uint8)
    cr = np.zeros((max(64, OUT_H // 2), cs), np.uint8)
    for t in range(GRID_COLS * GRID_ROWS):
        if use_ref:
            planes, info = orc.ref_decode(images0["streams"][t], 0)
            has_nclx, full, matrix = 1, info["full_range"], info["matrix"]
        else:
            planes, info = orc.oracle_decode(images0["blobs"][t], 3)
            has_nclx, full, matrix = 1, info["full_range"], info["matrix"]
        x0, y0 = (t % GRID_COLS) * TILE, (t // GRID_COLS) * TILE
        for c, (canvas, stride) in enumerate(((y, ys), (cb, cs), (cr, cs))):
            p8 = np.ascontiguousarray(planes[c].astype(np.uint8))
            rc = o.orc_paste_tile_plane(orc.ptr(p8), p8.shape[1], p8.shape[1], p8.shape[0], orc.ptr(canvas), stride,
                                        OUT_W, OUT_H, x0, y0, c, 1, 8, has_nclx, full, matrix)
            assert rc == 0
    out = np.zeros((max(64, OUT_H), os_), np.uint8)
    o.orc_ycbcr420_to_rgb_int(orc.ptr(y), ys, orc.ptr(cb), cs, orc.ptr(cr), cs, OUT_W, OUT_H, 0, 0, 0, orc.ptr(out), os_, 10)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=384, help="12 MP images per GPU per step (384 x 48 = 18432 independent tiles, 25 GB of the 288 GB HBM; "
                    "the reconstruction kernel's end-of-launch tail amortises with the batch: 48 images give 70, 192 give 83, 384 give 86, 768 give 87 GP/s)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N>1 code path on one GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the hm_decode_item single-image clock (profiling runs)")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    L = pkg.lib()
    capi = pkg.capi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    B = args.images
    batch, images, strides, host_parse_s = build_images(pkg, B, rank * B, dev)
    st = torch.cuda.current_stream().cuda_stream
    batch.upload(st)

    # the canvases of the batch share one colour descriptor: one conversion call (hm_colour_convert_batch)
    PtrArr = C.c_void_p * len(images)
    p_y, p_cb, p_cr, p_rgb = (PtrArr(*[im[k].data_ptr() for im in images]) for k in ("y", "cb", "cr", "rgb"))
    L.hm_colour_convert_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

    def colour():
        capi.check(L.hm_colour_convert_batch(C.byref(images[0]["desc"]), len(images), p_y, p_cb, p_cr, p_rgb, st))

    def step():
        batch.execute(3, st)
        colour()

    # ---- parity gate (rank 0, image 0) ----
    parity = "skipped"
    step()
    torch.cuda.synchronize()
    if rank == 0 and not args.no_parity:
        import numpy as np
        import orc
        got = images[0]["rgb"].cpu().numpy()
        exp = cpu_image(images[0], strides, use_ref=False)
        if not np.array_equal(got[:OUT_H, :OUT_W * 3], exp[:OUT_H, :OUT_W * 3]):
            print(json.dumps({"error": "parity gate failed: GPU RGB != CPU oracle"}))
            raise SystemExit(3)
        parity = "bit-exact vs oracle"
        if orc.have_ref():
            exp2 = cpu_image(images[0], strides, use_ref=True)
            if not np.array_equal(exp[:OUT_H, :OUT_W * 3], exp2[:OUT_H, :OUT_W * 3]):
                print(json.dumps({"error": "parity gate failed: oracle != reference decoder"}))
                raise SystemExit(3)
            parity = "bit-exact vs oracle and reference libde265"

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    batch.set_profiling(args.steps)  # one HIP-event slot per timed step, read back after the timed region
    if dist:
        dist.barrier()
    k_ms = [0.0, 0.0, 0.0, 0.0]  # recon, deblock, sao+paste, colour
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        batch.execute(3, st)
        cev[i][0].record()
        colour()
        cev[i][1].record()
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    for i, (a, b) in enumerate(cev):
        k_ms[3] += a.elapsed_time(b)
        ms = batch.timings_ms(i)
        for q in range(3):
            k_ms[q] += ms[q]
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # clock (D) of SURVEY 8d: H2D of the command streams + kernels, result on the device (not `value`)
    d_ms = None
    if rank == 0 and world == 1 and not args.no_e2e:  # the side clocks belong to the single-GPU run
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            batch.upload(st)
            step()
        torch.cuda.synchronize()
        d_ms = (time.perf_counter() - t1) / 3 * 1e3

    if rank == 0:
        total_mp = world * B * MP_PER_IMAGE * args.steps
        value = total_mp / elapsed
        stream_b, sample_b = batch.algorithmic_bytes()
        names = ["k_recon", "k_deblock", "k_sao_paste", "k_ycbcr420_int(colour)"]
        # algorithmic bytes per step of each kernel: recon = command stream + samples out;
        # deblock = one pass: read + write of the samples = 2 x sample bytes;
        # sao+paste = read + write samples; colour = 4.5 B per output pixel
        alg = [stream_b + sample_b, 2 * sample_b, 2 * sample_b, int(4.5 * OUT_W * OUT_H * B)]
        avg_ms = [m / args.steps for m in k_ms]
        dom = max(range(4), key=lambda q: avg_ms[q])
        kernels = {names[q]: {"ms_per_step": round(avg_ms[q], 4), "algorithmic_bytes": int(alg[q]),
                              "GBps": round(alg[q] / avg_ms[q] / 1e6, 1) if avg_ms[q] > 0 else None} for q in range(4)}
        achieved = alg[dom] / avg_ms[dom] / 1e6
        out = {
            "metric": "megapixels/sec HEIC grid->RGB24",
            "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "12MP HEIC grid (4032x3024, 8x6 tiles of 512x512, 8-bit 4:2:0, CTB32) -> RGB24",
                       "images_per_gpu_per_step": B, "tiles_per_step_per_gpu": B * 48,
                       "timed_region": "recon+deblock+SAO/paste+colour kernels; command streams resident in HBM",
                       "parity": parity},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": pmc_traffic(names[dom], B)},
            "kernels": kernels,
            "host_entropy_decode": {"MP_per_s_per_core": round(48 * 0.262144 / host_parse_s, 1) if host_parse_s else None,
                                    "note": "hm_hevc_parse (CABAC -> command stream), 1 thread, outside the timed region"},
        }
        if d_ms is not None:
            out["device_inclusive"] = {"ms_per_step": round(d_ms, 3), "MP_per_s": round(B * MP_PER_IMAGE / d_ms * 1e3, 1),
                                       "command_stream_bytes": int(stream_b),
                                       "note": "H2D of the command streams (pinned staging) + all kernels, RGB left on the device; 1 GPU"}
        if not args.no_e2e and world == 1:
            out["end_to_end"] = end_to_end(pkg, images[0])
        if args.cpu_seconds > 0 and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(images[0], strides, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def end_to_end(pkg, image0):
    """Clock (E) of SURVEY 8d, NOT `value`: one 12 MP grid .heic through hm_decode_item = box parsing + host
    entropy decode (threads) + H2D + kernels + D2H into a libheif-layout host plane, per image, one at a time."""
    import heifwriter
    import pipeline
    data = heifwriter.write_heic(image0["streams"], (TILE, TILE), grid=(GRID_ROWS, GRID_COLS, OUT_W, OUT_H))
    res = {}
    f = pipeline.HeifFile(pkg.lib(), data)
    try:
        for threads in (1, 8, 48):
            f.decode(f.primary(), 10, threads=threads, copy=False)  # warm-up (allocations, code objects, worker threads)
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                f.decode(f.primary(), 10, threads=threads, copy=False)
            dt = (time.perf_counter() - t0) / n
            res[f"host_threads_{threads}"] = {"ms_per_image": round(dt * 1e3, 2), "MP_per_s": round(MP_PER_IMAGE / dt, 1)}
    finally:
        f.close()
    res["note"] = "hm_decode_item: HEIF parse + CABAC on host threads + H2D + GPU kernels + D2H into pinned host planes, single image latency"
    return res


def pmc_traffic(kernel, images):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, separate
    passes, scaled per image); None when no measurement for this kernel is on file."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        per_image = t["kernels"][kernel]["hbm_bytes_per_image"]
        return int(per_image * images)
    except Exception:
        return None


def cpu_baseline(image0, strides, budget_s):
    """The reference CPU path timed on this host, one thread, on a bounded sample (whole 12 MP images)."""
    import orc
    use_ref = orc.have_ref()
    t0 = time.perf_counter()
    n = 0
    while True:
        cpu_image(image0, strides, use_ref)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 64:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n * MP_PER_IMAGE / dt, 2), "unit": "MP/s", "cores": 1,
            "kind": "reference" if use_ref else "port",
            "sample": f"{n} x 12 MP grid image (48 tiles): "
                      + ("libde265 of /root/reference built by oracle/Makefile (SSE4.1/AVX2 kernels) for the tile decode, "
                         "oracle C restatement for paste + colour (libheif is unbuildable without cmake)" if use_ref
                         else "oracle C restatement (scalar) for decode, paste and colour")
                      + "; entropy decode included; " + f"{os.cpu_count()} host cpus visible"}


if __name__ == "__main__":
    main()
