#!/usr/bin/env python3
"""bench.py — megapixels/s of the HEIC grid -> RGB24 hot path on MI355X.

One "step" = one pass of the GPU hot path (HEVC-intra reconstruction -> deblocking -> SAO + grid paste -> fused
YCbCr->RGB24) over a batch of synthetic 12 MP HEIC grids (4032x3024 output, 8x6 grid of 512x512 tiles, 8-bit 4:2:0,
CTB 32, QP 27 +- cu_qp_delta, SAO + deblocking + sign hiding) whose command streams (the host
CABAC output) are already resident in HBM: `value` is this KERNEL clock (K of SURVEY §8d) by the measurement contract;
the clocks that include the transfers and the host (D: + H2D, E: whole files in, host pixels out) are reported next to
it in the same JSON line and are never `value`.

`--gpus N`: one process per GPU.  Started under a launcher (WORLD_SIZE set) the process is one rank; started plainly
with N > 1 it launches the N ranks itself (`python -m torch.distributed.run`, as a child process, before anything has
touched a GPU).  Images are independent, so every rank decodes its own images (weak scaling, no data-path collective).
`--mode grid` shards ONE 16384x16384 grid by tile rows and gathers the RGB row slabs to rank 0
with one RCCL collective (timed separately, checked bit for bit against the one-rank result).

Parity gate: before timing, one randomly chosen image of EVERY rank is checked bit-exactly against the CPU oracle
(oracle/liboracle.so; and against the real reference decoder oracle/_ref when it is present).
"""
import argparse
import ctypes as C
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GRID_COLS, GRID_ROWS, TILE = 8, 6, 512
OUT_W, OUT_H = 4032, 3024
MP_PER_IMAGE = OUT_W * OUT_H / 1e6
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


# --------------------------------------------------------------------------------------------------------------
# workload construction (host): synthesise + entropy-decode tiles, allocate canvases, queue the batch
# --------------------------------------------------------------------------------------------------------------

def tile_stream(seed, **over):
    import synthutil
    from corpus import TILE as TILE_CFG
    kw = dict(TILE_CFG)
    kw.update(vui=1, full_range=1, matrix=6)
    kw.update(over)
    return synthutil.picture(seed, **kw)


def make_streams(capi, seeds, keep_data=True, **over):
    """synthesis + host entropy decode of the tiles on a small thread pool (both are C calls that release the GIL);
    yields (data, blob) in order"""
    from concurrent.futures import ThreadPoolExecutor

    def make(seed):
        data = tile_stream(seed, **over)
        return (data if keep_data else None), capi.parse_hevc(data)

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
        yield from pool.map(make, seeds)


class GridBatch:
    """n images, each a cols x rows grid of tile x tile coded pictures pasted into an out_w x out_h canvas (8-bit 4:2:0),
    converted to RGB24: device canvases + one hm_batch."""

    def __init__(self, pkg, dev, cols, rows, tile, out_w, out_h, nclx=(1, 1, 6)):
        # nclx: (present, full_range, matrix) of the tile items' colour profile as the decoder plugin / the colr box reports it
        # (decoder_libde265.cc:339-362, context.cc:1840-1850); present && !full_range && matrix != 0 = the paste rescales
        # (context.cc:2504-2528).  A stream without a VUI reports (1, 0, 2).
        self.pkg, self.dev = pkg, dev
        self.nclx = nclx
        self.cols, self.rows, self.tile, self.out_w, self.out_h = cols, rows, tile, out_w, out_h
        L = pkg.lib()
        self.ys, self.cs, self.os = L.hm_plane_stride(out_w, 1), L.hm_plane_stride((out_w + 1) // 2, 1), L.hm_plane_stride(out_w, 3)
        self.batch = pkg.capi.Batch()
        self.images = []

    def add_image(self, blobs, y0_tiles=0, rgb=None):
        """blobs: cols * rows command streams, row-major; rgb: the image's RGB rows somewhere else ([out_h, self.os] view, e.g. rows of a
        larger image: grid mode decodes a slab straight into its rows of the gathered image)"""
        import torch
        capi = self.pkg.capi
        y = torch.zeros((max(64, self.out_h), self.ys), dtype=torch.uint8, device=self.dev)
        cb = torch.zeros((max(64, (self.out_h + 1) // 2), self.cs), dtype=torch.uint8, device=self.dev)
        cr = torch.zeros((max(64, (self.out_h + 1) // 2), self.cs), dtype=torch.uint8, device=self.dev)
        if rgb is None:
            rgb = torch.zeros((max(64, self.out_h), self.os), dtype=torch.uint8, device=self.dev)
        else:
            assert rgb.shape[0] >= self.out_h and rgb.stride(0) == self.os and rgb.stride(1) == 1
        for t, blob in enumerate(blobs):
            d = capi.TileDest()
            d.plane[0], d.plane[1], d.plane[2] = y.data_ptr(), cb.data_ptr(), cr.data_ptr()
            d.pitch[0], d.pitch[1], d.pitch[2] = self.ys, self.cs, self.cs
            d.canvas_width, d.canvas_height = self.out_w, self.out_h
            d.x0, d.y0 = (t % self.cols) * self.tile, (t // self.cols) * self.tile
            d.tile_has_nclx, d.tile_full_range, d.tile_matrix = self.nclx
            self.batch.add(blob, d)
        self.images.append(dict(y=y, cb=cb, cr=cr, rgb=rgb))

    def finish(self, st, images_per_group=0):
        """upload, and attach the canvases' YCbCr -> RGB24 conversion to the batch (hm_batch_set_colour): one execute call
        is then the whole hot path, the filters and the conversion running group of images by group of images"""
        capi = self.pkg.capi
        self.batch.upload(st)
        n = len(self.images)
        PtrArr = C.c_void_p * n
        self.p = [PtrArr(*[im[k].data_ptr() for im in self.images]) for k in ("y", "cb", "cr", "rgb")]
        self.desc = capi.ColourDesc(self.out_w, self.out_h, 8, 1, 0, 0, 0, 0, capi.HM_OUT_RGB, self.ys, self.cs, self.cs, self.os)
        self.batch.set_colour(self.desc, n, *self.p, images_per_group)

    def step(self, st):
        self.batch.execute(int(os.environ.get("HM_BENCH_STAGES", "3")), st)  # (HM_BENCH_STAGES: probe runs only, with --no-parity)

    def pixels(self):
        return len(self.images) * self.out_w * self.out_h


def cpu_grid_image(streams, blobs, cols, rows, tile, out_w, out_h, strides, use_ref, nclx=None):
    """CPU restatement of the same path for one image: decode the tiles, paste, convert.  Returns the RGB array.
    nclx: (present, full_range, matrix) of the tile items when a colr box overrides the streams' VUI."""
    import numpy as np
    import orc
    ys, cs, os_ = strides
    o = orc.load()
    y = np.zeros((max(64, out_h), ys), np.uint8)
    cb = np.zeros((max(64, (out_h + 1) // 2), cs), np.uint8)
    cr = np.zeros((max(64, (out_h + 1) // 2), cs), np.uint8)
    for t in range(cols * rows):
        if use_ref:
            planes, info = orc.ref_decode(streams[t], 0)
        else:
            planes, info = orc.oracle_decode(blobs[t], 3)
        x0, y0 = (t % cols) * tile, (t // cols) * tile
        for c, (canvas, stride) in enumerate(((y, ys), (cb, cs), (cr, cs))):
            p8 = np.ascontiguousarray(planes[c].astype(np.uint8))
            rc = o.orc_paste_tile_plane(orc.ptr(p8), p8.shape[1], p8.shape[1], p8.shape[0], orc.ptr(canvas), stride,
                                        out_w, out_h, x0, y0, c, 1, 8, *(nclx if nclx else (1, info["full_range"], info["matrix"])))
            assert rc == 0
    out = np.zeros((max(64, out_h), os_), np.uint8)
    o.orc_ycbcr420_to_rgb_int(orc.ptr(y), ys, orc.ptr(cb), cs, orc.ptr(cr), cs, out_w, out_h, 0, 0, 0, orc.ptr(out), os_, 10)
    return out


def timed_steps(torch, gb, st, steps, dist=None):
    """K clock: `steps` passes bracketed by synchronize (+ barrier); per-kernel HIP-event times on the launch stream"""
    gb.batch.set_profiling(steps)  # one HIP-event timeline per timed step, read back after the timed region
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        gb.step(st)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    k_ms = [0.0, 0.0, 0.0, 0.0, 0.0]  # prediction chains (or the whole reconstruction), deblock, sao+paste, colour, residual pre-pass
    for i in range(steps):
        ms = gb.batch.timings5_ms(i)
        for q in range(5):
            k_ms[q] += ms[q]
    gb.batch.set_profiling(0)
    return elapsed, [m / steps for m in k_ms]


# names as rocprofv3 lists them (the headline workload - 8-bit 4:2:0 without rare syntax - runs the split-chain
# reconstruction: k_residual, then k_chain)
KERNEL_NAMES = ["k_chain", "k_deblock", "k_sao_paste", "k_ycbcr420_int(colour)", "k_residual"]
TAIL_NAME = "k_tail420(deblock+sao+paste+colour)"  # the fused kernel: timing slot 2, slots 1 and 3 are empty


def kernel_names(gb):
    names = list(KERNEL_NAMES)
    if gb.batch.tail_fused():
        names[1], names[2], names[3] = None, TAIL_NAME, None
    return names


def kernel_table(gb, avg_ms, colour_bytes_per_px=4.5):
    stream_b, sample_b, level_b, resid_b = gb.batch.algorithmic_bytes4()
    # algorithmic bytes per step of each kernel: residual pre-pass = command stream in + residual samples
    # out; prediction chains = command stream without the levels + residual in + samples out (r02's single kernel:
    # command stream + samples out); deblock = read + write of the samples; sao+paste = read + write; colour = 1.5 B in +
    # 3 B out per output pixel
    chain_b = stream_b - level_b + resid_b + sample_b
    alg = [chain_b, 2 * sample_b, 2 * sample_b, int(colour_bytes_per_px * gb.pixels()), stream_b + resid_b]
    names = kernel_names(gb)
    if gb.batch.tail_fused():  # reads the reconstruction once, writes the pixels once
        alg[1], alg[2], alg[3] = 0, sample_b + int((colour_bytes_per_px - 1.5) * gb.pixels()), 0
    table = {names[q]: {"ms_per_step": round(avg_ms[q], 4), "algorithmic_bytes": int(alg[q]),
                        "GBps": round(alg[q] / avg_ms[q] / 1e6, 1) if avg_ms[q] > 0 else None,
                        "frac_of_hbm_peak": round(alg[q] / avg_ms[q] / 1e6 / HBM_PEAK_GBPS, 4) if avg_ms[q] > 0 else None} for q in (4, 0, 1, 2, 3) if names[q]}
    # the north star's "HBM-read roofline" taken literally: only the 1.5 B/px the colour kernel reads (report both)
    if names[2] == TAIL_NAME and avg_ms[2] > 0:  # the fused tail on the literal "HBM-read" convention: the 1.5 B/px of samples it reads
        table[TAIL_NAME]["read_only_GBps"] = round(sample_b / avg_ms[2] / 1e6, 1)
        table[TAIL_NAME]["read_only_frac_of_hbm_peak"] = round(sample_b / avg_ms[2] / 1e6 / HBM_PEAK_GBPS, 4)
    cms = avg_ms[3]
    if cms > 0 and names[3]:
        table[KERNEL_NAMES[3]]["read_only_GBps"] = round(1.5 * gb.pixels() / cms / 1e6, 1)
        table[KERNEL_NAMES[3]]["read_only_frac_of_hbm_peak"] = round(1.5 * gb.pixels() / cms / 1e6 / HBM_PEAK_GBPS, 4)
    return table, alg, stream_b, sample_b


PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")


def pmc_traffic(kernels, images):
    """HBM bytes per launch of the named kernels (summed) from the COMMITTED rocprofv3 PMC passes (profiles/r0N_pmc_traffic.json:
    FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, separate passes of tools/pmc_traffic.sh).  Only a file taken at
    THIS run's load counts (r05: its images_per_launch equals --images - L2 behaviour and the traffic per image change with the
    rounds of waves in flight); otherwise (None, None): the line then says null, not a number scaled from another load.
    A table look-up of a committed measurement of the same command, not live counters of this run."""
    for name in PMC_FILES:
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
            if int(t.get("images_per_launch", -1)) != int(images):
                continue
            total = sum(t["kernels"][k]["hbm_bytes_per_step"] for k in kernels)
            return int(total), f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh at {images} images per launch = this run's load; not live counters of this run)"
        except Exception:
            continue
    return None, None


def roofline_lines(gb, avg_ms, alg, stream_b, sample_b, step_ms, images):
    """The roofline object of the JSON line, on the byte model of SURVEY 8(d) - which cannot rise by adding a pass:
      stage  reconstruction (k_residual + k_chain): command stream in + reconstructed samples out (1.5 B/px of 8-bit 4:2:0);
             the residuals / micro-ops the two kernels hand each other are intermediates, not algorithmic bytes
      step   the whole hot path: the stage + the tail (samples in, RGB out)
    `achieved` / `frac` at the top level are the stage's (the kernels that take the largest share of the step); the
    per-kernel figures on each kernel's own input + output stay in `kernels` and, for the dominant one, in `own_io`."""
    names = kernel_names(gb)
    recon = [names[q] for q in (4, 0) if names[q]]
    recon_ms = avg_ms[4] + avg_ms[0]
    stage_b = stream_b + sample_b
    tail_b = sum(alg[q] for q in (1, 2, 3))
    step_b = stage_b + tail_b
    stage_gbps = stage_b / recon_ms / 1e6
    tr_stage, src = pmc_traffic(recon, images)
    tr_step, _ = pmc_traffic([n for n in names if n], images)
    dom = max(range(5), key=lambda q: avg_ms[q])
    own = alg[dom] / avg_ms[dom] / 1e6
    tr_dom, _ = pmc_traffic([names[dom]], images)
    r = {"bound": "hbm", "kernel": " + ".join(recon) + " (the reconstruction stage of SURVEY 8d)",
         "bytes_model": "SURVEY 8(d): command stream in + 1.5 B/px of reconstructed samples out; intermediates between the two kernels not counted",
         "achieved": round(stage_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(stage_gbps / HBM_PEAK_GBPS, 5),
         "traffic": tr_stage, "traffic_ratio": round(tr_stage / stage_b, 2) if tr_stage else None, "traffic_source": src,
         "stage": {"algorithmic_bytes": int(stage_b), "ms": round(recon_ms, 4), "GBps": round(stage_gbps, 1), "frac": round(stage_gbps / HBM_PEAK_GBPS, 5)},
         "step": {"algorithmic_bytes": int(step_b), "ms": round(step_ms, 4), "GBps": round(step_b / step_ms / 1e6, 1),
                  "frac": round(step_b / step_ms / 1e6 / HBM_PEAK_GBPS, 5), "traffic": tr_step,
                  "traffic_ratio": round(tr_step / step_b, 2) if tr_step else None},
         "own_io": {"kernel": names[dom], "note": "the dominant kernel on its OWN input + output bytes (includes the intermediates of the split: secondary figure)",
                    "algorithmic_bytes": int(alg[dom]), "ms": round(avg_ms[dom], 4), "GBps": round(own, 1), "frac": round(own / HBM_PEAK_GBPS, 5),
                    "traffic": tr_dom, "traffic_ratio": round(tr_dom / alg[dom], 2) if tr_dom else None}}
    return r


# --------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a launcher around us
# --------------------------------------------------------------------------------------------------------------

def launch_ranks(args):
    """Start the N ranks as a child job.  Nothing in this process has initialised a GPU (no torch.cuda call, no HIP
    library loaded), and the child is a fresh process tree - never an exec of a process that holds a device."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def emit(obj):
    """the run's JSON line.  RCCL prints a version banner through C stdio when its first communicator comes up; that text
    sits in libc's buffer until the process exits - flushed here first, so that the JSON is the LAST line of stdout."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    print(json.dumps(obj), flush=True)


def rccl_selftest(torch, pkg, dev):
    """N = 1: a one-rank process group on the requested backend ("nccl" IS RCCL on ROCm), so that the line shows librccl
    loads on this image and a collective on a device tensor runs - an all_reduce and the design's only data-path
    collective, shard.gather_slabs (a padded dist.gather), checked against its input.  Never fatal: an error is reported
    as text (the decode path needs no collective)."""
    import torch.distributed as dist
    info = {}
    try:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        try:
            t = torch.arange(1024, dtype=torch.float32, device=dev)
            dist.all_reduce(t)
            slab = torch.randint(0, 256, (37, 12096), dtype=torch.uint8, device=dev)
            full = pkg.shard.gather_slabs(slab, [37], dst=0)
            torch.cuda.synchronize()
            info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                    "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                    "all_reduce_ok": bool(torch.equal(t.cpu(), torch.arange(1024, dtype=torch.float32))),
                    "gather_slabs_ok": bool(full is not None and torch.equal(full, slab)),
                    "note": "one-rank RCCL group at N = 1 (a load + execute check of the backend); the N > 1 legs use the same calls"}
        finally:
            dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 - reported, not raised
        info = {"backend": "nccl", "error": f"{type(e).__name__}: {e}"[:300]}
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("batch", "grid"), default="batch",
                    help="batch: images sharded over the ranks (configs 2 / 3); grid: ONE 16384x16384 grid sharded by tile rows + RCCL gather (config 5)")
    ap.add_argument("--images", type=int, default=384, help="12 MP images per GPU per step (384 x 48 = 18432 independent tiles, 25 GB of the 288 GB HBM; "
                    "the reconstruction kernel's end-of-launch tail amortises with the batch)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N>1 code path on one GPU)")
    ap.add_argument("--allow-shared-gpu", action="store_true", help="let several ranks share one GPU (functional test of the N>1 path only; the JSON says so)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each cpu_baseline leg")
    ap.add_argument("--quick", action="store_true", help="headline only: skip the side clocks (D, E, pipelined E, CPU baseline, real content, configs 4 / 5)")
    ap.add_argument("--no-e2e", action="store_true", help="alias of --quick (profiling runs)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--grid-chunk-rows", type=int, default=0, help="--mode grid: tile rows per chunk a rank decodes and hands to the gather (0 = its whole slab: "
                    "32 tiles per launch sit deep in the few-pictures regime of the chain kernel, profiles/r05_launcher_check.txt)")
    ap.add_argument("--group", type=int, default=0, help="images per filter + colour group (hm_batch_set_colour; 0 = the whole batch per kernel launch)")
    args = ap.parse_args()
    if args.no_e2e:
        args.quick = True
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    run(args)


def run(args):
    import torch
    import __graft_entry__ as g
    pkg = g.load_package(test_knobs=True)
    L = pkg.lib()
    capi = pkg.capi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one process per GPU")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("no HIP device: the hot path has no CPU fallback")
    if world > ndev and not args.allow_shared_gpu:
        raise SystemExit(f"{world} ranks but only {ndev} GPU(s) visible (use --allow-shared-gpu for a functional test of the N>1 path)")
    dev_index = local_rank % ndev
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
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"only {dist.get_world_size()} of {args.gpus} ranks joined")
    dist_info = None
    if world == 1 and args.dist_backend == "nccl":
        dist_info = rccl_selftest(torch, pkg, dev)
    st = torch.cuda.current_stream().cuda_stream
    if args.mode == "grid":
        return run_grid(args, torch, pkg, dev, dist, rank, world, st, shared=world > ndev)

    # ---------------- batch mode: configs 2 / 3 ----------------
    B = args.images
    NT = GRID_COLS * GRID_ROWS
    first_image = rank * B
    gb = GridBatch(pkg, dev, GRID_COLS, GRID_ROWS, TILE, OUT_W, OUT_H)
    keep = min(B, 256)            # coded tiles of the first images are kept for the end-to-end legs
    rng = random.Random(1234 + rank)
    check_image = rng.randrange(B)  # parity gate: a random image of every rank
    kept, check = [], None
    keep_blobs = world == 1 and not args.quick  # (the variant legs of the single-GPU run build batches of their own from them)
    all_blobs = []
    seeds = (1200000 + 48 * (first_image + k // NT) + k % NT for k in range(B * NT))
    made = make_streams(capi, seeds)
    for j in range(B):
        tiles = [next(made) for _ in range(NT)]
        gb.add_image([b for _, b in tiles])
        if keep_blobs or (world > 1 and j < SHARE_IMAGES):
            all_blobs.append([b for _, b in tiles])
        if j < keep:
            kept.append([d for d, _ in tiles])
        if j == check_image:
            check = tiles
    gb.finish(st, args.group)
    strides = (gb.ys, gb.cs, gb.os)

    # ---- parity gate ----
    parity = "skipped"
    gb.step(st)
    torch.cuda.synchronize()
    ok = 1
    if not args.no_parity:
        import numpy as np
        import orc
        got = gb.images[check_image]["rgb"].cpu().numpy()
        exp = cpu_grid_image([d for d, _ in check], [b for _, b in check], GRID_COLS, GRID_ROWS, TILE, OUT_W, OUT_H, strides, use_ref=False)
        ok = int(np.array_equal(got[:OUT_H, :OUT_W * 3], exp[:OUT_H, :OUT_W * 3]))
        parity = "bit-exact vs oracle"
        if ok and orc.have_ref():
            exp2 = cpu_grid_image([d for d, _ in check], [b for _, b in check], GRID_COLS, GRID_ROWS, TILE, OUT_W, OUT_H, strides, use_ref=True)
            ok = int(np.array_equal(exp[:OUT_H, :OUT_W * 3], exp2[:OUT_H, :OUT_W * 3]))
            parity = "bit-exact vs oracle and reference libde265"
        parity += f" (image {check_image} of rank {rank}; one random image of every rank is checked)"
    if dist:
        t = torch.tensor([ok], dtype=torch.int32, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    if not ok:
        if rank == 0:
            emit({"error": "parity gate failed: GPU RGB != CPU oracle / reference on at least one rank"})
        raise SystemExit(3)

    for _ in range(args.warmup):
        gb.step(st)
    elapsed, avg_ms = timed_steps(torch, gb, st, args.steps, dist)
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = {}
    if world > 1 and not args.quick:
        multi_rank_legs(out, args, torch, pkg, dev, dev_index, st, gb, kept, dist, rank, world)
        share_leg_all_ranks(out, args, torch, pkg, dev, st, all_blobs, dist, rank, world)
    if rank == 0:
        extra = out
        total_mp = world * B * MP_PER_IMAGE * args.steps
        value = total_mp / elapsed
        kernels, alg, stream_b, sample_b = kernel_table(gb, avg_ms)
        out = {
            "metric": "megapixels/sec HEIC grid->RGB24",
            "value": round(value, 1), "unit": "MP/s",
            "value_clock": "K: GPU kernels only (reconstruction -> deblocking -> SAO/paste -> colour) on command streams already in HBM - no CABAC, no H2D, no D2H; "
                           ".heic bytes in -> RGB in host memory is end_to_end_MP_per_s below (host entropy decode bound)",
            "end_to_end_MP_per_s": None, "device_inclusive_MP_per_s": None, "n_gpus": world, "world_size": dist.get_world_size() if dist else 1,
            "dist": {"backend": dist.get_backend(), "world_size": dist.get_world_size()} if dist else dist_info,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "12MP HEIC grid (4032x3024, 8x6 tiles of 512x512, 8-bit 4:2:0, CTB32) -> RGB24",
                       "images_per_gpu_per_step": B, "tiles_per_step_per_gpu": B * 48,
                       "timed_region": "K clock of SURVEY 8d: recon+deblock+SAO/paste+colour kernels, command streams (host CABAC output) resident in HBM; "
                                       "the transfer- and host-inclusive clocks are device_inclusive / end_to_end_pipelined below",
                       "parity": parity},
            "roofline": roofline_lines(gb, avg_ms, alg, stream_b, sample_b, elapsed / args.steps * 1e3, B),
            "kernels": kernels,
        }
        out.update(extra)
        if world > ndev:
            out["config"]["shared_gpu"] = f"{world} ranks on {ndev} GPU(s): functional run of the N>1 path, not a scaling number"
        gb.batch.check()  # (a reconstruction wave that gave up a bounded wait would have flagged its launch)
        if not args.quick and world == 1:  # the side clocks belong to the single-GPU run
            side_legs(out, args, torch, pkg, dev, st, gb, kept, strides, stream_b, all_blobs)
            try:
                out["end_to_end_MP_per_s"] = out["end_to_end_pipelined"]["MP_per_s"]
                out["device_inclusive_MP_per_s"] = max(v["MP_per_s"] for k, v in out["device_inclusive"].items() if isinstance(v, dict) and "MP_per_s" in v)
            except Exception:
                pass
        emit(out)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


# --------------------------------------------------------------------------------------------------------------
# N > 1: the transfer- and host-inclusive clocks on EVERY rank (one pipeline / one host crew per GPU)
# --------------------------------------------------------------------------------------------------------------

def rank_cpu_share(rank, world):
    """this rank's share of the CPUs the job may use: (threads, (first cpu, count)) - contiguous slices of the affinity
    mask (the CPUs next to a GPU are contiguous on the usual two-socket hosts), threads = usable CPUs (cgroup quota) / world"""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    per = max(1, len(cpus) // world)
    mine = cpus[(rank * per) % len(cpus):][:per]
    contiguous = bool(mine) and mine == list(range(mine[0], mine[0] + len(mine)))
    threads = max(1, effective_cpus() // world)
    return threads, ((mine[0], len(mine)) if contiguous else None)


def multi_rank_legs(out, args, torch, pkg, dev, dev_index, st, gb, kept, dist, rank, world):
    """D (H2D of the command streams under the kernels) and E (.heic bytes in, RGB in host memory out, one pipeline per
    rank) measured on all ranks at the same time; rank 0 reports the aggregate over the slowest rank and every rank's own
    numbers.  Called by every rank."""
    B = len(gb.images)
    cpu = "cpu" if args.dist_backend != "nccl" else dev

    def gather(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=cpu)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        return [[float(x) for x in p.tolist()] for p in parts]

    # Every rank always reaches every collective below: a rank whose leg fails (output hash mismatch, pipeline error, fewer
    # files) reports status 0 and zeros instead of leaving the others blocked in all_gather until the backend's timeout.
    # ---- D ----
    d_ms, d_ok, err = 0.0, 1, ""
    try:
        copy = torch.cuda.Stream(device=dev)
        gb.batch.upload_execute(3, 3, copy.cuda_stream, st)
        torch.cuda.synchronize()
    except Exception as ex:  # noqa: BLE001
        d_ok, err = 0, f"D warm-up: {ex}"
    dist.barrier()
    if d_ok:
        try:
            t0 = time.perf_counter()
            for _ in range(3):
                gb.batch.upload_execute(3, 3, copy.cuda_stream, st)
            torch.cuda.synchronize()
            d_ms = (time.perf_counter() - t0) / 3 * 1e3
            gb.batch.check()
        except Exception as ex:  # noqa: BLE001
            d_ok, err = 0, f"D: {ex}"
    # ---- E ----
    threads, cpus = rank_cpu_share(rank, world)
    n_files = max(16, min(len(kept), 256 // world))
    dist.barrier()
    e_ok = 1
    e = {"seconds": 0.0, "images": 0, "outputs_hash_checked": 0}
    try:
        e = end_to_end_pipelined(pkg, kept[:n_files], threads=threads, cpus=cpus, device=dev_index)
    except Exception as ex:  # noqa: BLE001
        e_ok, err = 0, (err + "; " if err else "") + f"E: {ex}"
    if err:
        print(f"[bench rank {rank}] multi_rank_legs: {err}", file=sys.stderr, flush=True)
    rows = gather([d_ms, e["seconds"], e["images"], threads, cpus[0] if cpus else -1, cpus[1] if cpus else 0, e["outputs_hash_checked"], d_ok, e_ok])
    if rank == 0 and not all(r[7] and r[8] for r in rows):
        out["multi_rank_legs_failed"] = {"ranks_D": [i for i, r in enumerate(rows) if not r[7]], "ranks_E": [i for i, r in enumerate(rows) if not r[8]],
                                         "note": "see the failing rank's stderr; D / E aggregates omitted"}
        out["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
        return
    if rank == 0:
        d_max = max(r[0] for r in rows)
        e_max = max(r[1] for r in rows)
        out["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
        out["device_inclusive_all_ranks"] = {
            "MP_per_s": round(world * B * MP_PER_IMAGE / d_max * 1e3, 1), "ms_per_step_slowest_rank": round(d_max, 3),
            "per_rank_ms": [round(r[0], 3) for r in rows],
            "note": "hm_batch_upload_execute, 3 chunks, every rank its own GPU and copy stream, all ranks at once"}
        out["end_to_end_pipelined_all_ranks"] = {
            "MP_per_s": round(sum(r[2] for r in rows) * MP_PER_IMAGE / e_max, 1),
            "per_rank": [{"images": int(r[2]), "seconds": round(r[1], 3), "host_threads": int(r[3]), "cpu_set": [int(r[4]), int(r[5])] if r[5] else None,
                          "MP_per_s": round(r[2] * MP_PER_IMAGE / r[1], 1), "outputs_hash_checked": int(r[6])} for r in rows],
            "note": "one hm_pipeline per rank (its GPU, its share of the usable CPUs, pinned to its slice of the affinity mask); aggregate = all images / slowest rank"}
        out["end_to_end_MP_per_s"] = out["end_to_end_pipelined_all_ranks"]["MP_per_s"]
        out["device_inclusive_MP_per_s"] = out["device_inclusive_all_ranks"]["MP_per_s"]


# --------------------------------------------------------------------------------------------------------------
# side clocks (rank 0 at N = 1): never `value`
# --------------------------------------------------------------------------------------------------------------

def effective_cpus():
    """CPUs this process can really use: the affinity mask capped by the cgroup CPU quota (a container may see every
    CPU of the host and still be limited to a few CPU-seconds per second; threads beyond the quota only get throttled)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def wpp_parse_rates(pkg):
    """hm_hevc_parse_mt on the three real 1080p WPP pictures: ms per picture with 1 thread and with the usable CPUs
    (rows of a picture entropy-decoded in parallel, two CTBs apart; single-image latency)"""
    capi = pkg.capi
    res = {}
    n_cpu = effective_cpus()
    for name in ("basketball_1080p_qp32", "basketball_1080p_qp25", "basketball_1080p_qp1"):
        path = os.path.join(ROOT, "tests", "data", name + ".hevc")
        if not os.path.exists(path):
            continue
        data = open(path, "rb").read()
        row = {}
        for threads in sorted({1, 4, min(8, n_cpu), n_cpu}):
            capi.parse_hevc(data, threads=threads)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                capi.parse_hevc(data, threads=threads)
                best = min(best, time.perf_counter() - t0)
            row[f"threads_{threads}_ms"] = round(best * 1e3, 2)
        res[name] = row
    res["note"] = "1920x1080 intra pictures, CTB 64, 17 rows: best of 5; the command stream is identical to the serial parser's (tests/test_oracle_decode.py)"
    return res


def tile_parse_rates(pkg):
    """hm_hevc_parse_mt on one synthetic 12 MP picture coded with 2 x 6 HEVC tiles (an entry point per tile, no WPP): ms
    with 1 thread and with the rows of tiles entropy-decoded side by side"""
    import synthutil
    capi = pkg.capi
    data = synthutil.picture(7300001, width=4032, height=3024, log2_ctb=5, tile_cols=2, tile_rows=6, qp=30, density=40)
    row = {"stream_bytes": len(data)}
    for threads in (1, 2, 3, 6):
        capi.parse_hevc(data, threads=threads)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            capi.parse_hevc(data, threads=threads)
            best = min(best, time.perf_counter() - t0)
        row[f"threads_{threads}_ms"] = round(best * 1e3, 2)
    row["note"] = "4032x3024 8-bit 4:2:0, CTB 32, 2 x 6 tiles: the six rows of tiles are the units of work; same command stream as the serial parser (tests/test_oracle_decode.py)"
    return row


def guarded(out, key, fn):
    try:
        out[key] = fn()
    except Exception as e:  # a side leg must not lose the headline
        out[key] = {"error": f"{type(e).__name__}: {e}"}


def side_legs(out, args, torch, pkg, dev, st, gb, kept, strides, stream_b, all_blobs=()):
    B = len(gb.images)
    guarded(out, "host_entropy_decode", lambda: host_parse_rate(pkg, kept[0]))
    guarded(out, "wpp_row_parallel_parse", lambda: wpp_parse_rates(pkg))
    guarded(out, "tile_parallel_parse", lambda: tile_parse_rates(pkg))
    guarded(out, "colour_kernel_standalone", lambda: colour_standalone(torch, pkg, gb, st))
    guarded(out, "kernels_grouped_streams", lambda: grouped_streams(torch, gb, st))
    guarded(out, "device_inclusive", lambda: device_inclusive(torch, pkg, dev, gb, st, stream_b))
    guarded(out, "end_to_end", lambda: end_to_end_single(pkg, kept[0]))
    guarded(out, "end_to_end_pipelined", lambda: end_to_end_pipelined(pkg, kept))
    guarded(out, "plugin_path", lambda: plugin_path(pkg, kept[0]))
    if args.cpu_seconds > 0:
        guarded(out, "cpu_baseline", lambda: cpu_baseline(kept, args.cpu_seconds, 1))
        guarded(out, "cpu_baseline_all_cores", lambda: cpu_baseline(kept, args.cpu_seconds, effective_cpus()))
    # free the headline batch before the other workloads are built
    gb.batch.close()
    gb.images.clear()
    torch.cuda.empty_cache()
    if all_blobs:
        guarded(out, "config2_no_vui", lambda: no_vui_leg(torch, pkg, dev, st, all_blobs, out.get("ms_per_step")))
        guarded(out, "config3_per_gpu_share", lambda: share_leg(torch, pkg, dev, st, all_blobs[:SHARE_IMAGES]))
        guarded(out, "config2_single_image", lambda: single_image_leg(out, torch, pkg, dev, st, all_blobs[0]))
    guarded(out, "real_content", lambda: real_content(torch, pkg, dev, st))
    guarded(out, "real_content_256_pictures", lambda: real_content(torch, pkg, dev, st, n=256))  # (a mid-size batch: k_chain's ring cut)
    guarded(out, "config4_422_10bit_rgb48", lambda: config4(torch, pkg, dev, st))
    guarded(out, "config5_16384_grid", lambda: config5_single(torch, pkg, dev, st))
    guarded(out, "hdr10_420_grid_rgb24", lambda: hdr10_leg(torch, pkg, dev, st))


SHARE_IMAGES = 128  # BASELINE config 3: 1024 images over 8 GPUs


def single_image_leg(out, torch, pkg, dev, st, blobs):
    """BASELINE config 2 as it is written: ONE 12 MP grid (48 tiles) on one MI355X.  K = the kernels of that one image (command streams
    resident), D = + the H2D of its command streams, E = the caller's clock (hm_decode_item: box parsing + entropy decode on the host
    threads + H2D + kernels + D2H, from the end_to_end leg); the stage roofline on SURVEY 8(d)'s bytes.  48 tiles are the few-pictures
    regime of the chain kernel: the time is the wavefront of ONE tile (16 x 16 CTUs), not the device's throughput."""
    gb, res = variant_batch(torch, pkg, dev, st, [blobs], nclx=(1, 1, 6), steps=20, warmup=3)
    stream_b, sample_b, level_b, resid_b = gb.batch.algorithmic_bytes4()
    recon_ms = sum(v for k, v in res["kernels_ms"].items() if k in ("k_residual", "k_chain"))
    tail_ms = sum(v for k, v in res["kernels_ms"].items() if k not in ("k_residual", "k_chain"))
    stage_b = stream_b + sample_b
    res["roofline_stage"] = {"bound": "hbm", "achieved": round(stage_b / recon_ms / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": round(stage_b / recon_ms / 1e6 / HBM_PEAK_GBPS, 5), "bytes": int(stage_b),
                             "note": "command stream + 1.5 B/px over k_residual + k_chain; latency-bound at 48 tiles (one tile's wavefront: ~36 CTU steps since r06's early CTU start, 46 before)"}
    res["recon_ms"] = round(recon_ms, 4)
    res["tail_ms"] = round(tail_ms, 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gb.batch.upload(st)
        gb.step(st)
    torch.cuda.synchronize()
    d_ms = (time.perf_counter() - t0) / 10 * 1e3
    res["K"] = {"ms": res["ms_per_step"], "MP_per_s": res["MP_per_s"]}
    res["D"] = {"ms": round(d_ms, 4), "MP_per_s": round(MP_PER_IMAGE / d_ms * 1e3, 1), "command_stream_bytes": int(stream_b)}
    e = out.get("end_to_end", {})
    best = min((v for k, v in e.items() if k.startswith("host_threads_")), key=lambda v: v["ms_per_image"], default=None) if isinstance(e, dict) else None
    if best:
        res["E"] = {"ms": best["ms_per_image"], "MP_per_s": best["MP_per_s"], "what": "hm_decode_item, the best of the end_to_end leg's thread counts"}
    gb.batch.close()
    gb.images.clear()
    torch.cuda.empty_cache()
    return res


def variant_batch(torch, pkg, dev, st, blobs_by_image, nclx, steps=10, warmup=2):
    """K clock of a batch of 12 MP grids built from parsed tiles (the headline's workload with another tile profile / count)"""
    gb = GridBatch(pkg, dev, GRID_COLS, GRID_ROWS, TILE, OUT_W, OUT_H, nclx=nclx)
    for blobs in blobs_by_image:
        gb.add_image(blobs)
    gb.finish(st)
    for _ in range(warmup):
        gb.step(st)
    elapsed, avg_ms = timed_steps(torch, gb, st, steps)
    gb.batch.check()
    kernels, _, _, _ = kernel_table(gb, avg_ms)
    n = len(blobs_by_image)
    res = {"MP_per_s": round(n * MP_PER_IMAGE * steps / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 4), "images_per_step": n,
           "tiles_per_step": n * GRID_COLS * GRID_ROWS, "tail_fused": bool(gb.batch.tail_fused()),
           "kernels_ms": {k: v["ms_per_step"] for k, v in kernels.items()}}
    return gb, res


def no_vui_leg(torch, pkg, dev, st, all_blobs, headline_ms):
    """SURVEY 8(d) config 2, second variant: tiles WITHOUT a full-range colour description - the class of the reference's own
    examples/example.heic.  The decoder plugin reports such a tile as limited range (decoder_libde265.cc:339-362: VUI defaults,
    matrix 2), and decode_and_paste_tile_image rescales every sample to full range while it pastes (context.cc:2504-2528).
    Image 0 of the leg is 48 freshly synthesised tiles without a VUI, checked bit for bit against the CPU flow (oracle and real
    libde265 tiles, the oracle's float paste); the other images are the headline's tiles pasted under the same limited-range
    profile (a tile item's colr box overrides the VUI: context.cc:1840-1850) - the kernels' work does not depend on which."""
    import numpy as np
    import orc
    NT = GRID_COLS * GRID_ROWS
    fresh = list(make_streams(pkg.capi, (1300000 + i for i in range(NT)), vui=0))
    images = [[b for _, b in fresh]] + list(all_blobs[1:])
    gb, res = variant_batch(torch, pkg, dev, st, images, nclx=(1, 0, 2))
    got = gb.images[0]["rgb"].cpu().numpy()
    strides = (gb.ys, gb.cs, gb.os)
    ok = True
    for use_ref in [False] + ([True] if orc.have_ref() else []):
        exp = cpu_grid_image([d for d, _ in fresh], [b for _, b in fresh], GRID_COLS, GRID_ROWS, TILE, OUT_W, OUT_H, strides, use_ref)
        ok = ok and bool(np.array_equal(got[:OUT_H, :OUT_W * 3], exp[:OUT_H, :OUT_W * 3]))
    res["parity"] = ("bit-exact vs oracle" + (" and reference libde265" if orc.have_ref() else "") + " (image 0: 48 tiles without a VUI)") if ok else "MISMATCH"
    if headline_ms:
        res["ratio_to_headline"] = round(res["ms_per_step"] / headline_ms, 4)
    res["note"] = "the headline's step with every tile rescaled limited -> full range in the paste (part of the fused tail kernel since r05)"
    gb.batch.close()
    gb.images.clear()
    torch.cuda.empty_cache()
    if not ok:
        raise RuntimeError("config2_no_vui: GPU RGB != CPU flow")
    return res


def share_leg(torch, pkg, dev, st, blobs):
    """BASELINE config 3 as it lands on ONE of 8 GPUs: 128 of the 1024 images = 6144 tiles per step (the default step keeps 384
    images per GPU).  6144 pictures are 1.2 rounds of the 5120 wave-per-picture chains the device holds at once."""
    gb, res = variant_batch(torch, pkg, dev, st, blobs, nclx=(1, 1, 6))
    recon = sum(v for k, v in res["kernels_ms"].items() if k in ("k_residual", "k_chain"))
    res["recon_us_per_tile"] = round(recon * 1e3 / res["tiles_per_step"], 4)
    gb.batch.close()
    gb.images.clear()
    torch.cuda.empty_cache()
    return res


def share_leg_all_ranks(out, args, torch, pkg, dev, st, blobs, dist, rank, world):
    """N > 1: the config-3 share (128 images per GPU) on every rank at once; rank 0 reports the aggregate over the slowest rank"""
    cpu = "cpu" if args.dist_backend != "nccl" else dev
    ms, ok = 0.0, 1
    try:
        dist.barrier()
        r = share_leg(torch, pkg, dev, st, blobs)
        ms = r["ms_per_step"]
    except Exception as ex:  # noqa: BLE001
        ok = 0
        print(f"[bench rank {rank}] config3_per_gpu_share: {ex}", file=sys.stderr, flush=True)
    t = torch.tensor([ms, ok], dtype=torch.float64, device=cpu)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    if rank == 0:
        rows = [[float(x) for x in p.tolist()] for p in parts]
        if all(r[1] for r in rows):
            slow = max(r[0] for r in rows)
            out["config3_per_gpu_share_all_ranks"] = {"MP_per_s": round(world * len(blobs) * MP_PER_IMAGE / slow * 1e3, 1), "images_per_gpu": len(blobs),
                                                      "ms_per_step_slowest_rank": round(slow, 4), "per_rank_ms": [round(r[0], 4) for r in rows],
                                                      "note": "BASELINE config 3's own load per GPU (1024 images / 8); K clock, all ranks at once"}
        else:
            out["config3_per_gpu_share_all_ranks"] = {"failed_ranks": [i for i, r in enumerate(rows) if not r[1]]}


def host_parse_rate(pkg, streams):
    dt = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for data in streams:
            pkg.capi.parse_hevc(data)
        dt = min(dt, time.perf_counter() - t0)
    return {"MP_per_s_per_core": round(len(streams) * TILE * TILE / 1e6 / dt, 1),
            "note": "hm_hevc_parse (CABAC -> command stream), 1 thread, best of 5 passes over the tiles of one image, outside the timed region"}


def colour_standalone(torch, pkg, gb, st):
    """the YCbCr 4:2:0 -> RGB24 kernel alone (k_ycbcr420_int over the batch's canvases, hm_colour_convert_batch): the
    north star's colour-kernel roofline figure.  The default hot path no longer launches it (k_tail420 converts while it
    filters), so it is timed here beside the headline, with events on the launch stream (torch's current stream)."""
    L = pkg.lib()
    L.hm_colour_convert_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    n = len(gb.images)

    def run():
        pkg.capi.check(L.hm_colour_convert_batch(C.byref(gb.desc), n, *gb.p, st))
    for _ in range(2):
        run()
    evs = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    px = gb.pixels()
    return {"kernel": KERNEL_NAMES[3], "ms_per_step": round(ms, 4), "images": n,
            "GBps": round(4.5 * px / ms / 1e6, 1), "frac_of_hbm_peak": round(4.5 * px / ms / 1e6 / HBM_PEAK_GBPS, 4),
            "read_only_GBps": round(1.5 * px / ms / 1e6, 1), "read_only_frac_of_hbm_peak": round(1.5 * px / ms / 1e6 / HBM_PEAK_GBPS, 4),
            "note": "1.5 B/px read + 3 B/px written (SURVEY 8d: both conventions); part of the hot path only when the tail is not fused"}


def grouped_streams(torch, gb, st):
    """K clock with hm_batch_set_concurrency: the images of the step in 2 / 3 / 4 groups, each group's reconstruction and
    fused tail on a stream of its own.  A side clock: kernels that run side by side have no separable launch times, so
    `value` and `roofline` stay on the single-stream launches."""
    B = len(gb.images)
    res = {}
    sample = list(range(0, B, max(1, B // 16)))
    gb.batch.set_concurrency(0)
    gb.step(st)  # (the legs before this one may have written other things into the output buffers)
    torch.cuda.synchronize()
    want = {i: gb.images[i]["rgb"].clone() for i in sample}  # what the (parity-gated) single-stream step produces
    same = True
    for groups in (2, 3, 4):
        gb.batch.set_concurrency(groups)
        for i in sample:
            gb.images[i]["rgb"].zero_()
        for _ in range(2):
            gb.step(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            gb.step(st)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        res[f"{groups}_groups"] = {"ms_per_step": round(ms, 3), "MP_per_s": round(B * MP_PER_IMAGE / ms * 1e3, 1)}
        same = same and all(torch.equal(gb.images[i]["rgb"], want[i]) for i in sample)
    gb.batch.set_concurrency(0)
    res["outputs_equal_single_stream"] = f"{'yes' if same else 'NO'} ({len(sample)} images compared in every mode)"
    res["note"] = "same step as the headline, images split over streams inside hm_batch_execute (opt-in API); never `value`"
    return res


def device_inclusive(torch, pkg, dev, gb, st, stream_b):
    pkg.lib().hm_colour_convert_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    """clock (D) of SURVEY 8d: H2D of the command streams + kernels, result on the device.  Two figures: everything on
    one stream (the H2D in front of the kernels), and chunked with the H2D of chunk i+1 on a copy stream under the
    kernels of chunk i."""
    B = len(gb.images)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        gb.batch.upload(st)
        gb.step(st)
    torch.cuda.synchronize()
    serial_ms = (time.perf_counter() - t1) / 3 * 1e3
    res = {"serial": {"ms_per_step": round(serial_ms, 3), "MP_per_s": round(B * MP_PER_IMAGE / serial_ms * 1e3, 1)},
           "command_stream_bytes": int(stream_b), "command_stream_bytes_per_pixel": round(stream_b / (B * GRID_COLS * GRID_ROWS * TILE * TILE), 3)}
    copy = torch.cuda.Stream(device=dev)
    for chunks in (3, 4, 6):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            gb.batch.upload_execute(3, chunks, copy.cuda_stream, st)  # (the conversion is attached to the batch: part of every chunk)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / 3 * 1e3
        res[f"overlapped_{chunks}_chunks"] = {"ms_per_step": round(ms, 3), "MP_per_s": round(B * MP_PER_IMAGE / ms * 1e3, 1)}
    res["note"] = ("H2D of the command streams (pinned staging) + all kernels, RGB left on the device; serial = one stream, overlapped = "
                   "hm_batch_upload_execute (copy stream feeding the compute stream chunk by chunk, chunks growing 1:2:3:...); 1 GPU")
    gb.batch.upload(st)  # back to the resident state
    torch.cuda.synchronize()
    return res


def end_to_end_single(pkg, tiles):
    """Clock (E) of SURVEY 8d for ONE image at a time: a 12 MP grid .heic through hm_decode_item = box parsing + host
    entropy decode (threads) + H2D + kernels + D2H into a libheif-layout host plane, strictly in sequence."""
    import heifwriter
    import pipeline
    data = heifwriter.write_heic(tiles, (TILE, TILE), grid=(GRID_ROWS, GRID_COLS, OUT_W, OUT_H))
    res = {}
    f = pipeline.HeifFile(pkg.lib(), data)
    try:
        for threads in sorted({1, 8, min(48, effective_cpus())}):
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


def plugin_path(pkg, tiles):
    """The decoder-plugin boundary driven the way the reference drives it for a grid (context.cc:2361-2401, 1787-1835;
    decoder_libde265.cc:306-369): one decoder instance per tile - new_decoder, push_data, decode_image, free_decoder - from
    a window of 8 concurrent C++ tasks, one 12 MP grid = 48 tiles.  Behind decode_image the calls meet in the device's shared worker
    (csrc/picture.cpp) and run as one GPU batch.  Beside it: hm_decode_item on the same tiles as a .heic (planar output,
    8 host threads), which also pastes."""
    import heifwriter
    import numpy as np
    import orc
    import pipeline
    import pluginapi
    api = pluginapi.load_api(pkg)
    plugin_ptr = api.hm_get_decoder_plugin()
    hm = pkg.lib()
    tiles = [bytes(t) for t in tiles]

    def run(window=8):
        return pluginapi.drive_grid(plugin_ptr, tiles, window)

    imgs = run()  # warm-up, and a check of tile 0 against the oracle
    stride = C.c_int()
    ptr = api.heif_image_get_plane_readonly(imgs[0], 0, C.byref(stride))
    got = np.ctypeslib.as_array(ptr, shape=(TILE, stride.value))[:, :TILE].copy()
    exp, _ = orc.oracle_decode(pkg.capi.parse_hevc(tiles[0]), 3, crop=True)
    if not np.array_equal(got.astype(np.uint16), exp[0]):
        raise RuntimeError("plugin path: tile 0 differs from the oracle")
    for im in imgs:
        api.heif_image_release(im)

    def best_plugin(window):
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            imgs = run(window)
            best = min(best, time.perf_counter() - t0)
            for im in imgs:
                api.heif_image_release(im)
        return best

    data = heifwriter.write_heic(tiles, (TILE, TILE), grid=(GRID_ROWS, GRID_COLS, OUT_W, OUT_H))
    f = pipeline.HeifFile(hm, data)

    def best_item(threads):
        f.decode(f.primary(), 0, threads=threads, copy=False)
        item = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            f.decode(f.primary(), 0, threads=threads, copy=False)
            item = min(item, time.perf_counter() - t0)
        return item

    try:
        best, item = best_plugin(8), best_item(8)
        # the window is the caller's choice (heif_context_set_threads -> m_max_decoding_threads, heif.cc:499-513): wider
        # windows next to hm_decode_item with as many host threads
        wider = {}
        for w in (16, 48):
            b, i = best_plugin(w), best_item(w)
            wider[str(w)] = {"ms_per_12MP_grid": round(b * 1e3, 2), "hm_decode_item_ms": round(i * 1e3, 2), "ratio_to_hm_decode_item": round(b / i, 2)}
    finally:
        f.close()
    return {"ms_per_12MP_grid": round(best * 1e3, 2), "MP_per_s": round(MP_PER_IMAGE / best, 1), "tiles": len(tiles), "threads": 8,
            "hm_decode_item_ms": round(item * 1e3, 2), "ratio_to_hm_decode_item": round(best / item, 2), "wider_windows": wider,
            "note": "48 decoder instances (new_decoder / push_data / decode_image / free_decoder) from a window of 8 async C++ tasks - the reference's caller, "
                    "context.cc:2361-2401 (tests/synth/plugin_driver.cpp) -, best of 5; host entropy decode on the calling threads, the GPU work of "
                    "concurrent calls coalesced by the shared device worker; tile 0 checked against the oracle; wider_windows: the same with 16 / 48 "
                    "tasks (and hm_decode_item with 16 / 48 host threads)"}


def end_to_end_pipelined(pkg, kept, threads=None, cpus=None, device=-1):
    """Clock (E) at throughput: hm_pipeline_* over >= 256 different 12 MP .heic files, all host cores parsing, the images'
    device work on their own streams (parse of k+1 || kernels of k || D2H of k-1); every output is hash-checked against
    the image-at-a-time path.  threads / cpus / device: the crew of ONE rank of a multi-GPU run (its share of the usable
    CPUs, pinned to its slice of the affinity mask, feeding its own GPU)."""
    import heifwriter
    import numpy as np
    import pipeline
    hm = pkg.lib()
    files = [heifwriter.write_heic(t, (TILE, TILE), grid=(GRID_ROWS, GRID_COLS, OUT_W, OUT_H)) for t in kept]
    ncpu = effective_cpus()
    threads = threads or max(1, min(ncpu, 192))
    depth = 16

    def fnv(arr, stride):
        return pipeline.survey_fnv(arr, stride, OUT_W * 3, OUT_H)

    # expected hashes: the synchronous path, a sample of 16 files (the whole set would take longer than the measurement)
    sample = list(range(0, len(files), max(1, len(files) // 16)))[:16]
    expected = {}
    for i in sample:
        f = pipeline.HeifFile(hm, files[i])
        planes, meta = f.decode(f.primary(), 10, threads=min(48, threads))
        expected[i] = fnv(planes[0], meta["stride"][0])
        f.close()
    pl = pipeline.Pipeline(hm, 10, host_threads=threads, max_in_flight=depth, device=device, cpus=cpus)
    checked = 0
    import collections
    order = collections.deque()  # results come back in submission order

    def take(check_hashes):
        nonlocal checked
        i = order.popleft()
        want = check_hashes and i in expected
        tag, status, arr, meta = pl.next(copy=want)
        if tag != i:
            raise RuntimeError(f"result {tag} out of order (expected {i})")
        if status:
            raise RuntimeError(f"image {tag}: status {status}: {meta}")
        if want:
            if fnv(arr, meta["stride"]) != expected[i]:
                raise RuntimeError(f"image {i}: pipelined output differs from hm_decode_item")
            checked += 1

    try:
        for rnd in range(2):  # round 0 warms the pools and checks the hashes, round 1 is timed
            t0 = time.perf_counter()
            for i, data in enumerate(files):
                while not pl.submit(data, i):
                    take(rnd == 0)
                order.append(i)
            while order:
                take(rnd == 0)
            dt = time.perf_counter() - t0
    finally:
        pl.close()
    n = len(files)
    return {"images": n, "seconds": dt, "host_threads": threads, "cpu_set": list(cpus) if cpus else None, "host_cpus_usable": ncpu, "host_cpus_visible": os.cpu_count(), "max_in_flight": depth,
            "ms_per_image": round(dt / n * 1e3, 3), "MP_per_s": round(n * MP_PER_IMAGE / dt, 1),
            "outputs_hash_checked": checked,
            "note": "hm_pipeline_*: .heic bytes in, RGB24 in pinned host memory out; box parsing + CABAC on the host crew, H2D, kernels, "
                    "D2H overlapped across images; bounded by the host entropy decode: usable CPUs (cgroup quota) x host_entropy_decode rate; "
                    "the PCIe D2H ceiling is ~ 18 GP/s (36.6 MB per image)"}


def cpu_baseline(kept, budget_s, threads):
    """The reference CPU path timed on this host (oracle/cpu_baseline.c: libde265 of the reference per tile, one tile per
    task like heif_context_set_threads, paste + colour by the oracle), on a bounded sample of the same 12 MP images."""
    import orc
    so = os.path.join(ROOT, "oracle", "_ref", "libcpu_baseline.so")
    if not os.path.exists(so):
        raise RuntimeError("oracle/_ref/libcpu_baseline.so not built (reference sources absent at build time)")
    orc.load()
    lib = C.CDLL(so)
    lib.cpu_baseline_run.restype = C.c_double
    lib.cpu_baseline_run.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]

    def run(n):
        tiles = [t for img in kept[:n] for t in img]
        arr = (C.c_char_p * len(tiles))(*tiles)
        sizes = (C.c_size_t * len(tiles))(*[len(t) for t in tiles])
        h = C.c_uint64()
        dt = lib.cpu_baseline_run(arr, sizes, n, GRID_COLS, GRID_ROWS, TILE, OUT_W, OUT_H, threads, C.byref(h))
        if dt < 0:
            raise RuntimeError("cpu_baseline_run failed")
        return dt

    n = 1 if threads == 1 else min(len(kept), max(2, threads // 8))
    dt = run(n)  # calibration
    n2 = int(max(1, min(len(kept), n * budget_s / max(dt, 1e-3))))
    if n2 > n:
        n, dt = n2, run(n2)
    return {"value": round(n * MP_PER_IMAGE / dt, 2), "unit": "MP/s", "cores": threads, "kind": "reference",
            "sample": f"{n} x 12 MP grid image (48 tiles each) in {dt:.1f} s: libde265 of /root/reference built by oracle/Makefile (SSE4.1/AVX2 kernels) "
                      f"for the tile decode incl. entropy decode, oracle C restatement for paste + colour (libheif is unbuildable without "
                      f"cmake); one tile per task on {threads} thread(s); {os.cpu_count()} host cpus visible"}


def real_content(torch, pkg, dev, st, n=32):
    """The three real 1080p intra frames of the reference's test material (third-party/libde265/testfile, committed
    as tests/data/*.hevc; CTB 64, WPP, SAO, SDH, transform skip), 32 copies of each in one batch: per-kernel ms / MP on
    real-encoder block statistics next to the random-syntax headline."""
    capi = pkg.capi
    res = {}
    for name in ("basketball_1080p_qp32", "basketball_1080p_qp25", "basketball_1080p_qp1"):
        path = os.path.join(ROOT, "tests", "data", name + ".hevc")
        if not os.path.exists(path):
            continue
        blob = capi.parse_hevc(open(path, "rb").read())
        gb = GridBatch(pkg, dev, 1, 1, 0, 1920, 1080)
        for _ in range(n):
            gb.add_image([blob])
        gb.finish(st)
        for _ in range(2):
            gb.step(st)
        elapsed, avg_ms = timed_steps(torch, gb, st, 5)
        mp = n * 1920 * 1080 / 1e6
        stream_b, sample_b = gb.batch.algorithmic_bytes()
        res[name] = {"MP_per_s": round(mp * 5 / elapsed, 1), "command_stream_bytes_per_pixel": round(stream_b / (n * 1920 * 1088), 3),
                     "ms_per_MP": {kernel_names(gb)[q]: round(avg_ms[q] / mp, 5) for q in (4, 0, 1, 2, 3) if kernel_names(gb)[q]}}
        gb.batch.close()
        gb.images.clear()
        torch.cuda.empty_cache()
    res["note"] = f"{n} copies of one 1080p intra picture per batch (2073600 px each), K clock; the headline's synthetic tiles cost the ms/MP of 'kernels' / 4644.9 MP"
    return res


def config4(torch, pkg, dev, st, n=32):
    """a single 2048x1536 10-bit 4:2:2 image (seed 4220010, CTB 32, VUI matrix 9 limited range)
    -> interleaved RRGGBB_LE (6 B/px, float chain); 32 copies per batch, K clock incl. the colour kernel."""
    import synthutil
    capi, L = pkg.capi, pkg.lib()
    W, H = 2048, 1536
    data = synthutil.picture(4220010, width=W, height=H, chroma_format=2, bit_depth=10, log2_ctb=5, qp=30, vui=1, full_range=0, matrix=9, primaries=9)
    blob = capi.parse_hevc(data)
    ys, cs, os_ = L.hm_plane_stride(W, 2), L.hm_plane_stride(W // 2, 2), L.hm_plane_stride(W, 6)
    batch = capi.Batch()
    ims = []
    for _ in range(n):
        y = torch.zeros((H, ys), dtype=torch.uint8, device=dev)
        cb = torch.zeros((H, cs), dtype=torch.uint8, device=dev)
        cr = torch.zeros((H, cs), dtype=torch.uint8, device=dev)
        rgb = torch.zeros((H, os_), dtype=torch.uint8, device=dev)
        d = capi.TileDest()
        d.plane[0], d.plane[1], d.plane[2] = y.data_ptr(), cb.data_ptr(), cr.data_ptr()
        d.pitch[0], d.pitch[1], d.pitch[2] = ys, cs, cs
        d.canvas_width, d.canvas_height = W, H
        batch.add(blob, d)
        ims.append((y, cb, cr, rgb))
    batch.upload(st)
    desc = capi.ColourDesc(W, H, 10, 2, 1, 9, 9, 0, capi.HM_OUT_RRGGBB_LE, ys, cs, cs, os_)
    # the conversion attached to the batch (hm_batch_set_colour): one execute is the whole hot path, and - r04 - this class
    # takes the fused float tail (filters.hip k_tailf: deblocking + SAO + paste + float matrix + repack in one kernel)
    PtrArr = C.c_void_p * n
    ptrs = [PtrArr(*[im[k].data_ptr() for im in ims]) for k in range(4)]
    batch.set_colour(desc, n, *ptrs, 0)
    for _ in range(2):
        batch.execute(3, st)
    batch.set_profiling(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5):
        batch.execute(3, st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k = [0.0, 0.0, 0.0, 0.0, 0.0]
    for i in range(5):
        ms = batch.timings5_ms(i)
        k = [k[q] + ms[q] / 5 for q in range(5)]
    batch.check()
    fused = bool(batch.tail_fused())
    mp = n * W * H / 1e6
    stream_b, sample_b = batch.algorithmic_bytes()
    batch.close()
    if fused:
        kernels = {"k_residual": round(k[4], 3), KERNEL_NAMES[0]: round(k[0], 3), "k_tailf(deblock+sao+paste+float colour)": round(k[2], 3)}
        tail_ms, tail_bytes = k[2], 10.0 * n * W * H  # 4 B/px of samples in + 6 B/px of pixels out
    else:
        kernels = {"k_residual": round(k[4], 3), KERNEL_NAMES[0]: round(k[0], 3), "k_deblock": round(k[1], 3), "k_sao_paste": round(k[2], 3), "k_ycbcr_float(colour)": round(k[3], 3)}
        tail_ms, tail_bytes = k[3], 10.0 * n * W * H
    return {"MP_per_s": round(mp * 5 / dt, 1), "images_per_step": n, "tail_fused": fused,
            "kernels_ms_per_step": kernels,
            "tail_GBps": round(tail_bytes / tail_ms / 1e6, 1) if tail_ms > 0 else None, "tail_frac_of_hbm_peak": round(tail_bytes / tail_ms / 1e6 / HBM_PEAK_GBPS, 4) if tail_ms > 0 else None,
            "command_stream_bytes_per_pixel": round(stream_b / (n * W * H), 3),
            "note": "10-bit 4:2:2 2048x1536 (seed 4220010) -> RRGGBB_LE, K clock; the tail reads 4 B/px of samples and writes 6 B/px of pixels (fused: once each)"}


def hdr10_leg(torch, pkg, dev, st, n=96):
    """The headline's 12 MP grid with 10-bit 4:2:0 tiles (the class of HDR photographs) -> RGB24 along the chain the reference's search
    picks for it (the shift to 8 bits and the integer 4:2:0 operation, fused: k_tail420's 16-bit instantiation since r06, k_tailf before):
    n copies of one image of 48 tiles per batch, K clock; the first image against the CPU flow (oracle executors, oracle paste, the searched chain - tests/pipeline.py: the flow of tests/test_configs_gpu.py)."""
    import numpy as np
    import pipeline
    capi, L = pkg.capi, pkg.lib()
    W, H, cols, rows, tile = OUT_W, OUT_H, GRID_COLS, GRID_ROWS, TILE
    datas = [tile_stream(9100000 + k, bit_depth=10) for k in range(cols * rows)]
    blobs = [capi.parse_hevc(d) for d in datas]
    ys, cs, os_ = L.hm_plane_stride(W, 2), L.hm_plane_stride((W + 1) // 2, 2), L.hm_plane_stride(W, 3)
    ch = (H + 1) // 2
    batch = capi.Batch()
    ims = []
    for _ in range(n):
        im = (torch.zeros((H, ys), dtype=torch.uint8, device=dev), torch.zeros((max(64, ch), cs), dtype=torch.uint8, device=dev),
              torch.zeros((max(64, ch), cs), dtype=torch.uint8, device=dev), torch.zeros((H, os_), dtype=torch.uint8, device=dev))
        for t in range(cols * rows):
            d = capi.TileDest()
            d.plane[0], d.plane[1], d.plane[2] = im[0].data_ptr(), im[1].data_ptr(), im[2].data_ptr()
            d.pitch[0], d.pitch[1], d.pitch[2] = ys, cs, cs
            d.canvas_width, d.canvas_height = W, H
            d.x0, d.y0 = (t % cols) * tile, (t // cols) * tile
            d.tile_has_nclx, d.tile_full_range, d.tile_matrix = 1, 1, 6  # (full range: nothing is rescaled in the paste)
            batch.add(blobs[t], d)
        ims.append(im)
    batch.upload(st)
    desc = capi.ColourDesc(W, H, 10, 1, 0, 6, 1, 1, capi.HM_OUT_RGB, ys, cs, cs, os_)
    PtrArr = C.c_void_p * n
    ptrs = [PtrArr(*[im[k].data_ptr() for im in ims]) for k in range(4)]
    batch.set_colour(desc, n, *ptrs, 0)
    for _ in range(2):
        batch.execute(3, st)
    batch.set_profiling(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5):
        batch.execute(3, st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k = [0.0, 0.0, 0.0, 0.0, 0.0]
    for i in range(5):
        ms = batch.timings5_ms(i)
        k = [k[q] + ms[q] / 5 for q in range(5)]
    batch.check()
    fused = bool(batch.tail_fused())
    got = ims[n // 2][3].cpu().numpy()[:H, :W * 3]
    batch.close()
    exp, _, _ = pipeline.cpu_decode(L, datas, tile, tile, W, H, cols, True, capi.HM_OUT_RGB, decoder="oracle")
    if not np.array_equal(got, exp[:H, :W * 3]):  # (a rate for a wrong picture is no rate: guarded() reports the error instead)
        raise RuntimeError("hdr10_420_grid_rgb24: the fused tail's image differs from the oracle flow")
    parity = "bit-exact vs the oracle flow"
    mp = n * W * H / 1e6
    if fused:
        kernels = {"k_residual": round(k[4], 3), KERNEL_NAMES[0]: round(k[0], 3), "k_tail420<16-bit>(deblock+sao+paste+shift+integer colour)": round(k[2], 3)}
        tail_ms = k[2]
    else:
        kernels = {"k_residual": round(k[4], 3), KERNEL_NAMES[0]: round(k[0], 3), "k_deblock": round(k[1], 3), "k_sao_paste": round(k[2], 3), "k_ycbcr_float(colour)": round(k[3], 3)}
        tail_ms = k[1] + k[2] + k[3]
    tail_bytes = 6.0 * n * W * H  # 3 B/px of samples in (1.5 samples of 2 bytes) + 3 B/px of pixels out
    return {"MP_per_s": round(mp * 5 / dt, 1), "images_per_step": n, "tiles_per_step": n * cols * rows, "tail_fused": fused, "kernels_ms_per_step": kernels,
            "tail_GBps": round(tail_bytes / tail_ms / 1e6, 1) if tail_ms > 0 else None, "tail_frac_of_hbm_peak": round(tail_bytes / tail_ms / 1e6 / HBM_PEAK_GBPS, 4) if tail_ms > 0 else None,
            "parity": parity,
            "note": "12 MP grids of 10-bit 4:2:0 tiles (CTB 32, full range, matrix 6) -> RGB24, K clock; the tail reads 3 B/px of samples and writes 3 B/px of pixels"}


def grid_tile_seeds(tile_rows, cols):
    return [5000000 + r * cols + c for r in tile_rows for c in range(cols)]


def config5_single(torch, pkg, dev, st):
    """SURVEY 8d config 5 on one GPU: ONE 16384x16384 grid (32x32 tiles of 512x512, seeds 5000000+i) -> RGB24, K clock; beside it (r06) the
    caller's clock E for the same grid as a .heic through hm_decode_item (box parsing + entropy decode on the host threads + the grid's slabs of tile
    rows under it + D2H into the 805 MB host plane)."""
    capi = pkg.capi
    gb = GridBatch(pkg, dev, 32, 32, TILE, 16384, 16384)
    made = list(make_streams(capi, grid_tile_seeds(range(32), 32), keep_data=True))
    datas = [d for d, _ in made]
    blobs = [b for _, b in made]
    del made
    gb.add_image(blobs)
    gb.finish(st)
    for _ in range(2):
        gb.step(st)
    elapsed, avg_ms = timed_steps(torch, gb, st, 10)
    kernels, alg, _, _ = kernel_table(gb, avg_ms)
    mp = 16384 * 16384 / 1e6
    res = {"MP_per_s": round(mp * 10 / elapsed, 1), "ms_per_grid": round(elapsed / 10 * 1e3, 3), "kernels": kernels,
           "note": "one 268 MP grid = 1024 tiles per launch (a third of the headline's tiles in flight: the reconstruction kernel's wavefront tail weighs more)"}
    gb.batch.close()
    gb.images.clear()
    del blobs
    torch.cuda.empty_cache()
    try:  # (the E clock is a side figure: its failure must not take the K figure with it)
        import heifwriter
        import pipeline
        f = pipeline.HeifFile(pkg.lib(), heifwriter.write_heic(datas, (TILE, TILE), grid=(32, 32, 16384, 16384)))
        threads = min(48, effective_cpus())
        try:
            f.decode(f.primary(), 10, threads=threads, copy=False)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                f.decode(f.primary(), 10, threads=threads, copy=False)
                ts.append(time.perf_counter() - t0)
        finally:
            f.close()
        res["E"] = {"ms": round(min(ts) * 1e3, 1), "MP_per_s": round(mp / min(ts), 1), "host_threads": threads,
                    "what": "hm_decode_item on the grid as a .heic, best of 3: host entropy decode of 1024 tiles with the grid's slabs of tile rows (kernels + D2H) under it"}
    except Exception as e:  # noqa: BLE001
        res["E"] = {"error": str(e)[:200]}
    return res


# --------------------------------------------------------------------------------------------------------------
# grid mode: ONE 16384^2 grid over the ranks, RGB row slabs gathered with one collective
# --------------------------------------------------------------------------------------------------------------

def run_grid(args, torch, pkg, dev, dist, rank, world, st, shared):
    """ONE 16384^2 grid, tile rows sharded over the ranks (shard.row_slabs).  r06: the gather is part of the timed loop and runs
    UNDER the decode - two image buffers per rank: while grid i + 1 is decoded into one, the slabs of grid i travel from the other
    (a second stream behind an event per chunk of tile rows; point-to-point, no padding, the root receiving straight into the
    rows of its final image and decoding its own slab in place: shard.SlabGather).  `value` at N > 1 is that pipelined rate;
    beside it the K-only rate (no gather), the latency of one grid (decode, then gather) and - the alternative SURVEY 8(e) prefers
    for a caller whose planes live in host memory (heif_image_get_plane*) - every rank copying its rows into ONE pinned host image
    (what hm_decode_item_devices does in-process), so that the first run on an 8-GPU box decides between the two."""
    capi = pkg.capi
    sh = pkg.shard
    COLS = ROWS = 32
    W = H = 16384
    nccl = dist is not None and args.dist_backend == "nccl"
    slabs = sh.row_slabs(ROWS, world)
    r0, nr = slabs[rank]
    y0, y1 = sh.slab_pixel_rows(r0, nr, TILE, H)
    chunk_rows = args.grid_chunk_rows
    gat = sh.SlabGather(slabs, TILE, H, chunk_tile_rows=chunk_rows, dst=0, stage_through_host=not nccl) if dist else None
    chunks = sh.slab_chunks(r0, nr, chunk_rows)  # the decode is cut like the transfer: a launch + an event per chunk
    out_stride = pkg.lib().hm_plane_stride(W, 3)
    NB = 2 if dist else 1
    # the image buffers: the root's are whole images (its own rows decoded in place, the others' received into theirs), the other
    # ranks' hold their slab
    if rank == 0:
        images = [torch.zeros((H, out_stride), dtype=torch.uint8, device=dev) for _ in range(NB)]
        local = [im[y0:y1] for im in images]
    else:
        images = None
        local = [torch.zeros((max(y1 - y0, 64), out_stride), dtype=torch.uint8, device=dev) for _ in range(NB)]
    # one batch per (buffer, chunk): the destination pointers of a batch's tiles are fixed when they are added
    batches = []
    for b in range(NB):
        per_chunk = []
        for c0, cn in chunks:
            cy0, cy1 = sh.slab_pixel_rows(c0, cn, TILE, H)
            if cy1 <= cy0:
                continue
            gbc = GridBatch(pkg, dev, COLS, cn, TILE, W, cy1 - cy0)
            gbc.add_image([bl for _, bl in make_streams(capi, grid_tile_seeds(range(c0, c0 + cn), COLS), keep_data=False)], rgb=local[b][cy0 - y0:cy1 - y0])
            gbc.finish(st)
            per_chunk.append(gbc)
        batches.append(per_chunk)
    comm = torch.cuda.Stream(device=dev) if dist else None
    events = [[torch.cuda.Event() for _ in per_chunk] for per_chunk in batches]

    def decode(b):
        for gbc, ev in zip(batches[b], events[b]):
            gbc.step(st)
            ev.record(torch.cuda.current_stream())

    def start_gather(b):
        """hand buffer b's rows to the gather (asynchronous); returns what finish_gather waits for"""
        if dist is None:
            return None
        with torch.cuda.stream(comm):
            if rank == 0:
                return gat.post_recvs(images[b])
            evs = events[b]
            return gat.send(local[b], chunk_ready=(lambda i: comm.wait_event(evs[min(i, len(evs) - 1)])) if evs else None)

    def finish_gather(works):
        if works is None:
            return
        if rank == 0:
            with torch.cuda.stream(comm):
                gat.wait(works)
        else:
            for w in works:
                w.wait()

    decode(0)
    torch.cuda.synchronize()
    # ---- self-check: the gathered image equals the one-rank decode, bit for bit ----
    works = start_gather(0)
    finish_gather(works)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ok = 1
    check = "skipped"
    if rank == 0 and not args.no_parity:
        one = GridBatch(pkg, dev, COLS, ROWS, TILE, W, H)
        one.add_image([b for _, b in make_streams(capi, grid_tile_seeds(range(ROWS), COLS), keep_data=False)])
        one.finish(st)
        one.step(st)
        torch.cuda.synchronize()
        ref = one.images[0]["rgb"][:H]
        ok = int(torch.equal(images[0][:, :W * 3], ref[:, :W * 3]))
        check = "gathered slabs == one-rank decode (bit-exact)" if ok else "MISMATCH"
        one.batch.close()
        del one, ref
        torch.cuda.empty_cache()
    if dist:
        t = torch.tensor([ok], dtype=torch.int32, device=dev if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    if not ok:
        if rank == 0:
            emit({"error": "grid mode self-check failed: gathered RGB != one-rank decode"})
        raise SystemExit(3)

    def barrier_sync():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()

    # ---- K only: the decode loop without any gather ----
    for _ in range(args.warmup):
        decode(0)
    barrier_sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        decode(i % NB)
    barrier_sync()
    decode_s = time.perf_counter() - t0
    # ---- the pipelined loop: grid i's slabs travel while grid i + 1 is decoded into the other buffer ----
    pending = [None] * NB
    barrier_sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        b = i % NB
        finish_gather(pending[b])  # (the buffer's previous grid has left it)
        decode(b)
        pending[b] = start_gather(b)
    for b in range(NB):
        finish_gather(pending[b])
    barrier_sync()
    piped_s = time.perf_counter() - t0
    # ---- one grid at a time: decode, then gather (the latency a single caller sees) ----
    n_g = max(1, min(args.steps, 5))
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(n_g):
        decode(0)
        finish_gather(start_gather(0))
        barrier_sync()
    serial_s = (time.perf_counter() - t0) / n_g
    # ---- the alternative: every rank copies its rows into ONE pinned host image (shared memory between the ranks' processes) ----
    host = host_gather_leg(args, torch, dist, rank, world, dev, st, H, out_stride, y0, y1, local, decode, barrier_sync, images)
    if dist:
        tt = torch.tensor([decode_s, piped_s, serial_s], dtype=torch.float64, device=dev if nccl else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        decode_s, piped_s, serial_s = (float(x) for x in tt.tolist())
    if rank == 0:
        mp = W * H / 1e6
        k_step = decode_s / args.steps
        per_step = (piped_s if dist else decode_s) / args.steps
        gather_bytes = int(sum(y1_ - y0_ for r in range(1, world) for (y0_, y1_) in (gat.rows[r] if gat else [])) * out_stride)
        out = {"metric": "megapixels/sec HEIC grid->RGB24", "value": round(mp / per_step, 1), "unit": "MP/s", "n_gpus": world,
               "world_size": dist.get_world_size() if dist else 1, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(per_step * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
               "data": "synthetic",
               "config": {"workload": "ONE 16384x16384 HEIC grid (32x32 tiles of 512x512, 8-bit 4:2:0, CTB32) -> RGB24, tile rows sharded over the ranks",
                          "tile_rows_per_rank": [b for _, b in slabs], "chunk_tile_rows": chunk_rows, "chunks_per_rank": [len(sh.slab_chunks(a, b, chunk_rows)) for a, b in slabs],
                          "self_check": check,
                          "timed_region": ("K clock + the gather of the RGB row slabs to rank 0, pipelined: grid i's slabs travel under the decode of grid i + 1 (two buffers per rank); "
                                           "point-to-point, no padding, received straight into the rows of the root's image") if dist else "K clock (one rank: nothing to gather)"},
               "k_only": {"MP_per_s": round(mp / k_step, 1), "ms_per_step": round(k_step * 1e3, 4)},
               "single_grid_latency": {"ms": round(serial_s * 1e3, 3), "MP_per_s": round(mp / serial_s, 1), "what": "decode, then gather, one grid at a time"},
               "gather": {"bytes": gather_bytes, "backend": args.dist_backend if dist else "none",
                          "ms_not_hidden_per_step": round((per_step - k_step) * 1e3, 3)},
               "with_gather_MP_per_s": round(mp / per_step, 1),
               "host_gather": host}
        if shared:
            out["config"]["shared_gpu"] = "ranks share one GPU: functional run of the N>1 path, not a scaling number"
        emit(out)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def host_gather_leg(args, torch, dist, rank, world, dev, st, H, out_stride, y0, y1, local, decode, barrier_sync, images):
    """Every rank copies its slab device -> host into ITS ROWS of one image in host memory that all ranks' processes map (a file in
    /dev/shm, registered with the HIP runtime so that the copy is a DMA into it) - the collective-free form of the gather for a caller
    that wants the planes in host memory anyway (in-process: hm_decode_item_devices).  -> dict (rank 0), timings max over ranks"""
    import mmap
    path = f"/dev/shm/hm_grid_{os.environ.get('MASTER_PORT', '0')}_{os.getppid() if dist else os.getpid()}.bin"
    nbytes = H * out_stride
    cpu = "cpu" if (dist is None or args.dist_backend != "nccl") else dev

    def all_ok(flag):  # the ranks agree before anything that waits for the others: a rank that failed must not leave them at a barrier
        if dist is None:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=cpu)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def cleanup():
        try:
            if rank == 0 and os.path.exists(path):
                os.unlink(path)
        except OSError:
            pass

    # ---- set-up: the file, the mapping, the registration (everything that can fail on a box's limits) ----
    err, f, mm, host, registered = None, None, None, None, False
    try:
        if rank == 0:
            with open(path, "wb") as f0:
                f0.truncate(nbytes)
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if not all_ok(err is None):
        cleanup()
        return {"error": err or "another rank could not create the shared image"} if rank == 0 else None
    try:
        f = open(path, "r+b")
        mm = mmap.mmap(f.fileno(), nbytes)
        host = torch.frombuffer(mm, dtype=torch.uint8).view(H, out_stride)
        rt = torch.cuda.cudart()
        registered = int(rt.cudaHostRegister(host.data_ptr(), nbytes, 0)) == 0  # (not registered: the copies still work, through the runtime's staging)
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if not all_ok(err is None):
        host = None
        if mm is not None:
            mm.close()
        if f is not None:
            f.close()
        if dist:
            dist.barrier()
        cleanup()
        return {"error": err or "another rank could not map the shared image"} if rank == 0 else None

    copy_stream = torch.cuda.Stream(device=dev)
    copied = [torch.cuda.Event() for _ in local]  # buffer b's rows have left it
    rows = host[y0:y1]

    def d2h(b):
        copy_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(copy_stream):
            rows.copy_(local[b][:y1 - y0], non_blocking=True)
            copied[b].record(copy_stream)

    n = max(1, min(args.steps, 5))
    decode(0)
    d2h(0)
    barrier_sync()
    ok = True
    if rank == 0 and images is not None and not args.no_parity and world > 1:
        ok = bool(torch.equal(host, images[0].cpu()))  # (images[0] holds the RCCL / gloo gather of the same grid)
    t0 = time.perf_counter()
    for _ in range(n):
        decode(0)
        d2h(0)
        barrier_sync()
    serial = (time.perf_counter() - t0) / n
    barrier_sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        b = i % len(local)
        torch.cuda.current_stream().wait_event(copied[b])  # (the buffer's previous grid has been copied out)
        decode(b)
        d2h(b)
    barrier_sync()
    piped = (time.perf_counter() - t0) / args.steps
    if registered:
        rt.cudaHostUnregister(host.data_ptr())
    res = [serial, piped]
    if dist:
        tt = torch.tensor(res, dtype=torch.float64, device=cpu)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        res = [float(x) for x in tt.tolist()]
    del rows, host
    try:
        mm.close()
    except BufferError:  # (a view of the mapping is still alive somewhere: the mapping goes with the process)
        pass
    f.close()
    if dist:
        dist.barrier()
    cleanup()
    if rank == 0:
        mp = H * H / 1e6
        return {"what": "every rank D2H into its rows of ONE host image shared by the ranks (file in /dev/shm, hipHostRegister'ed): no collective; the image ends in HOST memory, "
                        "where heif_image_get_plane* hands it out (the gather above ends on rank 0's GPU)",
                "registered": registered, "equals_the_gathered_image": ok,
                "single_grid_latency_ms": round(res[0] * 1e3, 3), "pipelined_ms_per_grid": round(res[1] * 1e3, 3), "pipelined_MP_per_s": round(mp / res[1], 1)}
    return None


if __name__ == "__main__":
    main()
