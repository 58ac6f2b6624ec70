"""GPU: the N>1 path's only collective on the real backend.  torch.distributed's "nccl" backend IS RCCL on ROCm; a
one-rank group is enough to show that librccl loads on this image, that a communicator comes up on the device and that
shard.gather_slabs (one padded dist.gather of RGB row slabs) moves a device tensor through it unchanged.
Runs in a child process so that a backend failure cannot take the pytest process (and its HIP context) with it.
The two-rank twin of this test runs on CPU with gloo (tests/test_shard_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=dev)
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
t = torch.arange(4096, dtype=torch.int32, device=dev)
dist.all_reduce(t)
out["all_reduce"] = bool(torch.equal(t.cpu(), torch.arange(4096, dtype=torch.int32)))
# a slab as the grid mode of bench.py gathers it: rows of a 4032-pixel RGB24 canvas with libheif's stride
slab = torch.randint(0, 256, (512, 12096), dtype=torch.uint8, device=dev)
full = pkg.shard.gather_slabs(slab, [512], dst=0)
torch.cuda.synchronize()
out["gather"] = bool(full is not None and full.is_cuda and torch.equal(full, slab))
# (r06) the same through SlabGather: the root's slab is a view of the final image's rows (decoded in place), nothing to receive at one rank
gat = pkg.shard.SlabGather([(0, 2)], 256, 500, chunk_tile_rows=1, dst=0)
image = torch.zeros((500, 12096), dtype=torch.uint8, device=dev)
mine = gat.root_rows(image)
mine.copy_(slab[:500])
gat.wait(gat.post_recvs(image))
torch.cuda.synchronize()
out["slab_gather"] = bool(mine.shape[0] == 500 and mine.data_ptr() == image.data_ptr() and torch.equal(image, slab[:500]) and gat.rows == [[(0, 256), (256, 500)]])
# an empty slab (a rank beyond the grid's tile rows) next to nothing else: the padded gather still returns the rows
out["version"] = list(torch.cuda.nccl.version())
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def test_rccl_one_rank_gather_of_device_slabs():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "port": port}], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, r.stdout + r.stderr
    res = json.loads(line[-1][7:])
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["all_reduce"] and res["gather"] and res["slab_gather"], res


def test_grid_mode_on_one_rank_decodes_in_place_and_fills_the_shared_host_image():
    """bench.py --mode grid at N = 1 (r06): the 16384 x 16384 grid decoded straight into the rows of the final image (the root's path of
    shard.SlabGather), checked bit for bit against a separate one-rank decode, and the collective-free alternative - the rank's rows copied
    device -> host into ONE image in shared memory that is registered with the HIP runtime - filled and timed."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "grid", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "error" not in res, res
    assert res["n_gpus"] == 1 and "bit-exact" in res["config"]["self_check"], res["config"]
    assert res["config"]["tile_rows_per_rank"] == [32] and res["k_only"]["MP_per_s"] > 0
    h = res["host_gather"]
    assert "error" not in h and h["registered"] is True and h["equals_the_gathered_image"] is True and h["pipelined_ms_per_grid"] > 0, h


def _gpus_here():
    import torch
    return min(torch.cuda.device_count(), 8)  # (counting devices does not initialise the GPU in this process)


def _bench_ranks(world, extra, timeout):
    """bench.py as the driver launches it for N > 1: one process per GPU, torch.distributed.run, RCCL; -> its JSON line"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("world", ["two_gpus", "all_gpus"])
def test_one_grid_over_the_gpus_that_are_there(world):
    """SURVEY 8e on real devices: ONE 16384 x 16384 grid (BASELINE config 5), its 32 tile rows cut into a slab per rank, every rank
    decoding its slab on its own GPU, the RGB row slabs sent point-to-point over RCCL straight into their rows of rank 0's image
    (shard.SlabGather, r06: no padding, no concatenation; pipelined under the next grid's decode) - and rank 0
    comparing the gathered image bit for bit with its own one-rank decode of the whole grid (bench.py --mode grid's self-check;
    the reference's in-process tile fan-out context.cc:2361-2401 across processes).  Scales itself to the box: 2 ranks and
    min(device_count, 8) ranks; skipped on a one-GPU box, where tests/test_shard_gloo.py (2 gloo ranks) and the one-rank test
    above stand in."""
    n = _gpus_here()
    w = 2 if world == "two_gpus" else n
    if n < 2 or (world == "all_gpus" and n == 2):
        pytest.skip(f"{n} GPU(s) visible: nothing beyond the other cases to run")
    res = _bench_ranks(w, ["--mode", "grid", "--steps", "2", "--warmup", "1"], timeout=1800)
    assert "error" not in res, res
    assert res["n_gpus"] == w and res["world_size"] == w and res["gather"]["backend"] == "nccl"
    assert "bit-exact" in res["config"]["self_check"], res["config"]
    assert sum(res["config"]["tile_rows_per_rank"]) == 32
    assert res["host_gather"].get("equals_the_gathered_image") is True, res["host_gather"]  # (the collective-free alternative: one host image)


@pytest.mark.parametrize("world", ["two_gpus", "all_gpus"])
def test_image_batch_sharded_over_the_gpus_that_are_there(world):
    """BASELINE config 3's partitioning on real devices: the image batch block-sharded over the ranks (no collective on the
    data path), every rank's parity gate (a random image of ITS shard against the oracle and the real libde265) reduced over
    the ranks with RCCL - bench.py --gpus N exactly as the driver's scaling run starts it, with a small batch."""
    n = _gpus_here()
    w = 2 if world == "two_gpus" else n
    if n < 2 or (world == "all_gpus" and n == 2):
        pytest.skip(f"{n} GPU(s) visible: nothing beyond the other cases to run")
    res = _bench_ranks(w, ["--quick", "--images", "8", "--steps", "2", "--warmup", "1"], timeout=1800)
    assert "error" not in res, res
    assert res["n_gpus"] == w and res["dist"]["backend"] == "nccl" and res["dist"]["world_size"] == w
    assert "bit-exact" in res["config"]["parity"], res["config"]
    assert res["config"]["tiles_per_step_per_gpu"] == 8 * 48
