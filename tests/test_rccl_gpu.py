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
    assert res["all_reduce"] and res["gather"], res
