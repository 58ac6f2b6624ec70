"""CPU, world_size 2, gloo: the N>1 path - sharding of the tile batch and the final RGB gather.
The per-rank decode is stood in for by the CPU oracle (there is no GPU here); what is under test is
the product's sharding / gather code (heif-decoder-lib_amd/shard.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_shard_arithmetic(pkg):
    sh = pkg.shard
    for n in (0, 1, 5, 8, 48, 1024):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(sh.image_shard(n, r, world))
            assert seen == list(range(n))
            sizes = [len(sh.image_shard(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert sh.row_slabs(32, 8) == [(4 * r, 4) for r in range(8)]
    assert sh.row_slabs(6, 4) == [(0, 2), (2, 2), (4, 1), (5, 1)]
    assert sh.slab_pixel_rows(4, 2, 512, 3024) == (2048, 3024)
    assert sh.slab_pixel_rows(6, 1, 512, 3024) == (3024, 3024)


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    import corpus
    import pipeline
    pkg = g.load_package()
    hm = pkg.lib()
    sh = pkg.shard
    # a 3x2 grid of 64x64 tiles, output 120x170 (cropped); tile rows sharded over the ranks
    rows, cols, T, W, H = 3, 2, 64, 120, 170
    tiles = [__import__("synthutil").picture(900 + i, width=T, height=T, vui=(i % 2), full_range=1, matrix=6) for i in range(rows * cols)]
    slabs = sh.row_slabs(rows, world)
    r0, nr = slabs[rank]
    y0, y1 = sh.slab_pixel_rows(r0, nr, T, H)
    stride = pipeline.orc.plane_stride(W, 3)
    if nr:
        mine = tiles[r0 * cols:(r0 + nr) * cols]
        out, os_, _ = pipeline.cpu_decode(hm, mine, T, T, W, y1 - y0, cols, True, 10)
        local = torch.from_numpy(out[:y1 - y0].copy())
        assert os_ == stride
    else:
        local = torch.zeros((0, stride), dtype=torch.uint8)
    heights = [sh.slab_pixel_rows(a, b, T, H)[1] - sh.slab_pixel_rows(a, b, T, H)[0] for a, b in slabs]
    full = sh.gather_slabs(local, heights, dst=0)
    if rank == 0:
        np.save(os.path.join(tmp, "gathered.npy"), full.numpy())
        ref, _, _ = pipeline.cpu_decode(hm, tiles, T, T, W, H, cols, True, 10)
        np.save(os.path.join(tmp, "single.npy"), ref[:H])
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_grid_gather_equals_single_process(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "gathered.npy")
    b = np.load(tmp_path / "single.npy")
    assert a.shape == b.shape
    np.testing.assert_array_equal(a[:, :120 * 3], b[:, :120 * 3])


def test_slab_chunks(pkg):
    sh = pkg.shard
    assert sh.slab_chunks(4, 4, 1) == [(4, 1), (5, 1), (6, 1), (7, 1)]
    assert sh.slab_chunks(5, 3, 2) == [(5, 2), (7, 1)]          # uneven: the last chunk is shorter
    assert sh.slab_chunks(2, 3, 0) == [(2, 3)] and sh.slab_chunks(2, 3, 8) == [(2, 3)]
    assert sh.slab_chunks(7, 0, 1) == []
    for rows, world, chunk in ((32, 8, 1), (5, 2, 2), (6, 4, 4), (3, 5, 1)):
        seen = []
        for a, b in sh.row_slabs(rows, world):
            for c0, cn in sh.slab_chunks(a, b, chunk):
                assert 0 < cn <= max(chunk, 1) or chunk <= 0
                seen += list(range(c0, c0 + cn))
        assert seen == list(range(rows))


def _worker_chunked(rank, world, port, tmp, rows, chunk):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    import pipeline
    pkg = g.load_package()
    hm = pkg.lib()
    sh = pkg.shard
    # a rows x 2 grid of 64x64 tiles whose last tile row is cut by the output height; tile rows sharded over the ranks, every slab
    # sent in chunks of `chunk` tile rows straight into its rows of the root's image (SlabGather: no padding, no concatenation)
    cols, T, W = 2, 64, 120
    H = rows * T - 22
    tiles = [__import__("synthutil").picture(1900 + i, width=T, height=T, vui=(i % 2), full_range=1, matrix=6) for i in range(rows * cols)]
    slabs = sh.row_slabs(rows, world)
    r0, nr = slabs[rank]
    y0, y1 = sh.slab_pixel_rows(r0, nr, T, H)
    stride = pipeline.orc.plane_stride(W, 3)
    gat = sh.SlabGather(slabs, T, H, chunk_tile_rows=chunk, dst=0, stage_through_host=True)
    local = None
    if nr and y1 > y0:
        out, os_, _ = pipeline.cpu_decode(hm, tiles[r0 * cols:(r0 + nr) * cols], T, T, W, y1 - y0, cols, True, 10)
        assert os_ == stride
        local = torch.from_numpy(out[:y1 - y0].copy())
    if rank == 0:
        full = torch.full((H, stride), 0xEE, dtype=torch.uint8)
        recvs = gat.post_recvs(full)
        mine = gat.root_rows(full)
        assert mine.shape[0] == y1 - y0 and (local is None or mine.data_ptr() == full[y0:].data_ptr())
        if local is not None:
            mine.copy_(local)  # (the root's decode writes its rows of the final image in place)
        gat.wait(recvs)
        np.save(os.path.join(tmp, "gathered.npy"), full.numpy())
        ref, _, _ = pipeline.cpu_decode(hm, tiles, T, T, W, H, cols, True, 10)
        np.save(os.path.join(tmp, "single.npy"), ref[:H])
        np.save(os.path.join(tmp, "chunks.npy"), np.array([len(r) for r in gat.rows]))
    else:
        seen = []
        works = gat.send(local if local is not None else torch.zeros((0, stride), dtype=torch.uint8), chunk_ready=seen.append)
        for w in works:
            w.wait()
        assert seen == list(range(len(gat.rows[rank])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,chunk", [(2, 5, 2), (2, 3, 0), (3, 2, 1), (3, 7, 2)])
def test_chunked_gather_straight_into_the_final_rows(tmp_path, world, rows, chunk):
    """r06: slabs in chunks of tile rows, uneven chunking (5 rows over 2 ranks in chunks of 2: 2 + 1 and 2), a rank without rows
    (2 rows over 3 ranks), a cropped last tile row; the gathered image equals the single-process decode bit for bit."""
    port = 31500 + (os.getpid() % 2000) + 7 * world + rows
    mp.spawn(_worker_chunked, args=(world, port, str(tmp_path), rows, chunk), nprocs=world, join=True)
    a = np.load(tmp_path / "gathered.npy")
    b = np.load(tmp_path / "single.npy")
    assert a.shape == b.shape
    np.testing.assert_array_equal(a[:, :120 * 3], b[:, :120 * 3])
    n_chunks = np.load(tmp_path / "chunks.npy")
    if (world, rows, chunk) == (2, 5, 2):
        assert list(n_chunks) == [2, 1]
    if (world, rows, chunk) == (3, 2, 1):
        assert list(n_chunks) == [1, 1, 0]
