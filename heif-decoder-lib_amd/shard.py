"""Multi-GPU sharding of the tile batch (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in CPU tests).

HEIF grid tiles are independent coded pictures (the reference already decodes them concurrently with
no ordering, libheif/context.cc:2361-2401), so the decode path needs NO collective:
  * a batch of images is block-sharded by image index  -> zero exchange (bench default, "weak");
  * one big grid is sharded by contiguous tile ROWS     -> each rank owns a contiguous slab of canvas
    rows; colour conversion is per-pixel nearest-neighbour chroma, so slabs that start on even rows
    need no halo.  The only exchange is the final gather of the RGB slabs into one pixel image on
    the root - the single RCCL collective of the design.
"""
from typing import List, Tuple


def image_shard(n_images: int, rank: int, world: int) -> range:
    """contiguous block of image indices owned by `rank` (sizes differ by at most one)"""
    base, extra = divmod(n_images, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def row_slabs(grid_rows: int, world: int) -> List[Tuple[int, int]]:
    """(first_tile_row, n_tile_rows) per rank: contiguous and balanced; ranks beyond the row count get (x, 0)"""
    out = []
    for r in range(world):
        rng = image_shard(grid_rows, r, world)
        out.append((rng.start, len(rng)))
    return out


def slab_pixel_rows(first_tile_row: int, n_tile_rows: int, tile_h: int, out_h: int) -> Tuple[int, int]:
    """canvas rows [y0, y1) covered by a slab of tile rows, cropped at the output height"""
    y0 = min(first_tile_row * tile_h, out_h)
    y1 = min((first_tile_row + n_tile_rows) * tile_h, out_h)
    return y0, y1


def gather_slabs(local, heights: List[int], dst: int = 0, group=None):
    """Gather per-rank row slabs (2-D uint8 tensors [rows_r, stride], same stride everywhere) into one
    [sum(rows), stride] tensor on `dst` (None elsewhere).  Slabs are padded to the tallest one so a
    single gather collective moves everything (one large transfer per xGMI link instead of many)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(heights) == world and local.shape[0] == heights[rank]
    hmax = max(heights)
    stride = local.shape[1]
    if local.shape[0] < hmax:
        pad = torch.zeros((hmax - local.shape[0], stride), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    local = local.contiguous()
    if rank == dst:
        bufs = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, bufs, dst=dst, group=group)
        return torch.cat([b[:h] for b, h in zip(bufs, heights)], 0)
    dist.gather(local, None, dst=dst, group=group)
    return None
