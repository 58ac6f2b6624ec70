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


def slab_chunks(first_tile_row: int, n_tile_rows: int, chunk_tile_rows: int) -> List[Tuple[int, int]]:
    """a slab of tile rows cut into chunks of at most `chunk_tile_rows` (the last one may be shorter; 0 or less: one chunk):
    [(first_tile_row, n_tile_rows), ...] - the unit a rank hands to the gather as soon as its rows are decoded"""
    if n_tile_rows <= 0:
        return []
    if chunk_tile_rows <= 0 or chunk_tile_rows >= n_tile_rows:
        return [(first_tile_row, n_tile_rows)]
    return [(r, min(chunk_tile_rows, first_tile_row + n_tile_rows - r)) for r in range(first_tile_row, first_tile_row + n_tile_rows, chunk_tile_rows)]


class SlabGather:
    """The grid gather without padding and without a copy on the root (r06): every rank sends the rows of its slab in chunks of tile
    rows (point-to-point, in order per peer), the root receives every chunk STRAIGHT INTO ITS ROWS of the final image and decodes its
    own slab in place (`root_rows`).  `gather_slabs` above pads every slab to the tallest one and concatenates on the root - one more
    pass over the whole image on the root GPU, and 1/world of the bytes moved for nothing when the rows do not divide.

    stage_through_host: the transport cannot take device tensors (gloo): the chunks travel as CPU tensors (tests, `--dist-backend gloo`).
    Usage per image:  root: recvs = g.post_recvs(full) ... g.wait(recvs);   every other rank: g.send(local, stream_event=None)"""

    def __init__(self, slabs: List[Tuple[int, int]], tile_h: int, out_h: int, chunk_tile_rows: int = 0, dst: int = 0, group=None,
                 stage_through_host: bool = False):
        import torch.distributed as dist
        self.dist = dist
        self.group, self.dst = group, dst
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        assert len(slabs) == self.world
        self.stage = stage_through_host
        # per rank: the pixel rows [y0, y1) of each of its chunks, in sending order; empty chunks (cropped away) are skipped on both sides
        self.rows = []
        for first, n in slabs:
            ch = [slab_pixel_rows(a, b, tile_h, out_h) for a, b in slab_chunks(first, n, chunk_tile_rows)]
            self.rows.append([(y0, y1) for y0, y1 in ch if y1 > y0])
        self.slab_y0 = [slab_pixel_rows(a, b, tile_h, out_h)[0] for a, b in slabs]

    def root_rows(self, full):
        """root only: the rows of `full` that hold the root's own slab (decode into them: no copy)"""
        r = self.rows[self.dst]
        return full[r[0][0]:r[-1][1]] if r else full[0:0]

    def post_recvs(self, full):
        """root only: one receive per (rank, chunk) into the chunk's rows of `full` ([out_h, stride], contiguous rows)"""
        import torch
        assert self.rank == self.dst
        works = []
        peers = [r for r in range(self.world) if r != self.dst]
        if self.stage:
            for r in peers:
                for y0, y1 in self.rows[r]:
                    view = full[y0:y1]
                    buf = torch.empty(view.shape, dtype=view.dtype, device="cpu")
                    works.append((self.dist.irecv(buf, src=r, group=self.group), buf, view))
            return works
        # device tensors (RCCL): receives posted one by one run one after the other on the root - one xGMI link busy at a time.  Chunk k
        # of EVERY peer goes into one group (ncclGroupStart / End behind batch_isend_irecv): the root's seven links fill side by side;
        # per peer the groups come in the order of its sends.
        k = 0
        while True:
            ops = [self.dist.P2POp(self.dist.irecv, full[self.rows[r][k][0]:self.rows[r][k][1]], r, self.group) for r in peers if k < len(self.rows[r])]
            if not ops:
                break
            for w in self.dist.batch_isend_irecv(ops):
                works.append((w, None, None))
            k += 1
        return works

    def wait(self, works):
        for w, buf, view in works:
            w.wait()
            if buf is not None:
                view.copy_(buf)

    def send(self, local, chunk_ready=None):
        """every other rank: `local` = its slab ([rows of the slab, stride], row 0 = the slab's first pixel row).  chunk_ready(i): called
        before chunk i is handed over (e.g. waits for the event behind that chunk's decode on the sending stream); returns the works"""
        assert self.rank != self.dst
        works = []
        base = self.slab_y0[self.rank]
        for i, (y0, y1) in enumerate(self.rows[self.rank]):
            if chunk_ready is not None:
                chunk_ready(i)
            chunk = local[y0 - base:y1 - base]
            if self.stage:
                chunk = chunk.cpu()
            works.append(self.dist.isend(chunk.contiguous(), dst=self.dst, group=self.group))
        return works
