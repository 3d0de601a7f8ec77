"""Image tiling across the GPUs of one node: one process per GPU, contiguous row strips, one
gather to rank 0 at the end of a render.

Pixels are independent and every sample's RNG stream is keyed by the GLOBAL pixel index, so the
assembled N-rank image is bit-identical to the 1-rank image.  The scene is replicated (tiny).
There is no exchange during rendering; the only collective is the final gather (RCCL when the
process group backend is "nccl", gloo in the CPU tests).
"""
from typing import List, Tuple

import numpy as np

TILE_ROWS = 8  # the render kernel's tile height: strips are cut on tile boundaries


def partition_rows(height: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous strips [begin, end) per rank, cut on 8-row tile boundaries, sizes differing by
    at most one tile; ranks beyond the number of tiles get an empty strip."""
    tiles = (height + TILE_ROWS - 1) // TILE_ROWS
    out = []
    for r in range(world):
        t0, t1 = r * tiles // world, (r + 1) * tiles // world
        out.append((min(t0 * TILE_ROWS, height), min(t1 * TILE_ROWS, height)))
    return out


def gather_strips(local, strips, rank, world, dst=0, group=None):
    """Gather per-rank strips (tensor [rows_r, W, C]) into the full image on `dst`.

    One collective: torch.distributed.gather of equal-size (max strip) buffers; the padding rows of
    shorter strips are dropped on the root.  Returns the assembled [H, W, C] tensor on dst, None
    elsewhere."""
    import torch
    import torch.distributed as dist
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()  # gloo rehearsal of the N-rank path: host tensors
    max_rows = max(e - b for b, e in strips)
    rows = strips[rank][1] - strips[rank][0]
    if rows == max_rows:
        send = local.contiguous()
    else:
        send = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[:rows] = local
    bufs = None
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: strips[r][1] - strips[r][0]] for r in range(world)], dim=0)


class StripRenderer:
    """Rank-local renderer of one row strip plus the end-of-step gather (bench.py's step)."""

    def __init__(self, data, rank=0, world=1, device=0, mode="repaired", max_bounces=-1,
                 seed=0x5EED, variant=0, want="f32", rows=None):
        from .renderer import Renderer
        self.data, self.rank, self.world = data, rank, world
        lo, hi = rows if rows else (0, data.height)
        self.strips = [(lo + b, lo + e) for b, e in partition_rows(hi - lo, world)]
        self.rows = self.strips[rank]
        self.want = want
        self.renderer = Renderer(data, mode=mode, max_bounces=max_bounces, seed=seed, device=device,
                                 variant=variant)
        self.image = None  # assembled frame on rank 0 after step()

    def step(self, stats=False, events=None):
        """Render this rank's strip into HBM, then gather the strips on rank 0."""
        import torch
        if events is not None:
            events[0].record()
        out, st = self.renderer.render_rows_device(self.rows[0], self.rows[1], want=(self.want,),
                                                   stats=stats)
        if events is not None:
            events[1].record()
        local = out[self.want]
        if self.world > 1:
            self.image = gather_strips(local, self.strips, self.rank, self.world)
        else:
            self.image = local
        return st
