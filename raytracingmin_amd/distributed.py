"""Image tiling across the GPUs of one node: one process per GPU, interleaved 8-row bands (band b of
the frame belongs to rank b mod N, so every rank gets the same mix of cheap and costly rows) or
contiguous row strips, and one gather to rank 0 at the end of a render.

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


def band_row_index(row_begin: int, row_end: int, world: int, rank: int) -> np.ndarray:
    """Image rows, in output order, of the call (band_count, band_index) = (world, rank) over
    [row_begin, row_end): the mirror of rtm_output_rows / band_row in the kernels."""
    span = row_end - row_begin
    bands = (span + TILE_ROWS - 1) // TILE_ROWS
    rows = [np.arange(b * TILE_ROWS, min((b + 1) * TILE_ROWS, span)) for b in range(rank, bands, max(world, 1))]
    return row_begin + (np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64))


_index_cache = {}


def _band_index_tensors(span, world, device):
    key = (span, world, str(device))
    if key not in _index_cache:
        import torch
        _index_cache[key] = [torch.as_tensor(band_row_index(0, span, world, r), device=device) for r in range(world)]
    return _index_cache[key]


def gather_bands(local, row_begin, row_end, rank, world, dst=0, group=None):
    """Gather per-rank band stacks (tensor [rows_r, W, C]) and put the bands back in image order on
    `dst`.  One collective (equal-size buffers, padded to the largest stack); the de-interleave is one
    index_copy per rank on the root.  Returns [row_end - row_begin, W, C] on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()
    index = _band_index_tensors(row_end - row_begin, world, local.device)
    max_rows = max(len(i) for i in index)
    rows = len(index[rank])
    assert local.shape[0] == rows, (local.shape, rows)
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
    full = torch.empty((row_end - row_begin,) + tuple(local.shape[1:]), dtype=local.dtype, device=send.device)
    for r in range(world):
        if len(index[r]):
            full.index_copy_(0, index[r], bufs[r][: len(index[r])])
    return full


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


class StageWatchdog:
    """Names and bounds the stages of a rank's start-up and collectives (the Python mirror of csrc/rtm_node.cpp's
    watchdog).  enter(name) prints `[rank r] stage: name` on stderr (and to `stage_file`, if given) and re-arms the
    limit; a stage that is still running after `limit_s` seconds ends the PROCESS with exit code 3 and a line naming
    it — a fresh non-zero exit (os._exit), never a re-exec — so a launcher sees a failed rank with a located cause
    instead of a job that dies silently at its caller's limit.  done() disarms it."""

    def __init__(self, limit_s=120.0, rank=0, stage_file=None, quiet=False):
        import threading
        self.limit_s, self.rank, self.stage_file, self.quiet = float(limit_s), rank, stage_file, quiet
        self._name, self._since, self._stop = None, 0.0, False
        self._lock = threading.Lock()
        self._thread = threading.Thread(target=self._watch, name="rtm-stage-watchdog", daemon=True)
        self._thread.start()

    def _say(self, text):
        import os
        os.write(2, (f"[rank {self.rank}] {text}\n").encode())

    def enter(self, name, limit_s=None):
        import time
        with self._lock:
            self._name, self._since = name, time.monotonic()
            self._limit = self.limit_s if limit_s is None else float(limit_s)
        if not self.quiet:
            self._say(f"stage: {name}")
        if self.stage_file:
            with open(self.stage_file, "w") as f:
                f.write(name)

    def done(self):
        with self._lock:
            self._name = None

    def _watch(self):
        import os
        import time
        while True:
            time.sleep(0.5)
            with self._lock:
                name, since, limit = self._name, self._since, getattr(self, "_limit", self.limit_s)
            if name is not None and limit > 0 and time.monotonic() - since > limit:
                self._say(f"WATCHDOG: stage '{name}' has not finished after {limit:.0f} s; exiting with code 3")
                if self.stage_file:
                    try:
                        with open(self.stage_file, "w") as f:
                            f.write(f"WATCHDOG: {name}")
                    except OSError:
                        pass
                os._exit(3)


class StripRenderer:
    """Rank-local renderer of this rank's share of the rows plus the end-of-step gather (bench.py's
    step).  layout "bands": interleaved 8-row bands (default); "strips": contiguous strips."""

    def __init__(self, data, rank=0, world=1, device=0, mode="repaired", max_bounces=-1,
                 seed=0x5EED, variant=0, want="f32", rows=None, layout="bands", host_trig=True,
                 force_collective=False, renderer=None):
        if layout not in ("bands", "strips"):
            raise ValueError(f"unknown layout {layout!r}")
        self.data, self.rank, self.world, self.layout = data, rank, world, layout
        lo, hi = rows if rows else (0, data.height)
        self.range = (lo, hi)
        self.strips = [(lo + b, lo + e) for b, e in partition_rows(hi - lo, world)]
        self.rows = self.strips[rank]
        self.want = want
        # world == 1 normally skips the gather; force_collective sends the frame through the process
        # group anyway (the one-rank RCCL rehearsal of the N-rank step)
        self.force_collective = bool(force_collective)
        if renderer is None:  # (`renderer`: a stand-in with render_rows_device, for the CPU rehearsal of the N-rank step)
            from .renderer import Renderer
            renderer = Renderer(data, mode=mode, max_bounces=max_bounces, seed=seed, device=device,
                                variant=variant, host_trig=host_trig)
        self.renderer = renderer
        self.image = None  # assembled frame on rank 0 after step()
        self.local = None  # this rank's rows of the last step (band stack or strip)

    def step(self, stats=False, events=None):
        """Render this rank's strip into HBM, then gather the strips on rank 0."""
        import torch
        if events is not None:
            events[0].record()
        if self.layout == "bands" and self.world > 1:
            out, st = self.renderer.render_rows_device(self.range[0], self.range[1], want=(self.want,),
                                                       stats=stats, band=(self.world, self.rank))
        else:
            out, st = self.renderer.render_rows_device(self.rows[0], self.rows[1], want=(self.want,),
                                                       stats=stats)
        if events is not None:
            events[1].record()
        self.local = out[self.want]
        self.gather()
        return st

    def gather(self):
        """The end-of-step collective alone, on the rows of the last step: rank 0 gets the assembled frame."""
        local = self.local
        if self.world == 1 and not self.force_collective:
            self.image = local
        elif self.layout == "bands":
            self.image = gather_bands(local, self.range[0], self.range[1], self.rank, self.world)
        else:
            self.image = gather_strips(local, self.strips, self.rank, self.world)
        return self.image

    def single_gpu_frame(self):
        """The whole row range rendered by THIS rank alone, in one launch — what the assembled frame must equal bit
        for bit (the RNG is keyed by the global pixel index: tiling must not change any sample)."""
        out, _ = self.renderer.render_rows_device(self.range[0], self.range[1], want=(self.want,), stats=False)
        return out[self.want]


def frames_equal_bitwise(a, b):
    """Two frames (torch tensors of one dtype and shape, any device) hold the same BITS (NaNs included)."""
    import torch
    if a is None or b is None or a.shape != b.shape or a.dtype != b.dtype:
        return False
    view = {torch.float32: torch.int32, torch.float64: torch.int64}.get(a.dtype)
    a, b = a.detach().cpu(), b.detach().cpu()
    return bool(torch.equal(a.view(view), b.view(view))) if view else bool(torch.equal(a, b))


def multi_gpu_evidence(sr, stats, kernel_ms, steps, barrier, use_group=True):
    """What lets an N-rank bench line prove itself (all of it OUTSIDE the timed region; every rank calls this):
      frame_matches_single_gpu  rank 0 renders the whole row range alone, in one launch on its own GPU, and compares the
                                frame the ranks assembled with it bit for bit;
      per_rank                  every rank's kernel time, rows, samples, casts, bounces and draws of the instrumented
                                step (load balance; their sums are the frame's counters);
      gather_ms                 the end-of-step collective alone (the rows of the last step gathered `steps` more times,
                                barrier + synchronise on both sides, slowest rank).
    Returns the dict on rank 0 and None elsewhere."""
    import time
    import torch
    import torch.distributed as dist
    rows = int(sr.local.shape[0]) if sr.local is not None else 0
    mine = torch.tensor([float(kernel_ms), float(rows)] + [float(stats[k]) for k in ("samples", "casts", "bounces", "draws")],
                        dtype=torch.float64)
    world = sr.world
    if use_group:
        on = sr.local.device if (sr.local is not None and dist.get_backend() == "nccl") else "cpu"
        parts = [torch.zeros_like(mine, device=on) for _ in range(world)]
        dist.all_gather(parts, mine.to(on))
        parts = [p.cpu() for p in parts]
    else:
        parts = [mine]
    per_rank = [{"rank": r, "kernel_ms": float(p[0]), "rows": int(p[1]), "samples": int(p[2]), "casts": int(p[3]),
                 "bounces": int(p[4]), "draws": int(p[5])} for r, p in enumerate(parts)]
    assembled = sr.image  # rank 0: the frame of the last step
    barrier()
    t0 = time.perf_counter()
    for _ in range(max(1, steps)):
        sr.gather()
    barrier()
    gather_ms = (time.perf_counter() - t0) / max(1, steps) * 1e3
    t = torch.tensor([gather_ms], dtype=torch.float64)
    if use_group:
        t = t.to(on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gather_ms = float(t[0])
    match = None
    if sr.rank == 0:
        match = frames_equal_bitwise(assembled, sr.single_gpu_frame())
    barrier()
    if sr.rank != 0:
        return None
    total = {k: sum(p[k] for p in per_rank) for k in ("rows", "samples", "casts", "bounces", "draws")}
    slowest = max(p["kernel_ms"] for p in per_rank)
    mean = sum(p["kernel_ms"] for p in per_rank) / len(per_rank)
    return {"frame_matches_single_gpu": match, "per_rank": per_rank, "totals": total, "gather_ms": gather_ms,
            "kernel_ms_slowest_over_mean": slowest / mean if mean > 0 else None,
            "how": "rank 0 re-rendered the whole row range alone (one launch) and compared the gathered frame with it bit "
                   "for bit; per_rank from the instrumented step; gather_ms = the collective alone, slowest rank; all "
                   "outside the timed region"}
