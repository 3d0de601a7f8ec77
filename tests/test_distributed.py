"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercising the row partition and
the single end-of-render gather that bench.py --gpus N uses (RCCL on the GPU node)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracingmin_amd.distributed import band_row_index, gather_bands, gather_strips, partition_rows


@pytest.mark.parametrize("height,world", [(1080, 1), (1080, 2), (1080, 4), (1080, 8), (2160, 8),
                                          (504, 8), (45, 4), (7, 4), (16, 3)])
def test_partition_rows_covers_image_on_tile_boundaries(height, world):
    strips = partition_rows(height, world)
    assert len(strips) == world and strips[0][0] == 0 and strips[-1][1] == height
    for (b0, e0), (b1, e1) in zip(strips, strips[1:]):
        assert e0 == b1 and b0 <= e0
    for b, e in strips:
        assert b % 8 == 0 and (e % 8 == 0 or e == height)
    sizes = [(e - b + 7) // 8 for b, e in strips]
    assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("lo,hi,world", [(0, 1080, 8), (0, 1080, 1), (0, 45, 4), (16, 61, 3), (0, 7, 4), (8, 8, 2),
                                         (0, 2160, 8), (0, 24, 5)])
def test_band_rows_partition_the_range_and_match_the_library(lo, hi, world):
    """Interleaved bands: every row of [lo, hi) belongs to exactly one rank, bands are dealt round
    robin, and the row count equals rtm_output_rows (the C side of the same rule)."""
    import ctypes as C
    from raytracingmin_amd import _lib
    seen = []
    for rank in range(world):
        idx = band_row_index(lo, hi, world, rank)
        o = _lib.rtm_options()
        o.row_begin, o.row_end, o.band_count, o.band_index = lo, hi, world, rank
        assert _lib.lib().rtm_output_rows(C.byref(o)) == len(idx)
        assert all(((r - lo) // 8) % world == rank for r in idx)
        assert list(idx) == sorted(idx)
        seen.extend(idx.tolist())
    assert sorted(seen) == list(range(lo, hi))
    sizes = [len(band_row_index(lo, hi, world, r)) for r in range(world)]
    assert max(sizes) - min(sizes) <= 8


def _collect(procs, q, n_items, timeout=120):
    """n_items results from the workers' queue; fails (instead of hanging) when a worker dies first."""
    import queue as _q
    import time
    items, deadline = [], time.time() + timeout
    try:
        while len(items) < n_items:
            try:
                items.append(q.get(timeout=1.0))
            except _q.Empty:
                dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
                assert not dead, f"a worker exited with {dead} before delivering its result"
                assert time.time() < deadline, "workers timed out"
        for p in procs:
            p.join(60)
            assert p.exitcode == 0, p.exitcode
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    return items


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, height, width, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        strips = partition_rows(height, world)
        b, e = strips[rank]
        # stand-in for the rendered strip: a value that encodes (global row, column, channel)
        rows = torch.arange(b, e, dtype=torch.float32).view(-1, 1, 1)
        cols = torch.arange(width, dtype=torch.float32).view(1, -1, 1)
        ch = torch.arange(3, dtype=torch.float32).view(1, 1, 3)
        local = rows * 10000 + cols * 10 + ch
        img = gather_strips(local, strips, rank, world)
        # the same frame dealt out in interleaved bands
        mine = torch.as_tensor(band_row_index(0, height, world, rank), dtype=torch.float32).view(-1, 1, 1)
        img2 = gather_bands(mine * 10000 + cols * 10 + ch, 0, height, rank, world)
        if rank == 0:
            q.put(img.numpy())
            q.put(img2.numpy())
        else:
            assert img is None and img2 is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 40), (2, 45), (3, 20)])
def test_gather_strips_and_bands_gloo(world, height):
    width = 13
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, width, q)) for r in range(world)]
    for p in procs:
        p.start()
    img, img_bands = _collect(procs, q, 2)
    r, c, ch = np.meshgrid(np.arange(height), np.arange(width), np.arange(3), indexing="ij")
    assert np.array_equal(img, (r * 10000 + c * 10 + ch).astype(np.float32))
    assert np.array_equal(img_bands, (r * 10000 + c * 10 + ch).astype(np.float32))


class _FakeRenderer:
    """Stand-in for raytracingmin_amd.Renderer on a machine without a GPU: a "pixel" is a function of its GLOBAL row and
    column (like the real one, whose RNG is keyed by the global pixel index), rendered into CPU tensors.  `bad_rank`
    corrupts one pixel of that rank's band part — the proof must notice."""

    def __init__(self, width, rank, bad_rank=None):
        self.width, self.rank, self.bad_rank = width, rank, bad_rank

    def render_rows_device(self, row_begin=0, row_end=None, want=("f32",), stats=True, stream=None, band=None):
        rows = band_row_index(row_begin, row_end, band[0], band[1]) if band else np.arange(row_begin, row_end)
        r = torch.as_tensor(rows, dtype=torch.float32).view(-1, 1, 1)
        cols = torch.arange(self.width, dtype=torch.float32).view(1, -1, 1)
        ch = torch.arange(3, dtype=torch.float32).view(1, 1, 3)
        img = torch.sin(r * 12.9898 + cols * 78.233 + ch)  # any deterministic function of the global pixel
        if band and self.bad_rank == self.rank and len(rows):
            img[0, 0, 0] += 1e-6
        n = len(rows) * self.width
        st = {"samples": n * 4, "casts": n * 9 + self.rank, "bounces": n * 5, "draws": n * 19, "kernel_ms": 1.0 + self.rank,
              "variant": 2, "split": 1}
        return {want[0]: img}, (st if stats else None)

    def stream_status(self, stream=None):
        pass


def _evidence_worker(rank, world, port, height, width, bad_rank, q):
    from raytracingmin_amd.distributed import StripRenderer, multi_gpu_evidence
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sr = StripRenderer(None, rank=rank, world=world, rows=(0, height), renderer=_FakeRenderer(width, rank, bad_rank))
        stats = sr.step(stats=True)
        ev = multi_gpu_evidence(sr, stats, stats["kernel_ms"], steps=2, barrier=dist.barrier)
        if rank == 0:
            q.put(ev)
        else:
            assert ev is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bad_rank", [(2, None), (2, 1), (3, None)])
def test_multi_gpu_evidence_keys_gloo(world, bad_rank):
    """What bench.py --gpus N puts into its line so that an N-rank run proves itself: the assembled frame
    compared bit for bit with rank 0's own single-launch render, every rank's kernel time / rows / counters, the gather
    alone timed.  World 2 and 3 over gloo with a stand-in renderer; a corrupted part must flip the flag."""
    height, width = 45, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_evidence_worker, args=(r, world, port, height, width, bad_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    (ev,) = _collect(procs, q, 1)
    assert ev["frame_matches_single_gpu"] is (bad_rank is None)
    assert [p["rank"] for p in ev["per_rank"]] == list(range(world))
    assert sum(p["rows"] for p in ev["per_rank"]) == height == ev["totals"]["rows"]
    for p in ev["per_rank"]:
        assert p["rows"] == len(band_row_index(0, height, world, p["rank"]))
        assert p["kernel_ms"] == 1.0 + p["rank"] and p["samples"] == p["rows"] * width * 4
        assert p["casts"] == p["rows"] * width * 9 + p["rank"]
    assert ev["gather_ms"] > 0.0 and ev["kernel_ms_slowest_over_mean"] >= 1.0
