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
