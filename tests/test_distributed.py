"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercising the row partition and
the single end-of-render gather that bench.py --gpus N uses (RCCL on the GPU node)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracingmin_amd.distributed import gather_strips, partition_rows


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
        if rank == 0:
            q.put(img.numpy())
        else:
            assert img is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 40), (2, 45), (3, 20)])
def test_gather_strips_gloo(world, height):
    width = 13
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, width, q)) for r in range(world)]
    for p in procs:
        p.start()
    img = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    r, c, ch = np.meshgrid(np.arange(height), np.arange(width), np.arange(3), indexing="ij")
    assert np.array_equal(img, (r * 10000 + c * 10 + ch).astype(np.float32))
