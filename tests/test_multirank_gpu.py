"""N-rank tiling on real renders (-m gpu): two and three ranks share cuda:0, render their row strips
with the HIP kernels and gather over gloo; the assembled frame must equal the single-rank frame bit
for bit (RNG streams are keyed by the global pixel index).  On the 8-GPU node the same code runs one
rank per GPU over RCCL (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scene, dims, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raytracingmin_amd as rtm
        from raytracingmin_amd.distributed import StripRenderer, gather_strips, partition_rows
        data = rtm.LoadData(scene).data
        data.width, data.height, data.samples, data.superSamples = dims
        strips = partition_rows(data.height, world)
        r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=0x5EED, device=0)
        out, stats = r.render_rows_device(strips[rank][0], strips[rank][1], want=("f64",))
        img = gather_strips(out["f64"].cpu(), strips, rank, world)  # gloo gathers host tensors
        # bench.py's step: interleaved 8-row bands (rtm_options.band_count/band_index) + one gather
        sr = StripRenderer(data, rank=rank, world=world, device=0, mode="repaired", max_bounces=8, seed=0x5EED,
                           want="f64", layout="bands")
        sr.step()
        if rank == 0:
            q.put(img.numpy())
            q.put(sr.image.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims", [(2, (96, 72, 4, 2)), (3, (80, 44, 3, 2))])
def test_strips_from_n_ranks_equal_single_rank_frame(world, dims):
    import torch.multiprocessing as mp
    import _oracle
    import raytracingmin_amd as rtm
    scene = _oracle.scene_path("cornellBoxSetting.json")
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scene, dims, q)) for r in range(world)]
    for p in procs:
        p.start()
    img = q.get()
    img_bands = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    data = rtm.LoadData(scene).data
    data.width, data.height, data.samples, data.superSamples = dims
    full, _ = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=0x5EED).render_rows(want=("f64",))
    assert np.array_equal(img.view(np.uint64), full["f64"].view(np.uint64))
    assert np.array_equal(img_bands.view(np.uint64), full["f64"].view(np.uint64))
