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


class WorkersTimedOut(Exception):
    pass


def _collect(procs, q, n_items, timeout=240):
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
                if time.time() >= deadline:
                    raise WorkersTimedOut()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0, p.exitcode
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    return items


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
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scene, dims, q)) for r in range(world)]
    for p in procs:
        p.start()
    img, img_bands = _collect(procs, q, 2)
    data = rtm.LoadData(scene).data
    data.width, data.height, data.samples, data.superSamples = dims
    full, _ = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=0x5EED).render_rows(want=("f64",))
    assert np.array_equal(img.view(np.uint64), full["f64"].view(np.uint64))
    assert np.array_equal(img_bands.view(np.uint64), full["f64"].view(np.uint64))


def _rccl_worker(port, scene, dims, q, stage_path):
    """ONE rank, backend nccl (= RCCL): the N-rank step with the collective forced — communicator init,
    dist.gather on device tensors, the de-interleave on the root.  Every stage is written to `stage_path` before it
    starts and bounded by raytracingmin_amd.distributed.StageWatchdog (exit code 3 with the stage's name)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    from raytracingmin_amd.distributed import StageWatchdog
    dog = StageWatchdog(limit_s=60, stage_file=stage_path)
    dog.enter("import torch")
    import torch
    import torch.distributed as dist
    dog.enter("torch.cuda.set_device(0)")
    torch.cuda.set_device(0)
    dog.enter("init_process_group(nccl, world 1, lo)")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dog.enter("renders + dist.gather over RCCL")
    try:
        import raytracingmin_amd as rtm
        from raytracingmin_amd.distributed import StripRenderer, gather_bands, gather_strips
        assert dist.get_backend() == "nccl"
        data = rtm.LoadData(scene).data
        data.width, data.height, data.samples, data.superSamples = dims
        out = {}
        for layout in ("bands", "strips"):
            sr = StripRenderer(data, rank=0, world=1, device=0, mode="repaired", max_bounces=8, seed=0x5EED,
                               want="f32", layout=layout, force_collective=True)
            sr.step()
            assert sr.image.is_cuda
            out[layout] = sr.image.cpu().numpy()
        # the collectives on their own, on device tensors that are not a render's output
        t = torch.arange(45 * 7 * 3, dtype=torch.float32, device="cuda").reshape(45, 7, 3)
        g = gather_bands(t, 0, 45, 0, 1)
        g2 = gather_strips(t, [(0, 45)], 0, 1)
        torch.cuda.synchronize()
        q.put((out["bands"], out["strips"], bool(torch.equal(g, t)) and bool(torch.equal(g2, t))))
        dog.enter("final barrier")
        dist.barrier()
    finally:
        dog.enter("destroy_process_group")
        dist.destroy_process_group()
        dog.done()


def test_one_rank_rccl_process_group_runs_the_gather(tmp_path):
    """torch.distributed over RCCL has run on this box: world size 1, the step's gather forced."""
    import torch.multiprocessing as mp
    import _oracle
    import raytracingmin_amd as rtm
    scene = _oracle.scene_path("cornellBoxSetting.json")
    dims = (88, 45, 2, 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    stage_file = tmp_path / "stage.txt"
    procs = [ctx.Process(target=_rccl_worker, args=(_free_port(), scene, dims, q, str(stage_file)))]
    procs[0].start()
    try:
        (bands, strips, ok), = _collect(procs, q, 1, timeout=150)
    except WorkersTimedOut:
        stage = stage_file.read_text() if stage_file.exists() else "(worker never started)"
        pytest.fail(f"the one-rank RCCL worker did not deliver within 150 s; its last stage: {stage!r}")
    assert ok
    data = rtm.LoadData(scene).data
    data.width, data.height, data.samples, data.superSamples = dims
    full, _ = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=0x5EED).render_rows(want=("f32",))
    assert np.array_equal(bands.view(np.uint32), full["f32"].view(np.uint32))
    assert np.array_equal(strips.view(np.uint32), full["f32"].view(np.uint32))
