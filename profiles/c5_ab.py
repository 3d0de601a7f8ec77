import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
import raytracingmin_amd as rtm
stress = rtm.make_stress_scene(n=100_000, seed=12345)
stress.width, stress.height, stress.samples, stress.superSamples = 1920, 1080, 256, 1
r = rtm.Renderer(stress, mode="repaired", max_bounces=8, seed=0x5EED)
r.render_rows_device(508, 516, want=("f32",), stats=True)
for rows in ((508, 572), (400, 656)):
    t0 = time.perf_counter()
    _, st = r.render_rows_device(rows[0], rows[1], want=("f32",), stats=True)
    dt = time.perf_counter() - t0
    print(os.environ.get("RTM_LIB_OVERRIDE", "tree").split("/")[-1], rows, "%.1f ms  %.2f Msamples/s  %.3e tests/s" % (dt*1e3, st["samples"]/dt/1e6, st["casts"]*1e5/dt), flush=True)
