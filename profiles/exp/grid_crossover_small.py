"""Below 257 spheres (a build with -DRTM_GRID_MIN=32): the grid kernel against what variant 0 picks there (the packed-record
kernel with global tables, variant 14), stress-scene family and boxes packed with small spheres, 1080p x 16 spp, cap 8."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import raytracingmin_amd as rtm  # noqa: E402

box = rtm.LoadData(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scenes", "cornellBoxSetting.json")).data
rng = np.random.default_rng(5)
for kind in ("stress", "box"):
    for n in (40, 64, 100, 160, 255):
        if kind == "stress":
            data = rtm.make_stress_scene(n=n, seed=12345)
        else:
            objs = list(box.object)[:7]
            while len(objs) < n:
                objs.append(rtm.SphereObject(rtm.vec3(*map(float, rng.uniform(-7, 7, 3))), float(rng.uniform(0.3, 1.0)),
                                             rtm.Material(rtm.vec3(0.6, 0.6, 0.6), rtm.vec3(0, 0, 0))))
            data = rtm.SettingData(width=8, height=8, samples=1, superSamples=1, camera=box.camera, object=objs)
        data.width, data.height, data.samples, data.superSamples = 1920, 1080, 16, 1
        row = []
        for v in (14, 17):
            try:
                r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=1, variant=v)
                r.render_rows_device(want=("f32",))
                best = min(r.render_rows_device(want=("f32",))[1]["kernel_ms"] for _ in range(3))
                row.append(f"{best:7.2f}")
            except rtm.RtmError as e:
                row.append("   none")
        print(f"{kind:6s} n={n:4d}  variant 14 (packed records, global tables) {row[0]} ms   17 (grid) {row[1]} ms", flush=True)
