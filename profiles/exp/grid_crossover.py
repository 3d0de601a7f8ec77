"""Where the grid kernel (variant 17) overtakes the exhaustive kernels for scenes of a few hundred to a few thousand
spheres: stress-scene family at 1080p x 16 spp, cap 8, kernel time per variant."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracingmin_amd as rtm  # noqa: E402

for n in (257, 300, 400, 511, 512, 1000, 3000, 10000):
    data = rtm.make_stress_scene(n=n, seed=12345)
    data.width, data.height, data.samples, data.superSamples = 1920, 1080, 16, 1
    row = []
    for v in (3, 12, 17):
        if v == 12 and n < 512:
            row.append("      -")
            continue
        r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=1, variant=v)
        r.render_rows_device(want=("f32",))
        best = min(r.render_rows_device(want=("f32",))[1]["kernel_ms"] for _ in range(3))
        row.append(f"{best:7.2f}")
    print(f"n={n:6d}  variant 3 (chunked loop, global tables) {row[0]} ms   12 (exhaustive pipeline) {row[1]} ms   17 (grid) {row[2]} ms", flush=True)
