"""C5 (100 000-sphere stress scene, 1080p x 256 spp, cap 8) through the grid kernel (variant 17): time of a 64-row strip
for a given RTM_DEBUG_GRID_CELLS (cells per sphere; read once per process, so one process per value — grid_tune.sh)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import raytracingmin_amd as rtm  # noqa: E402

rows = (508, 572) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
variant = int(os.environ.get("GRID_TUNE_VARIANT", "17"))
data = rtm.make_stress_scene(n=100_000, seed=12345)
data.width, data.height, data.samples, data.superSamples = 1920, 1080, 256, 1
t0 = time.perf_counter()
r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=0x5EED, variant=variant)
out, st = r.render_rows_device(rows[0], rows[1], want=("f32",))
t_first = time.perf_counter() - t0
best = 1e30
for _ in range(3):
    out, st = r.render_rows_device(rows[0], rows[1], want=("f32",))
    best = min(best, st["kernel_ms"])
samples = st["samples"]
print(f"cells/sphere {os.environ.get('RTM_DEBUG_GRID_CELLS', 'default')}: rows {rows[0]}:{rows[1]} variant {st['variant']} "
      f"{best:.2f} ms  {samples / best * 1e-3:.1f} Msamples/s  casts/sample {st['casts'] / samples:.3f}  "
      f"(first call incl. scene + grid build {t_first * 1e3:.0f} ms)", flush=True)
