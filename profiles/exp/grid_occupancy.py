"""One configs[4] frame through a -DRTM_GRID_EXP_OCC build of the grid kernel (profiles/build_ab.sh grid_OCC "-DRTM_GRID_EXP_OCC=1";
RTM_LIB_OVERRIDE=ab_libs/librtm_grid_OCC.so): grid_finalize_kernel prints, per region of render_grid_kernel, how often it ran
(wave level) and with how many lanes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401
import raytracingmin_amd as rtm  # noqa: E402

data = rtm.make_stress_scene(n=100_000, seed=12345)
data.width, data.height, data.samples, data.superSamples = 1920, 1080, 256, 1
r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=0x5EED, variant=17)
out, st = r.render_rows_device(0, 1080, want=("f32",))
torch.cuda.synchronize()
print(f"kernel {st['kernel_ms']:.1f} ms (instrumented), casts {st['casts']}, samples {st['samples']}")
