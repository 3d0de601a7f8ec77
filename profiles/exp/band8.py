import os, sys
sys.path.insert(0, os.getcwd())
import raytracingmin_amd as rtm
d = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
d.width, d.height, d.samples, d.superSamples = 1920, 1080, 64, 4
r = rtm.Renderer(d, mode="repaired", max_bounces=8, seed=0x5EED)
r.render_rows_device(0, 1080, want=("f32",), stats=True, band=(8, 0))
worst = 0.0
for rank in range(8):
    best = min(r.render_rows_device(0, 1080, want=("f32",), stats=True, band=(8, rank))[1]["kernel_ms"] for _ in range(3))
    worst = max(worst, best)
print(f"tail={os.environ.get('RTM_DEBUG_TAIL','rule')}: N=8 slowest part {worst:.2f} ms", flush=True)
