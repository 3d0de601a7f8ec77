"""Per-strip kernel time of the headline frame cut into N contiguous row strips (one GPU, strips one
after another): max/mean is the load-balance loss an N-GPU run would see."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracingmin_amd as rtm
from raytracingmin_amd.distributed import partition_rows
data = rtm.LoadData(os.path.join(os.path.dirname(__file__), "..", "scenes", "cornellBoxSetting.json")).data
data.width, data.height, data.samples, data.superSamples = 1920, 1080, 64, 4
r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=0x5EED)
r.render_rows_device(0, 8)
_, full = r.render_rows_device(0, 1080)
for n in (2, 4, 8):
    ms = []
    for b, e in partition_rows(1080, n):
        _, st = r.render_rows_device(b, e)
        ms.append(st["kernel_ms"])
    print(json.dumps({"strips": n, "full_ms": full["kernel_ms"], "max_ms": max(ms), "mean_ms": sum(ms) / n,
                      "ideal_speedup": n, "speedup_from_kernel_times": full["kernel_ms"] / max(ms),
                      "strip_ms": [round(m, 1) for m in ms]}))
