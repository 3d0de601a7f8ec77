"""Kernel time of the N=8 strips of the headline frame (and of BASELINE configs[1], 512x512x256spp)
with the sample split forced to RTM_DEBUG_SPLIT waves per tile (read once per process: run one
process per setting, see split_sweep.sh)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracingmin_amd as rtm
from raytracingmin_amd.distributed import partition_rows
scene = os.path.join(os.path.dirname(__file__), "..", "scenes", "cornellBoxSetting.json")
data = rtm.LoadData(scene).data
data.width, data.height, data.samples, data.superSamples = 1920, 1080, 64, 4
r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=0x5EED)
r.render_rows_device(0, 8)
out = {"RTM_DEBUG_SPLIT": os.environ.get("RTM_DEBUG_SPLIT", "auto")}
out["full_ms"] = round(min(r.render_rows_device(0, 1080)[1]["kernel_ms"] for _ in range(2)), 2)
for n in (8, 4, 2):
    ms = []
    for b, e in partition_rows(1080, n):
        r.render_rows_device(b, e)
        ms.append(min(r.render_rows_device(b, e)[1]["kernel_ms"] for _ in range(2)))
    out[f"strips{n}_ms"] = [round(m, 2) for m in ms]
    out[f"strips{n}_max"] = round(max(ms), 2)
    ms = []
    for k in range(n):  # the same frame dealt out in interleaved 8-row bands
        r.render_rows_device(0, 1080, band=(n, k))
        ms.append(min(r.render_rows_device(0, 1080, band=(n, k))[1]["kernel_ms"] for _ in range(2)))
    out[f"bands{n}_ms"] = [round(m, 2) for m in ms]
    out[f"bands{n}_speedup_from_kernel_times"] = round(out["full_ms"] / max(ms), 2)
d2 = rtm.LoadData(scene).data
d2.width, d2.height, d2.samples, d2.superSamples = 512, 512, 16, 4
r2 = rtm.Renderer(d2, mode="repaired", max_bounces=8, seed=0x5EED)
r2.render_rows_device(0, 512)
out["c2_512x512x256_ms"] = round(min(r2.render_rows_device(0, 512)[1]["kernel_ms"] for _ in range(3)), 2)
print(json.dumps(out), flush=True)
