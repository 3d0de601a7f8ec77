import json, os, sys
sys.path.insert(0, os.getcwd())
import raytracingmin_amd as rtm
scene = os.path.join("scenes", "cornellBoxSetting.json")
data = rtm.LoadData(scene).data
data.width, data.height, data.samples, data.superSamples = 1920, 1080, 64, 4
r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=0x5EED)
r.render_rows_device(0, 8)
out = {"split": os.environ.get("RTM_DEBUG_SPLIT", "auto")}
for name, (b, e) in {"full": (0, 1080), "strip4": (544, 680), "rows8": (544, 552), "rows32": (544, 576)}.items():
    r.render_rows_device(b, e)
    st = [r.render_rows_device(b, e)[1] for _ in range(2)]
    out[name] = [round(min(s["kernel_ms"] for s in st), 2), st[0]["casts"]]
print(json.dumps(out), flush=True)
