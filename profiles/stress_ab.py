import sys, json
sys.path.insert(0, '.')
import raytracingmin_amd as rtm
n = int(sys.argv[1]); w=int(sys.argv[2]); h=int(sys.argv[3]); s=int(sys.argv[4])
data = rtm.make_stress_scene(n=n, seed=12345)
data.width, data.height, data.samples, data.superSamples = w, h, s, 1
import numpy as np
base=None
for v in [int(a) for a in sys.argv[5].split(',')]:
    r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=5, variant=v)
    ts=[]
    for k in range(3):
        out, st = r.render_rows_device(want=("f64",), stats=True)
        ts.append(st["kernel_ms"])
    img = out["f64"].cpu().numpy()
    if base is None: base = img
    print(json.dumps({"variant": v, "name": rtm.lib().rtm_variant_name(v).decode(), "kernel_ms": sorted(ts)[1], "casts_per_sample": st["casts"]/st["samples"],
                      "Msamples/s": st["samples"]/sorted(ts)[1]/1e3, "G sphere tests/s": st["casts"]*n/sorted(ts)[1]/1e6, "same_bits": bool(np.array_equal(img.view(np.uint64), base.view(np.uint64)))}), flush=True)
