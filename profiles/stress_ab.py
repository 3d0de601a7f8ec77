import sys, json
import numpy as np
sys.path.insert(0, '.')
import raytracingmin_amd as rtm
n = int(sys.argv[1]); w=int(sys.argv[2]); h=int(sys.argv[3]); s=int(sys.argv[4])
data = rtm.make_stress_scene(n=n, seed=12345)
if len(sys.argv) > 6 and sys.argv[6] == "box":
    # a closed scene of n spheres: the Cornell box's seven plus n - 7 small ones inside it
    import os
    box = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
    rng = np.random.default_rng(1)
    objs = list(box.object)
    for k in range(n - len(objs)):
        c = rng.uniform(-7, 7, 3)
        col = rng.uniform(0.2, 0.9, 3)
        objs.append(rtm.SphereObject(rtm.vec3(*c), float(rng.uniform(0.3, 1.0)), rtm.Material(rtm.vec3(*col), rtm.vec3(0, 0, 0))))
    box.object = objs
    data = box
data.width, data.height, data.samples, data.superSamples = w, h, s, 1
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
