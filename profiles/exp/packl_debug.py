import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import raytracingmin_amd as rtm
d = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
for (w, h, s, ss) in ((200, 120, 8, 2), (61, 37, 5, 3), (200, 120, 64, 2)):
    d.width, d.height, d.samples, d.superSamples = w, h, s, ss
    ref, rs = rtm.Renderer(d, mode="repaired", max_bounces=-1, seed=0x5EED, variant=1).render_rows(0, h, want=("f64",))
    for v in (0, 2, 9, 14):
        out, st = rtm.Renderer(d, mode="repaired", max_bounces=-1, seed=0x5EED, variant=v).render_rows(0, h, want=("f64",))
        diff = np.argwhere((out["f64"] != ref["f64"]).any(axis=2))
        print((w, h, s, ss), "variant", v, "split", st["split"], "pixels differing", len(diff), diff[:5].tolist(),
              "casts equal", st["casts"] == rs["casts"], "max delta", float(np.abs(out["f64"] - ref["f64"]).max()), flush=True)
