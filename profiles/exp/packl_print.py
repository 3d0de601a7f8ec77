import os, sys
sys.path.insert(0, os.getcwd())
import raytracingmin_amd as rtm
d = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
d.width, d.height, d.samples, d.superSamples = 200, 120, 8, 2
out, st = rtm.Renderer(d, mode="repaired", max_bounces=-1, seed=0x5EED, variant=0).render_rows(0, 120, want=("f64",))
print("pixel", out["f64"][30, 179].tolist())
