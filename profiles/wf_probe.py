import sys, json, os
sys.path.insert(0, os.getcwd())
import raytracingmin_amd as rtm
v = int(sys.argv[1])
data = rtm.make_stress_scene(n=100000, seed=12345)
data.width, data.height, data.samples, data.superSamples = 1920, 512, 2, 1
r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=5, variant=v)
out, st = r.render_rows_device(want=("f64",), stats=True)
print(json.dumps(st))
