import os, sys
sys.path.insert(0, os.getcwd())
import raytracingmin_amd as rtm
d = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
d.width, d.height, d.samples, d.superSamples = 512, 512, 16, 4
for mb in (8, -1):
    r = rtm.Renderer(d, mode="repaired", max_bounces=mb, seed=0x5EED)
    r.render_rows_device(0, 512, want=("f32",), stats=True)
    ts = []
    for _ in range(5):
        _, st = r.render_rows_device(0, 512, want=("f32",), stats=True)
        ts.append(st["kernel_ms"])
    print(f"C2 max_bounces {mb} tail={os.environ.get('RTM_DEBUG_TAIL','rule')}: split {st['split']} kernel {min(ts):.3f} ms {st['samples']/min(ts)/1e6:.3f} Gsamples/s", flush=True)
