import os, sys
sys.path.insert(0, os.getcwd())
import raytracingmin_amd as rtm
d = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
for (w,h,s,ss,label) in ((512,512,16,4,"c2 512x512x256 unlimited"),(1920,1080,64,4,"headline unlimited")):
    d.width, d.height, d.samples, d.superSamples = w,h,s,ss
    for v in (0,18):
        r = rtm.Renderer(d, mode="repaired", max_bounces=-1, seed=0x5EED, variant=v)
        r.render_rows_device(0, h, want=("f32",), stats=True)
        best=1e9
        for _ in range(5):
            _, st = r.render_rows_device(0, h, want=("f32",), stats=True)
            best=min(best, st["kernel_ms"])
        print(label, "variant", v, "kernel %.3f ms" % best, flush=True)
