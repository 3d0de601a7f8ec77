"""How much of the headline launch is its tail?  Same scene and spp, frames of different tile counts:
rate(frame) = samples / kernel time; a frame of many more rounds of waves has a proportionally smaller tail."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import raytracingmin_amd as rtm
base = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
for (w, h) in ((1920, 1080), (2048, 1024), (1920, 952), (3840, 2160), (960, 540), (1920, 544)):
    d = base
    d.width, d.height, d.samples, d.superSamples = w, h, 64, 4
    r = rtm.Renderer(d, mode="repaired", max_bounces=8, seed=0x5EED)
    r.render_rows_device(0, h, want=("f32",), stats=True)
    best = 1e9
    for _ in range(3):
        _, st = r.render_rows_device(0, h, want=("f32",), stats=True)
        best = min(best, st["kernel_ms"])
    tiles = ((w + 7) // 8) * ((h + 7) // 8)
    print(f"{w}x{h}: tiles {tiles} = {tiles/4096:.2f} rounds of 4096 waves, split {st['split']}, kernel {best:.2f} ms, "
          f"{st['samples']/best/1e6:.3f} Gsamples/s, {best/tiles*4096:.2f} ms per round-equivalent", flush=True)
