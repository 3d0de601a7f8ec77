"""Round 4: how much of the configs[4] frame's time is the launch's tail?  The grid kernel runs one wave per 8x8 tile for the
whole of its 64 x spp units: 32 400 waves of ~23 ms each on 4 096 wave slots.  Same scene and frame at 64 / 128 / 256 / 512 spp:
if the tail (slots idle while the last waves finish) matters, time per sample falls with spp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracingmin_amd as rtm
data = rtm.make_stress_scene(n=100_000, seed=12345)
for spp in (64, 128, 256, 512):
    data.samples = spp
    r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=1)
    ms = []
    for k in range(4):
        _, st = r.render_rows_device(0, data.height, want=("f32",), stats=True)
        ms.append(st["kernel_ms"])
    ms = sorted(ms[1:])
    print(f"{spp} spp: kernel {ms[1]:.2f} ms, {ms[1] / spp * 256:.2f} ms per 256 spp, variant {st['variant']}")
