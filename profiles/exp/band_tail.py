"""Kernel time of the slowest of the N ranks' interleaved-band parts of the headline frame for the RTM_DEBUG_TAIL of the environment
(one process per value: the knob is read once)."""
import os, sys
sys.path.insert(0, os.getcwd())
import raytracingmin_amd as rtm
d = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
d.width, d.height, d.samples, d.superSamples = 1920, 1080, 64, 4
r = rtm.Renderer(d, mode="repaired", max_bounces=8, seed=0x5EED)
r.render_rows_device(0, 1080, want=("f32",), stats=True)
for n in (int(a) for a in sys.argv[1:]):
    parts = []
    for rank in range(n):
        parts.append(min(r.render_rows_device(0, 1080, want=("f32",), stats=True, band=(n, rank))[1]["kernel_ms"] for _ in range(3)))
    print(f"tail {os.environ.get('RTM_DEBUG_TAIL', 'rule'):>5s}  N={n}: parts " + " ".join(f"{p:.2f}" for p in parts) + f"  slowest {max(parts):.2f} ms", flush=True)
