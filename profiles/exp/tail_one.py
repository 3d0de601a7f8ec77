"""Headline frame, kernel time for the RTM_DEBUG_TAIL of the environment (profiles/exp/tail_sweep.sh)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import raytracingmin_amd as rtm
d = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
d.width, d.height, d.samples, d.superSamples = w, h, 64, 4
r = rtm.Renderer(d, mode="repaired", max_bounces=8, seed=0x5EED)
out, st = r.render_rows_device(0, h, want=("f32",), stats=True)
ts = []
for _ in range(4):
    out, st = r.render_rows_device(0, h, want=("f32",), stats=True)
    ts.append(st["kernel_ms"])
import hashlib
print(f"tail={os.environ.get('RTM_DEBUG_TAIL','default')} {w}x{h} split={st['split']} kernel ms {min(ts):.2f} (median {sorted(ts)[2]:.2f}) "
      f"{st['samples']/min(ts)/1e6:.3f} Gsamples/s  image sha {hashlib.sha1(out['f32'].cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
