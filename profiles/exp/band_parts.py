"""Kernel time of one rank's share of the headline frame under interleaved 8-row bands, N = 1, 2, 4, 8 (one GPU, parts in turn):
the projection of strong scaling from kernel times."""
import os, sys
sys.path.insert(0, os.getcwd())
import raytracingmin_amd as rtm
d = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
d.width, d.height, d.samples, d.superSamples = 1920, 1080, 64, 4
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # 18: the tolerance row
r = rtm.Renderer(d, mode="repaired", max_bounces=8, seed=0x5EED, variant=variant)
r.render_rows_device(0, 1080, want=("f32",), stats=True)
base = None
for n in (1, 2, 4, 8):
    worst = 0.0
    for rank in range(n):
        best = 1e9
        for _ in range(3):
            _, st = r.render_rows_device(0, 1080, want=("f32",), stats=True, band=(n, rank))
            best = min(best, st["kernel_ms"])
        worst = max(worst, best)
    base = base or worst
    print(f"N={n}: slowest part {worst:.2f} ms, split {st['split']}  => {base / worst:.2f}x of one GPU from kernel times "
          f"(variant={variant}, tail={os.environ.get('RTM_DEBUG_TAIL', 'rule')}, split={os.environ.get('RTM_DEBUG_SPLIT', 'rule')})", flush=True)
