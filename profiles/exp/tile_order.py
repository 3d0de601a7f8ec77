"""EXPERIMENT: what a cost-aware tile order would buy.  Per-tile cost = the oracle's casts for the tile's pixels at `spp_est` samples
(CPU, this container or the GPU box); order = whole tiles in descending cost, then the `tail` MOST expensive tiles, which the sample
split cuts (they come last in the launch's logical order).  Writes a u32 file for RTM_DEBUG_TILE_ORDER_FILE — a knob of a library built with
profiles/r3/tile_order_experiment.patch applied (git apply; the product does not read it).  Result: profiles/r3/tail_after_steal.txt.
    python profiles/exp/tile_order.py <width> <height> <row_begin> <row_end> <bands N> <band index> <tail> <out file> [mode]
mode: "expensive-split" (default) or "cheap-split" (descending cost throughout: the cheapest tiles are the split ones)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import _oracle  # noqa: E402

W, H, r0, r1, nb, bi, tail, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), \
    int(sys.argv[7]), sys.argv[8]
mode = sys.argv[9] if len(sys.argv) > 9 else "expensive-split"
st, arr, n = _oracle.load_scene(_oracle.scene_path("cornellBoxSetting.json"), width=W, height=H, samples=1, super_samples=4)
opt = _oracle.make_options(mode=1, max_bounces=8, seed=0x5EED, height=H)
# rows of this call's bands, 8-row bands dealt to nb ranks
rows = [r for r in range(r0, r1) if ((r - r0) // 8) % nb == bi]
tiles_x = (W + 7) // 8
tiles_y = (len(rows) + 7) // 8
cost = np.zeros(tiles_x * tiles_y)
for ty in range(tiles_y):
    band = rows[ty * 8:(ty + 1) * 8]
    for tx in range(tiles_x):
        xy = [(x, y) for y in band for x in range(tx * 8, min(tx * 8 + 8, W))]
        _, cnt = _oracle.render_pixels(st, arr, n, opt, np.array(xy, dtype=np.int32))
        cost[ty * tiles_x + tx] = cnt["casts"]
order = np.argsort(-cost, kind="stable").astype(np.uint32)  # descending cost
if mode == "expensive-split":
    tail = min(tail, len(order))
    order = np.concatenate([order[tail:], order[:tail]])
order.tofile(out)
print(f"{len(order)} tiles, cost per tile min {cost.min():.0f} mean {cost.mean():.1f} max {cost.max():.0f} (max/mean {cost.max() / cost.mean():.3f}), "
      f"mode {mode}, tail {tail} -> {out}")
