# EXPERIMENT: kernel ms with a cost-aware tile order (files made by profiles/exp/tile_order.py into ab_libs/orders/) against the plain order.
# Needs a library built with profiles/r3/tile_order_experiment.patch applied (the product does not read RTM_DEBUG_TILE_ORDER_FILE).
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],3))"; }
for o in none c2_exp c2_cheap; do
  if [ $o = none ]; then unset RTM_DEBUG_TILE_ORDER_FILE; else export RTM_DEBUG_TILE_ORDER_FILE=$PWD/ab_libs/orders/$o.u32; fi
  python bench.py --workload c2 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "c2 512x512x256spp order $o"
done
for o in none full_exp full_cheap; do
  if [ $o = none ]; then unset RTM_DEBUG_TILE_ORDER_FILE; else export RTM_DEBUG_TILE_ORDER_FILE=$PWD/ab_libs/orders/$o.u32; fi
  python bench.py --no-extras --cpu-rows 0 --steps 6 --warmup 1 2>/dev/null | one "headline order $o"
done
for o in none n8p0_exp n8p0_cheap; do
  if [ $o = none ]; then unset RTM_DEBUG_TILE_ORDER_FILE; else export RTM_DEBUG_TILE_ORDER_FILE=$PWD/ab_libs/orders/$o.u32; fi
  python - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
import raytracingmin_amd as rtm
d = rtm.LoadData(os.path.join("scenes", "cornellBoxSetting.json")).data
d.width, d.height, d.samples, d.superSamples = 1920, 1080, 64, 4
r = rtm.Renderer(d, mode="repaired", max_bounces=8, seed=0x5EED)
r.render_rows_device(0, 1080, want=("f32",), stats=True, band=(8, 0))
print("N=8 part 0 order $o", round(min(r.render_rows_device(0, 1080, want=("f32",), stats=True, band=(8, 0))[1]["kernel_ms"] for _ in range(5)), 3))
PY
done
