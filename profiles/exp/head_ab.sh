# same-box, alternating: shares (of 16) wave 0 of a split tile keeps — RTM_DEBUG_HEAD — headline, c2, one GPU's share of eight
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],3), 'split', j['config']['sample_split_waves_per_tile'])"; }
for rep in 1 2; do for H in 8 9 10 11; do
  RTM_DEBUG_HEAD=$H python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "headline head $H/16 rep $rep"
  RTM_DEBUG_HEAD=$H python bench.py --workload c2 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "c2       head $H/16 rep $rep"
  RTM_DEBUG_HEAD=$H python bench.py --rows 0:136 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "rows0:136 head $H/16 rep $rep"
done; done
