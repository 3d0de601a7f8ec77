# same-box, alternating processes: in-wave sample stealing on (default) / off (RTM_DEBUG_STEAL=0)
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],3))"; }
for rep in 1 2 3; do
  RTM_DEBUG_STEAL=0 python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "headline           steal off rep $rep"
  python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "headline           steal ON  rep $rep"
  RTM_DEBUG_STEAL=0 python bench.py --workload c2 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "c2 512x512x256spp  steal off rep $rep"
  python bench.py --workload c2 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "c2 512x512x256spp  steal ON  rep $rep"
  RTM_DEBUG_STEAL=0 python bench.py --rows 0:136 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "rows 0:136 (N=8)   steal off rep $rep"
  python bench.py --rows 0:136 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "rows 0:136 (N=8)   steal ON  rep $rep"
  RTM_DEBUG_STEAL=0 python bench.py --samples 4 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "1080p x 64 spp     steal off rep $rep"
  python bench.py --samples 4 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "1080p x 64 spp     steal ON  rep $rep"
done
