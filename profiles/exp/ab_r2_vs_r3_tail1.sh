# same-box: the SPLIT instantiation with ONE split tile (RTM_DEBUG_TAIL=64: whole waves only, in effect) isolates what the
# split's bookkeeping costs a whole wave per trip; and c2 with 1536 split tiles in both trees
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],2), 'kernel', round(j['roofline']['kernel_ms'],2), 'split', j['config']['sample_split_waves_per_tile'])"; }
for rep in 1 2 3; do
  (cd ab_libs/r2tree && RTM_DEBUG_TAIL=64 python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null) | one "r2-tree tail 64   rep $rep"
  RTM_DEBUG_TAIL=64 python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "r3      tail 64   rep $rep"
  RTM_DEBUG_TAIL=1536 python bench.py --workload c2 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "r3 c2   tail 1536 rep $rep"
done
