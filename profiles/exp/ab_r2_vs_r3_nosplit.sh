# same-box, alternating processes, NO sample split (RTM_DEBUG_TAIL=0): isolates the main loop from the split's bookkeeping
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],2), 'kernel', round(j['roofline']['kernel_ms'],2))"; }
for rep in 1 2 3; do
  (cd ab_libs/r2tree && RTM_DEBUG_TAIL=0 python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null) | one "r2-tree tail 0    rep $rep"
  RTM_DEBUG_TAIL=0 python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "r3 tail 0         rep $rep"
done
