# same-box, alternating processes: round 2's tree (ab_libs/r2tree, built from commit 45274d2) against this tree, headline frame;
# the round-3 tree with 1 024 (default) and 1 536 split tiles
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],2), 'kernel', round(j['roofline']['kernel_ms'],2))"; }
for rep in 1 2 3 4; do
  (cd ab_libs/r2tree && python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null) | one "r2-tree          rep $rep"
  python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "r3 tail 1024     rep $rep"
  RTM_DEBUG_TAIL=1536 python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "r3 tail 1536     rep $rep"
done
