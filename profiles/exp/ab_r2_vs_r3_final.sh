# same-box, alternating processes: round 2's tree (ab_libs/r2tree, commit 45274d2) against this tree — headline frame with the
# default rule (the last 1 536 tiles split in both), the SPLIT kernel with whole waves only (64 split tiles), no split, c2
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],2), 'kernel', round(j['roofline']['kernel_ms'],2), 'split', j['config']['sample_split_waves_per_tile'])"; }
for rep in 1 2 3; do
  (cd ab_libs/r2tree && python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null) | one "r2 default          rep $rep"
  python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "r3 default          rep $rep"
  (cd ab_libs/r2tree && RTM_DEBUG_TAIL=64 python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null) | one "r2 tail 64          rep $rep"
  RTM_DEBUG_TAIL=64 python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "r3 tail 64          rep $rep"
  (cd ab_libs/r2tree && python bench.py --workload c2 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null) | one "r2 c2               rep $rep"
  python bench.py --workload c2 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "r3 c2               rep $rep"
  (cd ab_libs/r2tree && python bench.py --rows 0:136 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null) | one "r2 rows 0:136       rep $rep"
  python bench.py --rows 0:136 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "r3 rows 0:136       rep $rep"
done
