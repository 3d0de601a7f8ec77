# same-box: EVERY tile sample-split (variant 9) — amplifies what a small wave costs — r2 tree against this tree; and the
# 512x512x256spp frame (4096 tiles: split whole by the default rule)
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],2), 'kernel', round(j['roofline']['kernel_ms'],2), 'split', j['config']['sample_split_waves_per_tile'])"; }
for rep in 1 2; do
  (cd ab_libs/r2tree && python bench.py --variant 9 --rows 0:272 --no-extras --cpu-rows 0 --steps 8 --warmup 2 2>/dev/null) | one "r2-tree v9 rows 0:272   rep $rep"
  python bench.py --variant 9 --rows 0:272 --no-extras --cpu-rows 0 --steps 8 --warmup 2 2>/dev/null | one "r3      v9 rows 0:272   rep $rep"
  (cd ab_libs/r2tree && python bench.py --workload c2 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null) | one "r2-tree c2              rep $rep"
  python bench.py --workload c2 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "r3      c2              rep $rep"
done
