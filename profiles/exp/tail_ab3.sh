# same-box, alternating processes: number of split tiles of the headline launch (RTM_DEBUG_TAIL), round-3 term layout
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for T in 1024 1536 768; do
  RTM_DEBUG_TAIL=$T python bench.py --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('tail $T rep $rep', round(j['ms_per_step'],2))"
done; done
