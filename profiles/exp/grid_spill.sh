#!/bin/bash
# Same-box A/B of the grid kernel's walk (BASELINE configs[4], full frame): the product build against builds with
#   RTM_GRID_SPILL=1        free slots of a trip filled from the NEXT cell's list (rtm_path.h: GridWalk::advance)
#   RTM_GRID_K=6            six records in flight per trip instead of four
# built into ab_tmp/ by hand (see the commands in profiles/r4/grid_spill.txt).  Alternating processes, kernel ms medians.
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for lib in product spill1 k6 spill1k6; do
    if [ $lib = product ]; then unset RTM_LIB_OVERRIDE; else export RTM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/ab_tmp/librtm_$lib.so; fi
    [ $lib != product ] && [ ! -f "$RTM_LIB_OVERRIDE" ] && continue
    echo -n "$lib: "; python bench.py --workload c5 --ab 0 --steps 5 --warmup 1 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*"
  done
done
