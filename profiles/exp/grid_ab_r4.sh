#!/bin/bash
# Round 4: same-box A/B of grid-kernel twins built by profiles/exp/build_ab.sh (ab_tmp/librtm_<name>.so): BASELINE
# configs[4], full frame, kernel ms medians, alternating processes.  usage: grid_ab_r4.sh name1 name2 ...  ("product" = the tree's library)
cd "$GRAFT_REPO_ROOT"
for round in 1 2 3; do
  for lib in "$@"; do
    if [ $lib = product ]; then unset RTM_LIB_OVERRIDE; else export RTM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/ab_tmp/librtm_$lib.so; fi
    echo -n "$lib: "; python bench.py --workload c5 --ab 0 --steps 5 --warmup 1 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*"
  done
done
