#!/bin/bash
# Round 4: axis-pattern discriminants (rtm_path.h: sphere_disc) against the tree without them, same box, alternating
# processes; kernel ms medians of the exact kernel (variant 0) and the tolerance row (variant 18).
#   bash profiles/exp/build_ab.sh axis0 "-DRTM_OPT_AXIS=0" tol ; axis_ab.sh axis0 tree [more libs]
cd "$GRAFT_REPO_ROOT"
for round in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = tree ]; then export RTM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/raytracingmin_amd/librtm_hip.so; else export RTM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/ab_tmp/librtm_$lib.so; fi
    echo -n "$lib: "; python bench.py --ab 0,18 --steps 7 --warmup 2 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*" | paste - -
  done
done
