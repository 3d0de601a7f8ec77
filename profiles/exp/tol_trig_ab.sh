#!/bin/bash
# Round 4: the tolerance row's sin / cos — quadrant-exact sequence on the draw's integer (product) against the Cody-Waite
# sequence on r1 (ab_tmp/librtm_trigold.so, built from the previous commit by profiles/exp/build_ab.sh trigold "" tol).
cd "$GRAFT_REPO_ROOT"
for round in 1 2 3; do
  for lib in ${LIBS:-trigold product}; do
    if [ $lib = product ]; then unset RTM_LIB_OVERRIDE; else export RTM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/ab_tmp/librtm_$lib.so; fi
    echo -n "$lib: "; python bench.py --ab 18 --steps 7 --warmup 2 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*"
  done
done
