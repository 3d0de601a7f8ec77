#!/bin/bash
# Round 4: the grid kernel's compile-time knobs again, with the next-cell fill and 1.5 cells per sphere in place:
# RTM_GRID_SHADE_AT (eighths of the busy lanes that must have finished their walks before a shading pass; default 5) and
# RTM_GRID_K (records in flight per trip; default 4).  Libraries built into ab_tmp/ by hand; BASELINE configs[4], kernel ms.
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for lib in product shade4 shade6 k5 k3; do
    if [ $lib = product ]; then unset RTM_LIB_OVERRIDE; else export RTM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/ab_tmp/librtm_$lib.so; fi
    [ $lib != product ] && [ ! -f "$RTM_LIB_OVERRIDE" ] && continue
    echo -n "$lib: "; python bench.py --workload c5 --ab 0 --steps 5 --warmup 1 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*"
  done
done
