#!/bin/bash
# Round 4: layout and cache policy of the pre-pass's table of primary directions (RenderParams::prim_dirs), same box,
# alternating processes: A component-major ([tile][sub][component][pixel]) with plain loads, B the same with non-temporal
# loads, C pixel-major ([tile][pixel][sub][component]) non-temporal, D pixel-major plain; "off": no table.  Kernel ms medians.
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for lib in A B C D; do
    export RTM_LIB_OVERRIDE=$GRAFT_REPO_ROOT/ab_tmp/librtm_$lib.so
    echo -n "$lib: "; python bench.py --ab 0,18 --steps 7 --warmup 2 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*" | paste - -
  done
  echo -n "off: "; RTM_DEBUG_PRIM_DIRS=0 python bench.py --ab 0,18 --steps 7 --warmup 2 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*" | paste - -
done
