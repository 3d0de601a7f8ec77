#!/bin/bash
# Round 4: cells per sphere of the uniform grid (RTM_DEBUG_GRID_CELLS, default 2) with the next-cell fill in place,
# BASELINE configs[4] full frame, kernel ms medians (bench.py --workload c5 --ab 0).
cd "$GRAFT_REPO_ROOT"
for c in ${CELLS:-2 1 1.5 3 4 6 2}; do
  echo -n "cells per sphere $c: "; RTM_DEBUG_GRID_CELLS=$c python bench.py --workload c5 --ab 0 --steps 5 --warmup 1 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*"
done
