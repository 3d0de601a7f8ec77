#!/bin/bash
# Round 4 (final kernels): the number of a launch's last tiles that are sample-split (RTM_DEBUG_TAIL, default 1 536) and the shares
# wave 0 of a split tile keeps (RTM_DEBUG_HEAD of 16, default 9), re-checked for the tolerance row and the exact kernel: headline frame
# (bench.py --ab 0,18 kernel ms medians) and one GPU's share of eight (profiles/exp/band_parts.py, slowest part).
cd "$GRAFT_REPO_ROOT"
for t in 1024 1536 2048 3072; do
  echo -n "tail $t: "; RTM_DEBUG_TAIL=$t python bench.py --ab 0,18 --steps 5 --warmup 1 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*" | paste - -
done
for h in 8 9 10 11; do
  echo -n "head $h: "; RTM_DEBUG_HEAD=$h python bench.py --ab 0,18 --steps 5 --warmup 1 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*" | paste - -
done
for t in 1024 1536 2048 3072; do
  echo "tail $t, band parts (variant 18):"; RTM_DEBUG_TAIL=$t python profiles/exp/band_parts.py 18 2>/dev/null | grep "N=[48]"
done
