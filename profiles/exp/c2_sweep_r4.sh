#!/bin/bash
# Round 4 (final kernels): BASELINE configs[1] (512 x 512 x 256 spp, a one-round launch of 4 096 tiles) against the sample split's knobs:
# RTM_DEBUG_TAIL (tiles split, default 1 536), RTM_DEBUG_SPLIT (granularity g), RTM_DEBUG_HEAD (wave 0's shares of 16).
# kernel ms medians of bench.py --workload c2 --ab 0,18 (exact kernel | tolerance row)
cd "$GRAFT_REPO_ROOT"
run() { echo -n "$1: "; env $2 python bench.py --workload c2 --ab 0,18 --steps 15 --warmup 3 2>/dev/null | grep -o "kernel_ms_median\": [0-9.]*" | paste - -; }
run "default" "X=1"
for t in 1024 2048 2560 3072 4096; do run "tail $t" "RTM_DEBUG_TAIL=$t"; done
for h in 6 7 8 10; do run "head $h" "RTM_DEBUG_HEAD=$h"; done
for t in 2048 3072 4096; do for h in 6 8; do run "tail $t head $h" "RTM_DEBUG_TAIL=$t RTM_DEBUG_HEAD=$h"; done; done
