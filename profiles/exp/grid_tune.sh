#!/bin/bash
# cells-per-sphere sweep of the grid kernel on the C5 frame (full frame: a strip does not fill the chip), XCD mapping on / off
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
out=gpurun_out/r3/grid_tune.txt
: > $out
for c in 0.5 1 2 4 8; do
  RTM_DEBUG_GRID_CELLS=$c timeout -k 10 300 python profiles/exp/grid_tune.py 0 1080 >> $out 2>&1
done
echo "XCD mapping off:" >> $out
RTM_DEBUG_GRID_XCD=0 timeout -k 10 300 python profiles/exp/grid_tune.py 0 1080 >> $out 2>&1
echo "strip 508:572:" >> $out
timeout -k 10 300 python profiles/exp/grid_tune.py >> $out 2>&1
grep -v amdgpu.ids $out
