#!/bin/bash
# builds of the grid kernel with other batch sizes (RTM_GRID_K) / launch bounds (RTM_GRID_WPE): C5 full frame each
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
out=gpurun_out/r3/grid_variants.txt
: > $out
echo "default (K=4, 4 waves/SIMD, shade at 4/8):" >> $out
timeout -k 10 300 python profiles/exp/grid_tune.py 0 1080 >> $out 2>&1
for v in K2 K3 K2W5; do
  echo "$v:" >> $out
  RTM_LIB_OVERRIDE=$PWD/ab_libs/librtm_grid_$v.so timeout -k 10 300 python profiles/exp/grid_tune.py 0 1080 >> $out 2>&1
done
grep -v amdgpu.ids $out
