#!/bin/bash
# Builds of the grid kernel with other compile-time knobs, C5 full frame each (profiles/r3/grid_variants.txt, grid_crossover.txt):
#   profiles/build_ab.sh grid_K2 "-DRTM_GRID_K=2" grid_W3 "-DRTM_GRID_WPE=3" grid_S8 "-DRTM_GRID_SHADE_AT=8" ...   (here, no GPU needed)
#   gpurun -- 'profiles/exp/grid_variants.sh K2 W3 S8'
# RTM_GRID_K: records in flight per trip; RTM_GRID_WPE: waves per SIMD of the launch bound; RTM_GRID_SHADE_AT: eighths of the busy
# lanes that must have finished their walks before a shading pass (8 = lockstep); RTM_GRID_MIN: gridded spheres a scene needs.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
out=gpurun_out/r3/grid_variants.txt
: > $out
echo "default (K=4, 4 waves/SIMD, shade at 5/8):" >> $out
timeout -k 10 300 python profiles/exp/grid_tune.py 0 1080 >> $out 2>&1
for v in "$@"; do
  echo "$v:" >> $out
  RTM_LIB_OVERRIDE=$PWD/ab_libs/librtm_grid_$v.so timeout -k 10 300 python profiles/exp/grid_tune.py 0 1080 >> $out 2>&1
done
grep -v amdgpu.ids $out
