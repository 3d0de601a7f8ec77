#!/bin/bash
# Build a twin of librtm_hip.so with extra -D flags into ab_tmp/librtm_<name>.so (same-box A/B through RTM_LIB_OVERRIDE).
# usage: build_ab.sh <name> "<flags>" [tol]   ("tol": the tolerance translation unit is rebuilt with the flags too)
set -e
cd "$(dirname "$0")/../../raytracingmin_amd/csrc"
name=$1; flags=$2
mkdir -p ../../ab_tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-parameter $flags -c rtm_kernels.hip -o /tmp/rtm_kernels_$name.o
tolo=rtm_kernels_tol.o
if [ "$3" = tol ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -Wno-unused-parameter $flags -c rtm_kernels_tol.hip -o /tmp/rtm_kernels_tol_$name.o
  tolo=/tmp/rtm_kernels_tol_$name.o
fi
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/rtm_kernels_$name.o $tolo rtm_api.o rtm_scene.o rtm_image.o -o ../../ab_tmp/librtm_$name.so
echo built ab_tmp/librtm_$name.so
