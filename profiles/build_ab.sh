#!/bin/bash
# Builds A/B copies of librtm_hip.so with different -D switches into ab_libs/ (travels to the GPU box, not committed):
#   profiles/build_ab.sh name1 "-DRTM_OPT_GUARD=0 ..." name2 "..." ...
# Run them with RTM_LIB_OVERRIDE=ab_libs/librtm_<name>.so python bench.py ...  (profiles/run_ab.sh)
set -e
cd "$(dirname "$0")/../raytracingmin_amd/csrc"
mkdir -p ../../ab_libs
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-parameter $flags -c rtm_kernels.hip -o /tmp/rtm_kernels_$name.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/rtm_kernels_$name.o rtm_api.o rtm_scene.o rtm_image.o -o ../../ab_libs/librtm_$name.so &&
    echo "built $name: $flags" ) &
done
wait
