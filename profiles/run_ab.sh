#!/bin/bash
# Same-box A/B of library builds: run_ab.sh <rounds> <bench args...> -- name1 name2 ...   (interleaved rounds)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
rounds=$1; shift
args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for n in "$@"; do
    lib=$R/ab_libs/librtm_$n.so; [ "$n" = "tree" ] && lib=$R/raytracingmin_amd/librtm_hip.so
    RTM_LIB_OVERRIDE=$lib python bench.py --cpu-rows 0 --no-extras "${args[@]}" 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read()); print('$n round $r: %.2f ms/step  kernel %.2f ms  %.0f Msamples/s' % (b['ms_per_step'], b['roofline']['kernel_ms'], b['value']))" | tee -a gpurun_out/ab.log
  done
done
