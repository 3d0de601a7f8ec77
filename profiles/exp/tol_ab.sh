#!/bin/bash
# Same-box A/B of the exact default kernel (variant 0) against the fp64 tolerance row (variant 18), interleaved rounds in one
# process (bench.py --ab), headline frame and the 512x512x256spp frame; then the tolerance row with no primary ray flagged
# (RTM_DEBUG_TOL_PRIMFIX=0: what the exact-tie handling costs at run time); then rocprofv3's per-kernel view of the row.
# Run on the GPU box from the repo root: bash profiles/exp/tol_ab.sh > gpurun_out/r4/tol_ab.txt 2>&1
set -e
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
echo "== headline 1920x1080x1024spp, variants 0 and 18 interleaved =="
python bench.py --ab 0,18 --steps 9 --warmup 2
echo "== 512x512x256spp =="
python bench.py --workload c2 --ab 0,18 --steps 15 --warmup 3
echo "== headline, variant 18 with no primary ray flagged (RTM_DEBUG_TOL_PRIMFIX=0) against variant 0 =="
RTM_DEBUG_TOL_PRIMFIX=0 python bench.py --ab 0,18 --steps 9 --warmup 2
echo "== rocprofv3 --kernel-trace --stats, variant 18 =="
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4/prof_tol; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --variant 18 --steps 5 --warmup 2 --no-extras --cpu-rows 0 > $O/bench.log 2>&1 || true
tail -1 $O/bench.log | cut -c1-600
cat $O/trace/*/*_kernel_stats.csv | head -12
