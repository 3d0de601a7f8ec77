set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1/trace -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rows 0 > $R/gpurun_out/prof_r1/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/prof_r1/pmc1 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 > $R/gpurun_out/prof_r1/bench_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/prof_r1/pmc2 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 > $R/gpurun_out/prof_r1/bench_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_r1/pmc3 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 > $R/gpurun_out/prof_r1/bench_pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r1/pmc4 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 > $R/gpurun_out/prof_r1/bench_pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r1/pmc5 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 > $R/gpurun_out/prof_r1/bench_pmc5.log 2>&1
find $R/gpurun_out/prof_r1 -name "*.csv" | head -40
