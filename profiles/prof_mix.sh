# dynamic VALU instruction mix of the default render kernel (two PMC passes)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; V=${1:-0}; O=$R/gpurun_out/prof_mix_v$V
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $O/a -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 --variant $V > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d $O/b -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 --variant $V > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VSKIPPED --output-format csv -d $O/c -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 --variant $V > $O/c.log 2>&1
python3 - <<PY
import csv,glob,collections,json
agg=collections.defaultdict(float)
for f in glob.glob('$O/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])/2
json.dump(dict(agg),open('$O/summary.json','w'),indent=1)
print(json.dumps(dict(agg),indent=1))
PY
