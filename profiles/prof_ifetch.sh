set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; V=${1:-0}; O=$R/gpurun_out/prof_ifetch_v$V
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_CYCLES --output-format csv -d $O/a -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 --variant $V > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES --output-format csv -d $O/b -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 --variant $V > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES --output-format csv -d $O/c -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-rows 0 --variant $V > $O/c.log 2>&1
python3 - <<PY
import csv,glob,collections,json
agg=collections.defaultdict(float)
for f in glob.glob('$O/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])/2
print(json.dumps(dict(agg),indent=1))
PY
