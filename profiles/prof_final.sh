# round summary: kernel-trace stats + HBM traffic counters of the default bench run
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_final
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-rows 0 > $O/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 0 --cpu-rows 0 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 0 --cpu-rows 0 > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $O/mix -- python3 $R/bench.py --steps 2 --warmup 0 --cpu-rows 0 > $O/mix.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/mix2 -- python3 $R/bench.py --steps 2 --warmup 0 --cpu-rows 0 > $O/mix2.log 2>&1
python3 - <<PY
import csv,glob,collections,json
out={}
for d in ('fetch','write','mix','mix2'):
    agg=collections.defaultdict(float); calls=0
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        rows=[r for r in csv.DictReader(open(f)) if 'render' in r['Kernel_Name']]
        disp=len(set(r['Dispatch_Id'] for r in rows))
        for r in rows: agg[r['Counter_Name']]+=float(r['Counter_Value'])
        for k,v in agg.items(): out[k]=v/max(1,disp)
stats=[r for f in glob.glob('$O/trace/*/*_kernel_stats.csv') for r in csv.DictReader(open(f))]
out['kernel_stats']=[r for r in stats if 'render' in r['Name']]
out['_note']='per launch of the default render kernel, headline frame; FETCH_SIZE/WRITE_SIZE in KB'
json.dump(out,open('$O/summary.json','w'),indent=1); print(json.dumps(out,indent=1))
PY
