# round-2 summary: kernel-trace stats + PMC passes (separate runs, program directly after --) of the default
# bench run, plus one PMC pass of a SPLIT launch (one eighth of the frame).  usage: prof_r2.sh <tag> [bench args]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-default}; shift || true
O=$R/gpurun_out/prof_r2_$TAG
rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --cpu-rows 0 --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 5 --warmup 1 > $O/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 2 --warmup 0 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 2 --warmup 0 > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $O/mix -- $B --steps 2 --warmup 0 > $O/mix.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/mix2 -- $B --steps 2 --warmup 0 > $O/mix2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/mix3 -- $B --steps 2 --warmup 0 > $O/mix3.log 2>&1 || echo "mix3 pass failed (counter names)" >> $O/notes.txt
# a split launch: rows 0:136 = 4080 tiles, one eighth of the frame (what one rank of eight gets in tile count)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/split_fetch -- $B --rows 0:136 --steps 2 --warmup 0 > $O/split_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/split_write -- $B --rows 0:136 --steps 2 --warmup 0 > $O/split_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/split_trace -- $B --rows 0:136 --steps 5 --warmup 1 > $O/split_trace.log 2>&1
python3 - <<PY
import csv,glob,collections,json
out={}
def per_launch(d, match):
    res={}
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        rows=[r for r in csv.DictReader(open(f)) if match(r['Kernel_Name'])]
        agg=collections.defaultdict(float)
        disp=len(set(r['Dispatch_Id'] for r in rows))
        for r in rows: agg[r['Counter_Name']]+=float(r['Counter_Value'])
        for k,v in agg.items(): res[k]=v/max(1,disp)
    return res
for d in ('fetch','write','mix','mix2','mix3'):
    out.update(per_launch(d, lambda n: 'render_tiles' in n))
stats=[r for f in glob.glob('$O/trace/*/*_kernel_stats.csv') for r in csv.DictReader(open(f))]
out['kernel_stats']=[r for r in stats if 'render' in r['Name'] or 'split' in r['Name']]
sp={}
for d in ('split_fetch','split_write'):
    for name,match in (('render_tiles', lambda n: 'render_tiles' in n), ('split_finalize', lambda n: 'split_finalize' in n)):
        for k,v in per_launch(d, match).items(): sp[name+'.'+k]=v
sp['kernel_stats']=[r for f in glob.glob('$O/split_trace/*/*_kernel_stats.csv') for r in csv.DictReader(open(f)) if 'render' in r['Name'] or 'split' in r['Name']]
out['split_launch_rows_0_136']=sp
out['_note']='per launch, headline frame, default kernel; FETCH_SIZE/WRITE_SIZE in KB as the counters report them; split_launch_*: rows 0:136 (4080 tiles) with the sample split'
json.dump(out,open('$O/summary.json','w'),indent=1); print(json.dumps(out,indent=1))
PY
