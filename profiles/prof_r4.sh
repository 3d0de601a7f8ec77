# round-4 summary of the default bench run: kernel-trace stats + PMC passes (separate runs, the program directly after --).
# usage: prof_r4.sh <tag> [traffic-only] [bench args, e.g. --exact];  environment knobs (RTM_DEBUG_TAIL ...) pass through.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-default}; shift || true
ONLY=""; if [ "$1" = "traffic-only" ]; then ONLY=1; shift; fi
O=$R/gpurun_out/prof_r4_$TAG
rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --cpu-rows 0 --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 5 --warmup 1 > $O/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 2 --warmup 0 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 2 --warmup 0 > $O/write.log 2>&1
if [ -z "$ONLY" ]; then
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $O/mix -- $B --steps 2 --warmup 0 > $O/mix.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/mix2 -- $B --steps 2 --warmup 0 > $O/mix2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/mix3 -- $B --steps 2 --warmup 0 > $O/mix3.log 2>&1 || echo "mix3 pass failed (counter names)" >> $O/notes.txt
fi
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
for d in ('fetch','write'):
    for k,v in per_launch(d, lambda n: 'split_finalize' in n).items(): out['split_finalize.'+k]=v
    for k,v in per_launch(d, lambda n: 'steal_finalize' in n).items(): out['steal_finalize.'+k]=v
    for k,v in per_launch(d, lambda n: 'prim_prepass' in n).items(): out['prim_mask.'+k]=v
stats=[r for f in glob.glob('$O/trace/*/*_kernel_stats.csv') for r in csv.DictReader(open(f))]
out['kernel_stats']=[r for r in stats if 'render' in r['Name'] or 'split' in r['Name'] or 'steal' in r['Name'] or 'prim_prepass' in r['Name']]
try:  # the figure bench.py replays as roofline.traffic (profiles/latest_traffic.json)
    kb=lambda k: out.get(k,0.0)
    tr={'render_fetch_kb':kb('FETCH_SIZE'),'render_write_kb':kb('WRITE_SIZE'),
        'split_finalize_fetch_kb':kb('split_finalize.FETCH_SIZE'),'split_finalize_write_kb':kb('split_finalize.WRITE_SIZE'),
        'steal_finalize_fetch_kb':kb('steal_finalize.FETCH_SIZE'),'steal_finalize_write_kb':kb('steal_finalize.WRITE_SIZE'),
        'prim_mask_fetch_kb':kb('prim_mask.FETCH_SIZE'),'prim_mask_write_kb':kb('prim_mask.WRITE_SIZE')}
    tr['bytes_per_launch']=1024.0*sum(tr.values())
    tr['bytes_per_launch_with_wide_reads_doubled']=tr['bytes_per_launch']+1024.0*(tr['split_finalize_fetch_kb']+tr['steal_finalize_fetch_kb'])
    json.dump(tr,open('$O/traffic.json','w'),indent=1)
except Exception as e:
    out['traffic_error']=repr(e)
out['_note']='per launch, headline frame, the kernel of the bench arguments given (default: the tolerance row, --exact: the bit-exact kernel); FETCH_SIZE/WRITE_SIZE in KB as the counters report them (x 1024 = bytes)'
json.dump(out,open('$O/summary.json','w'),indent=1); print(json.dumps(out,indent=1))
PY
