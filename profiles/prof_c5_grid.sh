# C5 (100 000 spheres, 1080p x 256 spp) through the grid kernel: kernel-trace stats + PMC passes (separate runs, the
# program directly after --).  usage: prof_c5_grid.sh <tag>;  environment knobs pass through.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-c5grid}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --workload c5 --cpu-rows 0 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 3 --warmup 1 > $O/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- $B --steps 2 --warmup 0 > $O/sq.log 2>&1 || echo "sq pass failed" >> $O/notes.txt
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -- $B --steps 2 --warmup 0 > $O/l2.log 2>&1 || echo "l2 pass failed" >> $O/notes.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 2 --warmup 0 > $O/fetch.log 2>&1 || echo "fetch pass failed" >> $O/notes.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 2 --warmup 0 > $O/write.log 2>&1 || echo "write pass failed" >> $O/notes.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH --output-format csv -d $O/mix -- $B --steps 2 --warmup 0 > $O/mix.log 2>&1 || echo "mix pass failed" >> $O/notes.txt
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
for d in ('sq','l2','fetch','write','mix'):
    out.update(per_launch(d, lambda n: 'render_grid' in n))
for d in ('fetch','write'):
    for k,v in per_launch(d, lambda n: 'grid_finalize' in n).items(): out['grid_finalize.'+k]=v
stats=[r for f in glob.glob('$O/trace/*/*_kernel_stats.csv') for r in csv.DictReader(open(f))]
out['kernel_stats']=[r for r in stats if 'render' in r['Name'] or 'grid' in r['Name']]
clk=out.get('GRBM_GUI_ACTIVE',0)
try:
    ns=float([r for r in out['kernel_stats'] if 'render_grid' in r['Name']][0]['AverageNs'])
    out['valu_busy_at_2.4GHz']=out['SQ_ACTIVE_INST_VALU']*4/1024/(ns*2.4)  # (GRBM_GUI_ACTIVE comes out summed over the 8 XCDs here)
    out['active_lanes_frac']=out['SQ_THREAD_CYCLES_VALU']/(out['SQ_ACTIVE_INST_VALU']*64)
    out['wave_wait_any_share']=out['SQ_WAIT_ANY']/out['SQ_WAVE_CYCLES']
    out['wave_wait_inst_share']=out['SQ_WAIT_INST_ANY']/out['SQ_WAVE_CYCLES']
    out['l2_hit_rate']=out['TCC_HIT_sum']/(out['TCC_HIT_sum']+out['TCC_MISS_sum'])
except Exception as e:
    out['derived_error']=repr(e)
out['_note']='per launch, C5 full frame, render_grid_kernel (variant 17) and, prefixed, grid_finalize_kernel; FETCH_SIZE/WRITE_SIZE in KB as the counters report them'
json.dump(out,open('$O/summary.json','w'),indent=1); print(json.dumps(out,indent=1))
PY
