# PMC of the C5 nearest kernels: two rays per lane (default) against two spheres per instruction (RTM_DEBUG_WF_X1=1)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in x2 x1; do
  O=$R/gpurun_out/prof_c5_$tag; rm -rf $O; mkdir -p $O
  if [ $tag = x1 ]; then export RTM_DEBUG_WF_X1=1; else unset RTM_DEBUG_WF_X1; fi
  B="python3 $R/bench.py --workload c5 --rows 508:572 --steps 1 --warmup 0 --cpu-rows 0 --no-extras"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/mix -- $B > $O/mix.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/mix2 -- $B > $O/mix2.log 2>&1
  python3 - <<PY
import csv,glob,collections,json
out={}
for d in ('mix','mix2'):
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'wf_nearest_f32' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
        out.update(agg)
json.dump(out,open('$O/summary.json','w'),indent=1); print('$tag',json.dumps(out))
PY
done
