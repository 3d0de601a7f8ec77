# PMC view of the large-scene nearest-hit kernel: a 256-row strip of BASELINE configs[4] (rocprofv3, program after --)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_c5
rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --workload c5 --rows 400:656 --steps 1 --warmup 0 --cpu-rows 0 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/mix -- $B > $O/mix.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/mix2 -- $B > $O/mix2.log 2>&1 || echo "mix2 failed" >> $O/notes.txt
python3 - <<PY
import csv,glob,collections,json
out={}
for d in ('mix','mix2'):
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k='nearest_f32' if 'wf_nearest_f32' in r['Kernel_Name'] else 'shade' if 'wf_shade' in r['Kernel_Name'] else None
            if k: agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        for k,v in agg.items(): out.setdefault(k,{}).update(v)
stats=[r for f in glob.glob('$O/trace/*/*_kernel_stats.csv') for r in csv.DictReader(open(f))]
out['kernel_stats']=[{k:r[k] for k in ('Name','Calls','TotalDurationNs','AverageNs','Percentage')} for r in stats if 'wf_' in r['Name']]
out['_note']='totals over ALL launches of one render of rows 400:656 of BASELINE configs[4] (100k spheres, 1080p, 256 spp)'
json.dump(out,open('$O/summary.json','w'),indent=1); print(json.dumps(out,indent=1))
PY
