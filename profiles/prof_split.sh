# PMC + timing of a SPLIT launch: rows 0:136 of the headline frame (4080 tiles, what one rank of eight gets in tile count)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_split
rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --cpu-rows 0 --no-extras --rows 0:136"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 2 --warmup 0 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 2 --warmup 0 > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 5 --warmup 1 > $O/trace.log 2>&1
python3 - <<PY
import csv,glob,collections,json
out={}
for d in ('fetch','write'):
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        for name in ('render_tiles','split_finalize'):
            rows=[r for r in csv.DictReader(open(f)) if name in r['Kernel_Name']]
            disp=len(set(r['Dispatch_Id'] for r in rows))
            agg=collections.defaultdict(float)
            for r in rows: agg[r['Counter_Name']]+=float(r['Counter_Value'])
            for k,v in agg.items(): out[name+'.'+k+'_KB_per_launch']=v/max(1,disp)
out['kernel_stats']=[{k:r[k] for k in ('Name','Calls','AverageNs')} for f in glob.glob('$O/trace/*/*_kernel_stats.csv') for r in csv.DictReader(open(f)) if 'render' in r['Name'] or 'split' in r['Name']]
json.dump(out,open('$O/summary.json','w'),indent=1); print(json.dumps(out,indent=1))
PY
