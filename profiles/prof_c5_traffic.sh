# HBM traffic of the large-scene pipeline over the FULL frame of BASELINE configs[4] (100k spheres, 1080p, 256 spp):
# FETCH_SIZE / WRITE_SIZE over every launch of two frames (bench.py's instrumented step + one timed step), separate passes.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_c5_traffic
rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --workload c5 --steps 1 --warmup 0 --cpu-rows 0 --no-extras $*"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
python3 - <<PY
import csv,glob,collections,json
out={'frames_profiled':2}
for d in ('fetch','write'):
    for f in glob.glob('$O/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(float); n=collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k='wf_nearest_f32' if 'wf_nearest_f32' in r['Kernel_Name'] else 'wf_shade' if 'wf_shade' in r['Kernel_Name'] else 'other'
            agg[k+'.'+r['Counter_Name']]+=float(r['Counter_Value']); n[k]+=1
        out.update(agg); out.setdefault('launches',{}).update(n)
out['_note']='KB over ALL launches of TWO full frames'
json.dump(out,open('$O/summary.json','w'),indent=1); print(json.dumps(out,indent=1))
PY
