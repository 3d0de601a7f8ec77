# wavefront nearest-kernel configurations vs the monolithic tiled kernel on a strip large enough to
# fill the chip (256 rows = 1920 workgroups)
python bench.py --workload c5 --rows 400:656 --steps 1 --warmup 0 --cpu-rows 0 --variant 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tiled monolithic', d['value'], d['ms_per_step'])"
for c in 0 1 2; do RTM_WF_CONFIG=$c python bench.py --workload c5 --rows 400:656 --steps 1 --warmup 0 --cpu-rows 0 --variant 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wavefront cfg $c', d['value'], d['ms_per_step'])"; done
