# same-box, alternating processes, with in-wave sample stealing in place: number of split tiles of a launch (RTM_DEBUG_TAIL; default: choose_split)
# usage: tail_after_steal.sh "<tails>" ; launches: headline, shares of 2 / 4 / 8 GPUs (rows 0:540, 0:272, 0:136), 512x512x256spp, 1080p at 256 and 64 spp
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],3))"; }
for T in ${1:-"default 1536 2048 3072"}; do
  if [ "$T" = default ]; then unset RTM_DEBUG_TAIL; else export RTM_DEBUG_TAIL=$T; fi
  python bench.py --no-extras --cpu-rows 0 --steps 6 --warmup 1 2>/dev/null | one "headline(32400 tiles,1024spp) tail $T"
  python bench.py --rows 0:540 --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "rows 0:540 (16200 tiles)      tail $T"
  python bench.py --rows 0:272 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "rows 0:272 (8160 tiles)       tail $T"
  python bench.py --rows 0:136 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "rows 0:136 (4080 tiles)       tail $T"
  python bench.py --workload c2 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "c2 512x512 (4096 tiles,256spp) tail $T"
  python bench.py --samples 16 --no-extras --cpu-rows 0 --steps 10 --warmup 2 2>/dev/null | one "1080p x 256spp (32400 tiles)   tail $T"
  python bench.py --samples 4 --no-extras --cpu-rows 0 --steps 20 --warmup 2 2>/dev/null | one "1080p x 64spp (32400 tiles)    tail $T"
done
