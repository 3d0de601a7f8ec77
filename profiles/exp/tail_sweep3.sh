# short launches of 1.5-3 rounds: all-split (round 1's rule) vs tail-only with several tails
run() { RTM_DEBUG_LONG=$1 RTM_DEBUG_TAIL=$2 python3 profiles/exp/tail_one.py $3 2>&1 | grep tail= | sed "s/^/LONG=$1 /"; }
for f in 1920x200 1920x320 1920x408 1920x680; do
  run 6 0 $f
  for t in 1024 1536 2048; do run 1 $t $f; done
done
run 1 1904 1920x200
run 1 5504 1920x320
run 1 4048 1920x408
