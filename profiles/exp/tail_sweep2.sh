# medium launches: all tiles split (RTM_DEBUG_LONG=6, round 1's rule) against tail-only split (RTM_DEBUG_LONG=1) and tail sizes
for f in 1920x544 960x540 1920x272 1920x136; do
  RTM_DEBUG_LONG=6 python3 profiles/exp/tail_one.py $f 2>&1 | grep tail= | sed 's/^/all-split  /'
  for t in 1024 2048 3072; do
    RTM_DEBUG_LONG=1 RTM_DEBUG_TAIL=$t python3 profiles/exp/tail_one.py $f 2>&1 | grep tail= | sed 's/^/tail-split /'
  done
done
for t in 1536 2048 2560; do RTM_DEBUG_TAIL=$t python3 profiles/exp/tail_one.py 1920x1080 2>&1 | grep tail=; done
python3 profiles/exp/tail_one.py 3840x2160 2>&1 | grep tail=
RTM_DEBUG_TAIL=0 python3 profiles/exp/tail_one.py 3840x2160 2>&1 | grep tail=
