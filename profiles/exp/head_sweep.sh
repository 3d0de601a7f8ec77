# the tail's shape: granularity g (RTM_DEBUG_SPLIT forces every tile; so use the share of 1536 via frames) -- here: head fraction of the split tiles
run() { RTM_DEBUG_HEAD=$1 RTM_DEBUG_TAIL=$2 timeout -k 10 100 python3 profiles/exp/tail_one.py ${3:-1920x1080} 2>&1 | grep tail= | sed "s/^/head=$1\/16 /"; }
for h in 4 8 12 14; do run $h 1536; done
for h in 4 8 12 14; do run $h 1536 512x512; done
for h in 4 8 12; do run $h 2048; done
