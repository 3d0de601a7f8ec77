run() { RTM_DEBUG_PERSIST=$1 RTM_DEBUG_TAIL=$2 timeout -k 10 100 python3 profiles/exp/tail_one.py 1920x1080 2>&1 | grep tail= | sed "s/^/persist=$1 /"; }
run 0 1536
run 16384 1536
run 8192 1536
run 4096 1536
run 4096 3728
run 3072 1536
run 2048 1536
