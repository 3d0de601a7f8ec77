run() { RTM_DEBUG_SPLIT=$1 RTM_DEBUG_TAIL=$2 timeout -k 10 100 python3 profiles/exp/tail_one.py $3 2>&1 | grep tail= | sed "s/^/SPLIT=${1:-rule} /"; }
for f in 1920x136 1920x272; do
run "" "" $f
run 2 "" $f
run 4 "" $f
run 8 "" $f
run 1 "" $f
done
