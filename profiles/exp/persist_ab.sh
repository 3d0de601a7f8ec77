# same box: tail split without the persistent-lane code (compile-time off) against persistent waves of 2x / 1x / 4x the resident waves
run() { RTM_LIB_OVERRIDE=$1 RTM_DEBUG_PERSIST=$2 RTM_DEBUG_TAIL=$3 timeout -k 10 100 python3 profiles/exp/tail_one.py ${4:-1920x1080} 2>&1 | grep tail= | sed "s/^/$(basename ${1:-tree}) persist=${2:-default} /"; }
for rep in 1 2; do
run ab_libs/librtm_nopersist.so "" 1536
run "" 8192 1536
run "" 16384 1536
run "" 8192 2560
run "" 6144 1536
done
run ab_libs/librtm_nopersist.so "" 1536 1920x544
run "" 8192 1536 1920x544
run ab_libs/librtm_nopersist.so "" 1536 1920x272
run "" 8192 1536 1920x272
