run() { RTM_LIB_OVERRIDE=$1 timeout -k 10 100 python3 profiles/exp/tail_one.py 1920x1080 2>&1 | grep tail= | sed "s/^/$(basename ${1:-tree}) /"; }
run ab_libs/librtm_nounit.so; run ""; run ab_libs/librtm_nounit.so; run ""
