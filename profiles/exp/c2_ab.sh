for l in base noforce; do RTM_LIB_OVERRIDE=ab_libs/librtm_$l.so python3 profiles/exp/c2_tail.py 2>&1 | grep C2 | sed "s/^/$l /"; done
for l in base noforce; do RTM_LIB_OVERRIDE=ab_libs/librtm_$l.so python3 profiles/exp/c2_tail.py 2>&1 | grep C2 | sed "s/^/$l /"; done
