# same box: the headline frame with 0 .. 8192 of its last tiles sample-split (RTM_DEBUG_TAIL)
for t in 0 1024 2048 3072 4096 6144 8192 12288; do
  RTM_DEBUG_TAIL=$t python3 profiles/exp/tail_one.py 1920x1080 2>&1 | grep tail=
done
