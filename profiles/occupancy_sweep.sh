# kernel time vs resident waves per SIMD (LDS padding caps occupancy): latency- or issue-bound?
for pad in 0 12000 19000 39000 79000; do
  echo "pad=$pad"; RTM_DEBUG_LDS_PAD=$pad python bench.py --ab 2 --steps 2 --warmup 1 2>/dev/null | tail -1
done
