#!/bin/bash
# run on the GPU box: bash profiles/split_sweep.sh > gpurun_out/split_sweep.txt
for f in 1 2 4 8 0; do
  RTM_DEBUG_SPLIT=$f timeout -k 10 300 python profiles/split_sweep.py || exit 1
done
