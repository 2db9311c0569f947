#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ht.py tests/test_gpu_ht_twostage.py -m gpu -q -x 2>&1 | tail -3
for n in 4000 8000 12000; do timeout 600 python scratch/r5_ht2.py $n 2>&1 | grep "n="; done | tee gpurun_out/r6_ht_sizes_after_xcd.txt
