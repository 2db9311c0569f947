#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for r in 1 2; do for n in 8000 12000; do timeout 600 python scratch/r5_ht2.py $n 2>&1 | grep "n="; done; done | tee gpurun_out/r6_ht_rep.txt
