#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export STARNEIG_AMD_TUNING=1
for n in 600 800 1000 1200 1500 2000; do
  for m in 0 1; do echo -n "SN_HT_TWOSTAGE=$m "; SN_HT_TWOSTAGE=$m timeout 300 python scratch/r5_ht2.py $n $n 2>&1 | grep "n=" | tail -1 | cut -c1-100; done
done | tee gpurun_out/r6_ht_crossover.txt
