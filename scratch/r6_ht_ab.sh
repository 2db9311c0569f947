#!/bin/bash
# A/B of stage 2 on ONE box: the in-tree library against scratch/ab/libstarneig_amd_prev.so (the commit before)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp starneig_amd/libstarneig_amd.so /tmp/new.so
for r in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then cp scratch/ab/libstarneig_amd_prev.so starneig_amd/libstarneig_amd.so; else cp /tmp/new.so starneig_amd/libstarneig_amd.so; fi
    for n in 4000 8000 12000; do echo -n "$v "; timeout 600 python scratch/r5_ht2.py $n 2>&1 | grep "n=" | cut -c1-90; done
  done
done | tee gpurun_out/r6_ht_ab.txt
cp /tmp/new.so starneig_amd/libstarneig_amd.so
