#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1 SN_CORES=64
for n in 2000 4000 8000 12000; do
  for r in 0 2 4 8; do
    echo -n "n=$n SN_SCHUR_REUSE=$r: "
    SN_SCHUR_REUSE=$r timeout 300 python scratch/schur_configs.py $n -1,-1,-1 2>&1 | grep "^-1" | cut -c1-200
  done
done
