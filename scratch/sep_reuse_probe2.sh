#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1 SN_CORES=64
for r in 0 4 6 8 0; do
  echo -n "n=20000 SN_SCHUR_REUSE=$r: "
  SN_SCHUR_REUSE=$r timeout 300 python scratch/schur_configs.py 20000 -1,-1,-1 2>&1 | grep "^-1" | cut -c1-200
done
