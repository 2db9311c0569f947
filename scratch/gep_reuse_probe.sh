#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for r in 4 8; do
  echo -n "SN_GEP_REUSE=$r  "
  SN_GEP_REUSE=$r timeout 600 python scratch/gep_chain.py 8000 2>&1 | grep "^n="
done
for r in 2 4 8; do
  echo -n "SN_GEP_REUSE=$r n=3000  "
  SN_GEP_REUSE=$r timeout 600 python scratch/gep_chain.py 3000 2>&1 | grep "^n="
done
for r in 1 4; do
  echo -n "SN_GEP_REUSE=$r n=1500  "
  SN_GEP_REUSE=$r timeout 600 python scratch/gep_chain.py 1500 2>&1 | grep "^n="
done
echo "GEP tests with SN_GEP_REUSE=4:"
SN_GEP_REUSE=4 timeout 900 python -m pytest tests/test_gpu_gep.py tests/test_gpu_testdriver.py tests/test_gpu_baseline_configs.py -m gpu -q -x -k "gep or qz or generalized or config5 or pencil" 2>&1 | tail -3
