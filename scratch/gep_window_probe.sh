#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for w in 64 80 96 128; do
  echo -n "SN_GEP_WINDOW=$w  "
  SN_GEP_WINDOW=$w timeout 600 python scratch/gep_chain.py 8000 2>&1 | grep "^n=" | cut -c1-260
done
