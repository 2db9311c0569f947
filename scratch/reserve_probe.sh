#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for k in 0 32 16 48 64 0; do
  echo -n "reserve $k  "
  SN_SCHUR_RESERVE=$k timeout 300 python scratch/queue_probe.py plain 2>&1 | grep "^plain"
done
