#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for mode in 3 1 3 1; do
  echo -n "setprio, mode $mode  "
  SN_STREAM_MODE=$mode timeout 300 python scratch/queue_probe.py plain 2>&1 | grep "^plain"
done
