#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1 SN_STREAM_MODE=0
for prio in 1 2; do
  for pad in 1 2 3 4 5; do
    echo -n "pad $pad of priority level $prio  "
    SN_STREAM_PAD=$pad SN_STREAM_PAD_PRIO=$prio timeout 300 python scratch/queue_probe.py plain 2>&1 | grep "^plain"
  done
done
