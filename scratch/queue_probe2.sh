#!/bin/bash
# stream mode (bit 0 critical / bit 1 lazy streams on queues of their own) x dummy high-priority streams created first
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for mode in 0 1 2 3; do
  for pad in 0 1 2; do
    echo -n "mode $mode pad $pad  "
    SN_STREAM_MODE=$mode SN_STREAM_PAD=$pad timeout 300 python scratch/queue_probe.py plain 2>&1 | grep "^plain"
  done
done
for mode in 0 1 3; do
  echo -n "mode $mode  "
  SN_STREAM_MODE=$mode timeout 300 python scratch/queue_probe.py pg 2>&1 | grep "^pg"
done
