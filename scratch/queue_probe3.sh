#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for mode in 0 1; do
  for pad in 3 4 5 6 7 8; do
    echo -n "mode $mode pad $pad  "
    SN_STREAM_MODE=$mode SN_STREAM_PAD=$pad timeout 300 python scratch/queue_probe.py plain 2>&1 | grep "^plain"
  done
done
echo "profile of a slow one (mode 0 pad 1):"
SN_STREAM_MODE=0 SN_STREAM_PAD=1 SN_SCHUR_PROFILE=1 timeout 300 python scratch/queue_probe.py plain 2>&1 | grep -v amdgpu.ids | tail -8
echo "profile of mode 0 pad 0:"
SN_STREAM_MODE=0 SN_STREAM_PAD=0 SN_SCHUR_PROFILE=1 timeout 300 python scratch/queue_probe.py plain 2>&1 | grep -v amdgpu.ids | tail -8
