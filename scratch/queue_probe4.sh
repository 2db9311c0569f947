#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for free in 0 8 16 32 64; do
  echo -n "mode 3 lazy_free $free  "
  SN_STREAM_MODE=3 SN_STREAM_LAZY_FREE=$free timeout 300 python scratch/queue_probe.py plain 2>&1 | grep "^plain"
done
echo -n "mode 1            "; SN_STREAM_MODE=1 timeout 300 python scratch/queue_probe.py plain 2>&1 | grep "^plain"
echo -n "mode 3 free 32 pg "; SN_STREAM_MODE=3 SN_STREAM_LAZY_FREE=32 timeout 300 python scratch/queue_probe.py pg 2>&1 | grep "^pg"
echo -n "mode 3 free 32 norm3 "; SN_STREAM_MODE=3 SN_STREAM_LAZY_FREE=32 timeout 300 python scratch/queue_probe.py norm3 2>&1 | grep "^norm3"
