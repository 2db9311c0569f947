#!/bin/bash
# creation order in a plain process: hess main(0) side(1) | schur own(2) far(3) qs(4) hs(5) aed(6)
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1 SN_STREAM_MODE=1
for sp in 0000000 0010000 0020000 0030000 0001000 0002000 0000001 0000002 0000003 1000000 2000000 0011001; do
  echo -n "space $sp  "
  SN_STREAM_SPACE=$sp timeout 300 python scratch/queue_probe.py plain 2>&1 | grep "^plain"
done
