#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for mode in 1 1 0; do
  echo "== mode $mode"
  SN_STREAM_MODE=$mode timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_node_team.py -m gpu -q -x 2>&1 | tail -3
done
