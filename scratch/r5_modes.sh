#!/bin/bash
# the round-4 recipe that produced the wrong sharded result (the two multi-rank
# test files in one process, dedicated hardware queues), once per fold variant of the sharded gemv
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1
for cell in "1 1" "1 0" "1 2" "0 0"; do
  set -- $cell
  echo "== SN_STREAM_MODE=$1 SN_HESS_FOLD=$2"
  SN_STREAM_MODE=$1 SN_HESS_FOLD=$2 timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_node_team.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|assert|Error" | head -30
done
