#!/bin/bash
cd "$(dirname "$0")/.."
export STARNEIG_AMD_TUNING=1 SN_SCHUR_PROFILE=1
timeout 3000 python -m pytest tests/test_gpu_testdriver.py -m gpu -q -x -s -k "generalized" --durations=25 2>&1 | grep -v "^$" | tail -80
timeout 1500 python -m pytest tests/test_gpu_hessenberg.py tests/test_gpu_gep.py -m gpu -q -s --durations=12 2>&1 | grep -E "panel width|passed|failed|FAILED|Error|assert|s call" | tail -40
