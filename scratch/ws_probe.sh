#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_schur.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python scratch/queue_probe.py plain 2>&1 | grep "^plain"
timeout 300 python scratch/queue_probe.py plain 2>&1 | grep "^plain"
