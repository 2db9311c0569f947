#!/bin/bash
cd "$(dirname "$0")/.."
echo "=== team tests, default mode (device exchange on virtual ranks)"
timeout 300 python -m pytest tests/test_gpu_node_team.py -m gpu -q -x -s 2>&1 | grep -E "elementwise|passed|failed|FAILED|starneig-amd|Error|rror" | cut -c1-300 | head -20
echo "=== preemption experiment (single-GPU path, queues created and destroyed meanwhile)"
timeout 300 python scratch/r5_preempt.py 20 1500 8 2>&1 | tail -5
echo "=== team timing: host exchange against device exchange"
timeout 400 python scratch/r5_team_time.py 8000 1 2 4 2>&1 | tail -8
