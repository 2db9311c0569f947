#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x 2>&1 | tail -3
echo "== gemm tail bench"; timeout 200 python scratch/gemm_tail_bench.py 2>&1 | grep -v amdgpu.ids
timeout 200 python scratch/gemm_bench.py 2>&1 | grep -v amdgpu.ids
echo "== node team"
timeout 600 python -X faulthandler -m pytest tests/test_gpu_node_team.py -m gpu -q -x > gpurun_out/r4_node_team.log 2>&1
head -c 5000 gpurun_out/r4_node_team.log
