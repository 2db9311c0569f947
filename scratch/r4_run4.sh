#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x 2>&1 | tail -3
echo "== split"; timeout 200 python scratch/gemm_tail_bench.py 2>&1 | grep -v amdgpu.ids
echo "== nosplit"; STARNEIG_AMD_TUNING=1 SN_GEMM_NOSPLIT=1 timeout 200 python scratch/gemm_tail_bench.py 2>&1 | grep -v amdgpu.ids
