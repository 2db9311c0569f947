#!/bin/bash
# round 4, GPU session 3: GEMM after the wait-placement / boundary-tile changes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_hessenberg.py tests/test_gpu_testdriver.py -m gpu -q -x > gpurun_out/r4_s3_tests.log 2>&1
tail -5 gpurun_out/r4_s3_tests.log
timeout 300 python scratch/gemm_bench.py > gpurun_out/r4_s3_gemm_bench.log 2>&1
cat gpurun_out/r4_s3_gemm_bench.log
timeout 300 python scratch/known_gep_diag.py 4000 > gpurun_out/r4_s3_known_gep.log 2>&1
tail -5 gpurun_out/r4_s3_known_gep.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r4_s3_bench.log 2>&1
tail -3 gpurun_out/r4_s3_bench.log
