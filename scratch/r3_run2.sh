#!/bin/bash
# round-3 experiment batch 2: CU mask for the lazy Schur streams, split-K chunk vs Hessenberg accuracy
mkdir -p gpurun_out
export STARNEIG_AMD_TUNING=1 SN_SCHUR_PROFILE=1
for m in 0 32 64 96; do
  echo "== CUMASK $m" >> gpurun_out/r3_run2.log
  SN_SCHUR_CUMASK=$m timeout 300 python scratch/schur_configs.py 20000 160,106,-1 160,106,-1 >> gpurun_out/r3_run2.log 2>&1
done
for k in 0 256 512 1024 2048; do
  echo "== KCHUNK $k" >> gpurun_out/r3_run2.log
  SN_GEMM_KCHUNK=$k timeout 300 python scratch/acc_diag.py 4000 2>&1 | grep -v "per-column" >> gpurun_out/r3_run2.log
done
timeout 900 python -m pytest tests/test_gpu_reorder.py tests/test_gpu_hessenberg.py -x -q -m gpu > gpurun_out/r3_run2_tests.log 2>&1
tail -3 gpurun_out/r3_run2_tests.log >> gpurun_out/r3_run2.log
