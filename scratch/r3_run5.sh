#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_run5.log; : > $L
timeout 600 python -m pytest tests/test_gpu_schur_agg.py tests/test_gpu_gemm.py -x -q 2>&1 | tail -5 >> $L
export STARNEIG_AMD_TUNING=1
for cfg in "" "SN_GEMM_SEPSUM=1"; do
  for n in 4000 8000; do
    echo "== acc $cfg n=$n" >> $L
    env $cfg timeout 300 python scratch/acc_diag.py $n 2>&1 | grep -E "hessenberg|schur alone|chain" >> $L
  done
done
for a in 0 1; do
  echo "== SCHUR AGG=$a" >> $L
  SN_SCHUR_AGG=$a SN_SCHUR_PROFILE=1 timeout 300 python scratch/schur_configs.py 20000 160,106,-1 160,106,-1 >> $L 2>&1
done
timeout 1200 python -m pytest tests/test_gpu_schur.py tests/test_gpu_hessenberg.py -x -q 2>&1 | tail -5 >> $L
cat $L
