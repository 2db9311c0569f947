#!/bin/bash
mkdir -p gpurun_out
export STARNEIG_AMD_TUNING=1 SN_GEMM_SEPSUM=1
L=gpurun_out/r3_run4.log
for p in 1 2 4 8; do
  echo "== MAX_PANELS $p" >> $L
  SN_HESS_MAX_PANELS=$p timeout 200 python scratch/hess_acc_probe.py 4000 2>&1 | grep "n=" >> $L
done
echo "== panel widths" >> $L
timeout 300 python scratch/hess_acc_probe.py 4000 32 64 128 280 480 2>&1 | grep "n=" >> $L
cat $L
