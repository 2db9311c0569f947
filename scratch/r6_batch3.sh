#!/bin/bash
# round 6, third batch: GEMM priority experiment, Schur step anatomy, CTest decouple timings, PMC retries
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== gemm, default"; python scratch/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_gemm_default.txt
echo "== gemm, SN_GEMM_PRIO=1"; STARNEIG_AMD_TUNING=1 SN_GEMM_PRIO=1 python scratch/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_gemm_prio.txt
echo "== hess with prio"; STARNEIG_AMD_TUNING=1 SN_GEMM_PRIO=1 python scratch/hess_only.py 20000 2 2>&1 | grep -v amdgpu.ids | tail -2
echo "== hess default"; python scratch/hess_only.py 20000 2 2>&1 | grep -v amdgpu.ids | tail -2
# Schur step anatomy
cd /tmp; rm -rf /tmp/p_s
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_s -- python3 $R/scratch/schur_only.py 20000 > /tmp/s.log 2>&1
cd $R
tail -1 /tmp/s.log | cut -c1-300
python3 scratch/step_timeline.py /tmp/p_s > gpurun_out/r6_schur_step_anatomy.txt 2>&1; tail -25 gpurun_out/r6_schur_step_anatomy.txt
# decouple vs plain
timeout 900 python -m pytest tests/test_gpu_testdriver.py -m gpu -q -s -k "ctest_schur_standard and (default or aed-50-)" 2>&1 | grep -E "aed=|passed|failed" | tee gpurun_out/r6_ctest_decouple_timing.txt
# PMC retry, filtered to the two chase kernels
rm -rf /tmp/pmh; mkdir -p /tmp/pmh; cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "ht2_apply_right|ht2_geng_left" --output-format csv -d /tmp/pmh/fetch -- python3 $R/scratch/r5_ht2.py 2500 > $R/gpurun_out/r6_ht2_pmc_fetch2.log 2>&1
echo "pmc fetch rc=$?"; ls /tmp/pmh/fetch/*/ 2>/dev/null | head
grep -v "^    @" $R/gpurun_out/r6_ht2_pmc_fetch2.log | tail -4 | cut -c1-200
