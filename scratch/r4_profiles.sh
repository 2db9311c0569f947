#!/bin/bash
# Round-4 profiles: run on the GPU box (gpurun), results land in gpurun_out/r4_profiles/ and are then
# copied into profiles/ by hand.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4_profiles
mkdir -p $O /tmp/pm
# 1. kernel trace of the bench command (same flags as the driver's line, two steps)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 > $O/bench_under_rocprof.log 2>&1
cp $(ls /tmp/p_bench/*/*kernel_stats.csv | head -1) $O/hess_schur_n20000_kernel_stats.csv
python3 scratch/kstats.py /tmp/p_bench 25 > $O/hess_schur_n20000_summary.txt 2>&1
tail -1 $O/bench_under_rocprof.log >> $O/hess_schur_n20000_summary.txt
rm -rf /tmp/p_bench
# 2. MFMA busy over the whole Hessenberg reduction (65 panels)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma1 -- python3 $R/scratch/pmc_run.py > $O/pmc_insitu1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma2 -- python3 $R/scratch/pmc_run.py > $O/pmc_insitu2.log 2>&1
# 3. gemv traffic, first two panels
export STARNEIG_AMD_TUNING=1 SN_HESS_MAX_PANELS=2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm/pmc_fetch -- python3 $R/scratch/pmc_run.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pm/pmc_write -- python3 $R/scratch/pmc_run.py > $O/pmc_write.log 2>&1
unset STARNEIG_AMD_TUNING SN_HESS_MAX_PANELS
# 4. the GEMM alone
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm/alone_mfma1 -- python3 $R/scratch/gemm_bench.py > $O/gemm_bench.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm/alone_mfma2 -- python3 $R/scratch/gemm_bench.py > $O/gemm_bench2.log 2>&1
python3 scratch/r4_pmc_summarise.py /tmp/pm > $O/pmc_summary.json 2> $O/pmc_summary.err
# 5. Hessenberg-triangular reduction, n = 4000
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ht -- python3 $R/scratch/ht_time.py 4000 > $O/ht_n4000.log 2>&1
cp $(ls /tmp/p_ht/*/*kernel_stats.csv | head -1) $O/ht_n4000_kernel_stats.csv
python3 scratch/kstats.py /tmp/p_ht 14 > $O/ht_summary.txt 2>&1
grep "^n=" $O/ht_n4000.log >> $O/ht_summary.txt
rm -rf /tmp/p_ht /tmp/pm
head -30 $O/hess_schur_n20000_summary.txt; head -c 2500 $O/pmc_summary.json; cat $O/ht_summary.txt
