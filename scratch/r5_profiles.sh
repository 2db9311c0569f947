#!/bin/bash
# Round-5 profiles: run on the GPU box (gpurun), results land in gpurun_out/r5_profiles/ and are then
# copied into profiles/ by hand.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r5_profiles
mkdir -p $O /tmp/pm
# 1. kernel trace of the bench command (same flags as the driver's line, two steps)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 > $O/bench_under_rocprof.log 2>&1
cp $(ls /tmp/p_bench/*/*kernel_stats.csv | head -1) $O/hess_schur_n20000_kernel_stats.csv
python3 scratch/kstats.py /tmp/p_bench 25 > $O/hess_schur_n20000_summary.txt 2>&1
tail -1 $O/bench_under_rocprof.log >> $O/hess_schur_n20000_summary.txt
rm -rf /tmp/p_bench
# 2. gemv traffic, first two panels (separate passes, FETCH_SIZE / WRITE_SIZE)
export STARNEIG_AMD_TUNING=1 SN_HESS_MAX_PANELS=2
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm/pmc_fetch -- python3 $R/scratch/pmc_run.py > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pm/pmc_write -- python3 $R/scratch/pmc_run.py > $O/pmc_write.log 2>&1
unset STARNEIG_AMD_TUNING SN_HESS_MAX_PANELS
python3 scratch/r4_pmc_summarise.py /tmp/pm > $O/pmc_summary.json 2> $O/pmc_summary.err
rm -rf /tmp/pm
head -30 $O/hess_schur_n20000_summary.txt; head -c 1500 $O/pmc_summary.json
