#!/bin/bash
# Round-3 profile set (run on the GPU box): kernel stats of the bench command, PMC traffic of the
# panel gemv, PMC MFMA utilisation of the GEMM updates in situ (first two panels of n = 20000, side
# stream on and off) and standalone, and of the aggregated Schur update kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof; mkdir -p $O
export STARNEIG_AMD_TUNING=1
# 1. kernel stats of the bench step (one warm-up + one timed reduction)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 > $O/bench_under_rocprof.log 2>&1
cp $(ls /tmp/p_bench/*/*kernel_stats.csv | head -1) $O/hess_schur_n20000_kernel_stats.csv
python3 $R/scratch/kstats.py /tmp/p_bench 24 > $O/hess_schur_n20000_summary.txt 2>&1
# 2. gemv traffic (two panels) and in-situ MFMA busy of the GEMM kernels
export SN_HESS_MAX_PANELS=2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm/pmc_fetch -- python3 $R/scratch/pmc_run.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pm/pmc_write -- python3 $R/scratch/pmc_run.py > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma1 -- python3 $R/scratch/pmc_run.py > $O/pmc_insitu1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma2 -- python3 $R/scratch/pmc_run.py > $O/pmc_insitu2.log 2>&1
export SN_HESS_NOSIDE=1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm/noside_mfma1 -- python3 $R/scratch/pmc_run.py > $O/pmc_noside1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm/noside_mfma2 -- python3 $R/scratch/pmc_run.py > $O/pmc_noside2.log 2>&1
unset SN_HESS_NOSIDE SN_HESS_MAX_PANELS
# 3. standalone GEMM shapes and the aggregated Schur update kernels
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm/pmc_mfma1 -- python3 $R/scratch/gemm_bench.py > $O/gemm_bench.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm/pmc_mfma2 -- python3 $R/scratch/gemm_bench.py > $O/gemm_bench2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm/agg_mfma1 -- python3 $R/scratch/agg_bench.py > $O/agg_bench.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm/agg_mfma2 -- python3 $R/scratch/agg_bench.py > $O/agg_bench2.log 2>&1
python3 $R/scratch/pmc_summarise.py /tmp/pm > $O/pmc_summary.json 2> $O/pmc_summary.err
python3 $R/scratch/pmc_insitu.py /tmp/pm > $O/pmc_insitu_summary.json 2> $O/pmc_insitu.err
ls -la $O
tail -30 $O/pmc_insitu_summary.json
