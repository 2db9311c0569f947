#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4_profiles
mkdir -p $O /tmp/pm
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex "dgemm" --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma1 -- python3 $R/scratch/pmc_run.py > $O/pmc_insitu1.log 2>&1
tail -2 $O/pmc_insitu1.log | cut -c1-200
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex "dgemm" --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma2 -- python3 $R/scratch/pmc_run.py > $O/pmc_insitu2.log 2>&1
tail -2 $O/pmc_insitu2.log | cut -c1-200
python3 scratch/r4_pmc_summarise.py /tmp/pm > $O/pmc_summary_insitu.json 2> $O/pmc_summary.err
head -c 3000 $O/pmc_summary_insitu.json
timeout 300 python -m pytest tests/test_gpu_hessenberg.py -m gpu -q -x -k "larger_than" 2>&1 | tail -2
