#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4_profiles
mkdir -p $O /tmp/pm
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma1 -- python3 $R/scratch/pmc_run2.py 4000 > $O/pmc_small.log 2>&1
echo "n=4000 rc $?"; tail -3 $O/pmc_small.log | cut -c1-200
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma1 -- python3 $R/scratch/pmc_run2.py > $O/pmc_insitu1.log 2>&1
rc1=$?
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma2 -- python3 $R/scratch/pmc_run2.py > $O/pmc_insitu2.log 2>&1
rc2=$?
echo "full: rc $rc1 $rc2"
if [ $rc1 -eq 0 ] && [ $rc2 -eq 0 ]; then
  python3 scratch/r4_pmc_summarise.py /tmp/pm > $O/pmc_summary_insitu.json 2> $O/pmc_summary.err
  head -c 3000 $O/pmc_summary_insitu.json
fi
for i in 1 2; do
timeout 900 python bench.py --secondary 0 --cpu-n 0 --cpu-port-n 0 --host-api 0 --steps 3 > gpurun_out/r4_bench_line5.json 2> gpurun_out/r4_bench_err5.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench_line5.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"]["hessenberg_s"], d["config"]["schur_s"], d["roofline"]["frac"], d["roofline_mfma"]["frac"], d["roofline_mfma"]["critical_update_frac"], d["config"]["residual_u"])
PY
done
