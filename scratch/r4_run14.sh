#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hessenberg.py tests/test_gpu_distributed.py tests/test_gpu_node_team.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
timeout 900 python bench.py --secondary 0 --cpu-n 0 --cpu-port-n 0 --host-api 0 --steps 3 > gpurun_out/r4_bench_line4.json 2> gpurun_out/r4_bench_err4.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench_line4.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"]["hessenberg_s"], d["config"]["schur_s"], d["roofline"]["frac"], d["roofline_mfma"]["frac"], d["roofline_mfma"]["critical_update_frac"], d["config"]["residual_u"])
PY
done
O=gpurun_out/r4_profiles
mkdir -p $O /tmp/pm
export STARNEIG_AMD_TUNING=1
for P in 65 16; do
  export SN_HESS_MAX_PANELS=$P
  rm -rf /tmp/pm/insitu_mfma1 /tmp/pm/insitu_mfma2
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma1 -- python3 $R/scratch/pmc_run.py > $O/pmc_insitu1.log 2>&1
  rc1=$?
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma2 -- python3 $R/scratch/pmc_run.py > $O/pmc_insitu2.log 2>&1
  rc2=$?
  echo "panels $P: rc $rc1 $rc2"
  if [ $rc1 -eq 0 ] && [ $rc2 -eq 0 ]; then
    python3 scratch/r4_pmc_summarise.py /tmp/pm > $O/pmc_summary_insitu_${P}panels.json 2> $O/pmc_summary.err
    head -c 2500 $O/pmc_summary_insitu_${P}panels.json
    break
  fi
done
