#!/bin/bash
# round 6, end-of-round set: profiles (kernel statistics of the bench command, gemv PMC traffic over two panels, one
# attempt at the in-situ MFMA counters over 4 panels), the bench lines (default, ht), smoke and the whole GPU suite
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6_profiles
mkdir -p $O /tmp/pm
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r6_smoke.log
tail -2 gpurun_out/r6_smoke.log
# 1. the driver-style line
( time timeout 900 python bench.py ) > gpurun_out/r6_bench_line.json 2> gpurun_out/r6_bench_line.err
tail -4 gpurun_out/r6_bench_line.err
# 2. kernel trace of the bench command (two steps)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 > $R/$O/bench_under_rocprof.log 2>&1
cd $R
cp $(ls /tmp/p_bench/*/*kernel_stats.csv | head -1) $O/hess_schur_n20000_kernel_stats.csv
python3 scratch/kstats.py /tmp/p_bench 25 > $O/hess_schur_n20000_summary.txt 2>&1
tail -1 $O/bench_under_rocprof.log >> $O/hess_schur_n20000_summary.txt
rm -rf /tmp/p_bench
# 3. gemv traffic, first two panels (separate passes, FETCH_SIZE / WRITE_SIZE)
export STARNEIG_AMD_TUNING=1 SN_HESS_MAX_PANELS=2
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm/pmc_fetch -- python3 $R/scratch/pmc_run.py > $R/$O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pm/pmc_write -- python3 $R/scratch/pmc_run.py > $R/$O/pmc_write.log 2>&1
# 4. MFMA busy over the first four panels (the whole reduction crashed the profiler in round 5)
export SN_HESS_MAX_PANELS=4
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma1 -- python3 $R/scratch/pmc_run.py > $R/$O/pmc_mfma1.log 2>&1; echo "mfma1 rc=$?"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm/insitu_mfma2 -- python3 $R/scratch/pmc_run.py > $R/$O/pmc_mfma2.log 2>&1; echo "mfma2 rc=$?"
unset STARNEIG_AMD_TUNING SN_HESS_MAX_PANELS
cd $R
python3 scratch/r4_pmc_summarise.py /tmp/pm > $O/pmc_summary.json 2> $O/pmc_summary.err
rm -rf /tmp/pm
head -12 $O/hess_schur_n20000_summary.txt; head -c 2500 $O/pmc_summary.json
# 5. the Hessenberg-triangular line
( time timeout 900 python bench.py --workload ht --steps 1 --warmup 1 ) > gpurun_out/r6_bench_ht_line.json 2> gpurun_out/r6_bench_ht_line.err
tail -c 1500 gpurun_out/r6_bench_ht_line.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_bench_line.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline_mfma"]["frac"], d["roofline_mfma"]["critical_update_frac"])
print({k: v for k, v in d["config"].items() if k.endswith("_s")})
print(json.dumps(d.get("secondary"))[:1800])
PY
# 6. the whole GPU suite
timeout 2700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r6_gpu_tests.log 2>&1
tail -6 gpurun_out/r6_gpu_tests.log
