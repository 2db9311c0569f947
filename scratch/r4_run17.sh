#!/bin/bash
cd "$GRAFT_REPO_ROOT"
( time python bench.py ) > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
tail -4 gpurun_out/r4_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline_mfma"]["frac"])
print(json.dumps(d["cpu_baseline"], indent=1))
print(d["config"].get("host_api_s"))
print([ (x.get("n"), x.get("hessenberg_triangular_s"), x.get("seconds_per_step")) for x in d["secondary"]])
PY
SN_BENCH_ONE_GPU=1 SN_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --size 6000 --steps 1 --warmup 0 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 2>&1 | tail -2 | cut -c1-600
for w in 48 56 80; do
  STARNEIG_AMD_TUNING=1 SN_GEP_WINDOW=$w timeout 300 python bench.py --workload qz --steps 1 --warmup 1 --cpu-n 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gep window $w:', d.get('ms_per_step'), d['config'].get('aeds'), d['config'].get('qz_sweeps'), d['config'].get('aed_host_s'))
"
done
