#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_hessenberg.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do
timeout 900 python bench.py --secondary 0 --cpu-n 0 --cpu-port-n 0 --host-api 0 --steps 3 > gpurun_out/r4_bench_line3.json 2> gpurun_out/r4_bench_err3.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench_line3.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"]["hessenberg_s"], d["config"]["schur_s"], d["roofline"]["frac"], d["roofline_mfma"]["frac"], d["roofline_mfma"]["critical_update_frac"], d["config"]["residual_u"])
PY
done
