#!/bin/bash
# round-5 validation: smoke, the whole GPU suite, the default bench line
cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r5_smoke.log
tail -2 gpurun_out/r5_smoke.log
timeout 2700 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r5_gpu_tests.log 2>&1
tail -22 gpurun_out/r5_gpu_tests.log
( time timeout 900 python bench.py ) > gpurun_out/r5_bench_line.json 2> gpurun_out/r5_bench_line.err
tail -4 gpurun_out/r5_bench_line.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_bench_line.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline_mfma"]["frac"], d["roofline_mfma"]["critical_update_frac"])
print(json.dumps(d["cpu_baseline"])[:900])
print(json.dumps(d.get("secondary"))[:1500])
PY
