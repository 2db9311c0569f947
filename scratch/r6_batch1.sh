#!/bin/bash
# round 6, first GPU batch: smoke, the tests of everything touched so far, the default bench line, HT timings
cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r6_smoke.log
tail -2 gpurun_out/r6_smoke.log
timeout 2400 python -m pytest tests -m gpu -q -x --durations=10 -k "hessenberg or gemm or c_caller or ht or node_team or distributed" > gpurun_out/r6_tests1.log 2>&1
tail -18 gpurun_out/r6_tests1.log
( time timeout 900 python bench.py ) > gpurun_out/r6_bench1.json 2> gpurun_out/r6_bench1.err
tail -4 gpurun_out/r6_bench1.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_bench1.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline_mfma"]["frac"], d["roofline_mfma"]["critical_update_frac"])
print({k: v for k, v in d["config"].items() if k.endswith("_s")})
print(json.dumps(d.get("secondary"))[:1500])
PY
