#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_hessenberg.py tests/test_gpu_ht.py tests/test_gpu_distributed.py -m gpu -q -x 2>&1 | tail -4
echo "== gemm tail"; timeout 200 python scratch/gemm_tail_bench.py 2>&1 | grep -v amdgpu.ids
echo "== HT"; for n in 4000 8000; do timeout 300 python scratch/ht_time.py $n 2>&1 | grep -v amdgpu.ids; done
timeout 900 python bench.py --secondary 0 > gpurun_out/r4_bench_line2.json 2> gpurun_out/r4_bench_err2.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench_line2.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"]["hessenberg_s"], d["config"]["schur_s"], d["roofline"]["frac"], d["roofline_mfma"]["frac"], d["roofline_mfma"]["critical_update_frac"], d["config"]["residual_u"])
PY
