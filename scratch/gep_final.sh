#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_gpu_gep.py tests/test_gpu_testdriver.py tests/test_gpu_baseline_configs.py tests/test_gpu_ht.py -m gpu -q -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
python bench.py --workload secondary 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
d = d if isinstance(d, list) else d.get('secondary', d)
for e in d: print({k: (round(v,3) if isinstance(v,float) else v) for k,v in e.items() if k in ('pencil','n','hessenberg_triangular_s','qz_s','seconds_per_step','qz_sweeps','aeds','residual_a_u','residual_b_u')})
"
