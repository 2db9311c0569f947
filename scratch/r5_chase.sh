#!/bin/bash
cd "$(dirname "$0")/.."
echo "=== chase kernel alone (us per launch, 8 chains): 0 = U in registers, 1024 threads; 8 = U in LDS (rounds 1-4); 9 = U in registers, 512 threads"
timeout 200 python - <<'PY'
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
L = S.lib.load_test_hooks()
L.sn_internal_chase_bench.restype = C.c_double
L.sn_internal_chase_bench.argtypes = [C.c_int, C.c_int, C.c_int]
for chains in (1, 8, 29):
    print(chains, "chains:", {v: round(L.sn_internal_chase_bench(chains, 20, v), 1) for v in (0, 8, 9)}, flush=True)
PY
echo "=== Schur tests with 512 threads"
STARNEIG_AMD_TUNING=1 SN_SCHUR_CHASE_THREADS=512 timeout 600 python -m pytest tests/test_gpu_schur.py tests/test_gpu_baseline_configs.py -m gpu -q -x 2>&1 | tail -3
echo "=== bench: 1024 threads U in registers / 512 threads / U in LDS, twice each"
for rep in 1 2; do
for v in "A" "B SN_SCHUR_CHASE_THREADS=512" "C SN_SCHUR_CHASE_ULDS=1"; do
  set -- $v
  env STARNEIG_AMD_TUNING=1 $2 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-n 0 --cpu-port-n 0 --host-api 0 --secondary 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1 $2', round(d['ms_per_step']), round(c['hessenberg_s'],3), round(c['schur_s'],3), round(c['residual_u'],1), c['schur_sweeps'], c['schur_aeds'])"
done; done
