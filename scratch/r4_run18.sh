#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for w in 24 32 40 48; do
  STARNEIG_AMD_TUNING=1 SN_GEP_WINDOW=$w timeout 300 python bench.py --workload qz --steps 2 --warmup 1 --cpu-n 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config']
print('gep window $w:', d.get('ms_per_step'), c.get('aeds'), c.get('qz_sweeps'), c.get('aed_host_s'), c.get('residual_a_u'), c.get('residual_b_u'))
"
  STARNEIG_AMD_TUNING=1 SN_GEP_WINDOW=$w timeout 300 python bench.py --workload qz --pencil wellcond --steps 2 --warmup 1 --cpu-n 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config']
print('   wellcond $w:', d.get('ms_per_step'), c.get('aeds'), c.get('qz_sweeps'), c.get('aed_host_s'))
"
done
python - <<'PY'
import os, time
os.environ["OMP_NUM_THREADS"]="32"
import oracle as O, starneig_amd as S
L = S.lib.load_test_hooks()
n=2000
A0 = O.random_fullpos(n); H = A0.copy(order="F"); Q = O.identity(n)
t=time.time(); O.hessenberg(H, Q); t1=time.time()-t
t=time.time(); rc, wr, wi, st = O.msqr_port(H, Q, L.sn_internal_aed_window, L.sn_internal_small_schur); t2=time.time()-t
print("port n=2000 32 threads: hess", t1, "schur", t2, st)
PY
