#!/bin/bash
# round-3 batch 3: where does the Hessenberg rounding error come from (accumulation order of the GEMMs)?
mkdir -p gpurun_out
export STARNEIG_AMD_TUNING=1
L=gpurun_out/r3_run3.log
for cfg in "" "SN_GEMM_SEPSUM=1" "SN_GEMM_SEPSUM=1 SN_GEMM_KCHUNK=256" "SN_GEMM_SEPSUM=1 SN_GEMM_KCHUNK=128"; do
  for n in 4000 8000; do
    echo "== $cfg n=$n" >> $L
    env $cfg timeout 300 python scratch/acc_diag.py $n 2>&1 | grep "hessenberg" >> $L
  done
done
echo "== host api n=8000" >> $L
timeout 300 python - >> $L 2>&1 <<'P'
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch, starneig_amd as S, oracle as O
torch.cuda.set_device(0); torch.zeros(1, device='cuda')
S.node_init(-1, 1, S.NO_MESSAGES)
n = 8000
A0 = O.random_fullpos(n); A = A0.copy(order='F'); Q = O.identity(n)
for rep in range(2):
    A[:] = A0; Q[:] = O.identity(n)
    t = time.time(); rc = S.SEP_SM_Hessenberg(n, A, A.shape[0], Q, Q.shape[0]); t1 = time.time() - t
    print('host-api hessenberg n=%d rc=%d %.3f s' % (n, rc, t1), flush=True)
tA = S.device_matrix(n); S.lcg_fill_device(tA, n, n); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
torch.cuda.synchronize(); t = time.time(); S.hessenberg_device(tA, tQ, n=n); torch.cuda.synchronize()
print('device-resident %.3f s' % (time.time() - t))
H = np.asfortranarray(tA.cpu().numpy().T)
print('max |A_host - A_dev| =', np.abs(H[:n] - A[:n]).max(), 'res', O.residual_u(Q, A, A0), 'orth', O.orthogonality_u(Q))
P
tail -30 $L
