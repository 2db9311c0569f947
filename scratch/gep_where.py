import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import starneig_amd as S
S.node_init(1, 1, S.NO_MESSAGES)
torch.zeros(1, device='cuda')
n = int(sys.argv[1])
tH0, tR0 = S.device_matrix(n), S.device_matrix(n)
S.lcg_pencil_device(tH0, tR0, n)
tH, tR = tH0.clone(), tR0.clone()
tQ, tZ = S.device_matrix(n), S.device_matrix(n)
S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
rc, ar, ai, be, st = S.gep_schur_device(tH, tR, tQ, tZ, n=n)
torch.cuda.synchronize()
# tensors are (n, ld) = transposed column-major: X_colmajor = t[:, :n].T
Q = tQ[:, :n].T; Z = tZ[:, :n].T; Sm = tH[:, :n].T; A0 = tH0[:, :n].T
# back-transform the residual into the coordinates of the final form: E = S - Q^T A0 Z
E = Sm - Q.T @ A0 @ Z
print('rc', rc, st['sweeps'], st['aeds'], '||E||/||A|| in u', (E.norm() / A0.norm()).item() * 2**52)
v, idx = torch.topk(E.abs().flatten(), 12)
for val, k in zip(v.tolist(), idx.tolist()):
    print('E[%d,%d] = %.3e  S=%.3e' % (k // n, k % n, val, Sm[k // n, k % n].item()))
# also in original coordinates
E2 = Q @ Sm @ Z.T - A0
v, idx = torch.topk(E2.abs().flatten(), 5)
for val, k in zip(v.tolist(), idx.tolist()):
    print('E2[%d,%d] = %.3e' % (k // n, k % n, val))
