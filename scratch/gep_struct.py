import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import starneig_amd as S, oracle as O
from helpers import to_host
S.node_init(1, 1, S.NO_MESSAGES); torch.zeros(1, device='cuda')
n = int(sys.argv[1])
tH, tR = S.device_matrix(n), S.device_matrix(n)
S.lcg_pencil_device(tH, tR, n)
tQ, tZ = S.device_matrix(n), S.device_matrix(n)
S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
rc, ar, ai, be, st = S.gep_schur_device(tH, tR, tQ, tZ, n=n)
Sm, T = to_host(tH)[:n], to_host(tR)[:n]
print('rc', rc, 'violations', O.check_gep_schur_form(np.asfortranarray(Sm), np.asfortranarray(T)))
print('below2', np.count_nonzero(np.tril(Sm, -2)), 'Tlow', np.count_nonzero(np.tril(T, -1)))
sub = np.diag(Sm, -1)
for k in np.nonzero(sub)[0]:
    a11, a12, a21, a22 = Sm[k, k], Sm[k, k+1], Sm[k+1, k], Sm[k+1, k+1]
    b11, b12, b22 = T[k, k], T[k, k+1], T[k+1, k+1]
    p = b11*b22; q = a11*b22 + a22*b11 - a21*b12; r = a11*a22 - a12*a21
    disc = q*q - 4*p*r
    bad = (disc > 1e-9*(q*q+abs(4*p*r))) or b12 != 0 or not (b11 > 0) or not (b22 > 0) or (k+2 < n and Sm[k+2, k+1] != 0)
    if bad: print('block', k, 'A', [a11, a12, a21, a22], 'B', [b11, b12, b22], 'disc rel', disc/(q*q+abs(4*p*r)))
