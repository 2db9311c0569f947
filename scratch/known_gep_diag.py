"""Diagnostics for the generalized `--init known` experiment (tests/test_gpu_testdriver.py)."""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle as O
import starneig_amd as S
from helpers import to_device, to_host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(-1, 1, S.NO_MESSAGES)
A0, B0, kr, ki, kb = O.known_pencil(n, generalized=True, zero_ratio=0.0)
tA, tB = to_device(A0), to_device(B0)
ld = tA.shape[1]
tQ, tZ = S.device_matrix(n, ld=ld), S.device_matrix(n, ld=ld)
S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
rc, st = S.hessenberg_triangular_device(tA, tB, tQ, tZ, n=n)
print("HT", rc, st)
H, T = to_host(tA), to_host(tB)
print("HT: min |diag T| / ||T||:", np.sort(np.abs(np.diag(T[:n])))[:10] / np.linalg.norm(T[:n]))
rc, ar, ai, be, st = S.gep_schur_device(tA, tB, tQ, tZ, n=n)
torch.cuda.synchronize()
print("QZ", rc, st, "beta==0:", int((be == 0).sum()), "prescribed:", int((kb == 0).sum()))
Sm, Tm = to_host(tA), to_host(tB)
print("form check:", O.check_gep_schur_form(Sm, Tm))
for k in range(n - 1):
    if Sm[k + 1, k] != 0.0:
        a11, a12, a21, a22 = Sm[k, k], Sm[k, k + 1], Sm[k + 1, k], Sm[k + 1, k + 1]
        b11, b12, b22 = Tm[k, k], Tm[k, k + 1], Tm[k + 1, k + 1]
        p, q, r = b11 * b22, a11 * b22 + a22 * b11 - a21 * b12, a11 * a22 - a12 * a21
        bad = []
        if k + 2 < n and Sm[k + 2, k + 1] != 0.0: bad.append("consecutive")
        if q * q - 4 * p * r > 1e-9 * (q * q + abs(4 * p * r)): bad.append("real pair")
        if b12 != 0.0: bad.append("b12")
        if not (b11 > 0 and b22 > 0): bad.append("sign")
        if bad:
            print(k, bad, "S:", a11, a12, a21, a22, "T:", b11, b12, b22, "alpha/beta", ar[k:k + 2], ai[k:k + 2], be[k:k + 2])
print(O.known_eigenvalues_check((ar, ai, be), (kr, ki, kb)))
