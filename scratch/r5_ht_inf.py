"""Known pencil with 1 % infinite eigenvalues (test_known_eigenvalues_generalized): the small diagonal entries of T
after the Hessenberg-triangular reduction, rotation path against two-stage path, and what QZ makes of them."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
import starneig_amd as S
import oracle as O
from helpers import to_device, to_host
S.node_init(4, 1, S.NO_MESSAGES)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
A0, B0, kr, ki, kb = O.known_pencil(n, generalized=True, zero_ratio=0.0)
ninf = int((kb == 0.0).sum())
tA, tB = to_device(A0), to_device(B0)
ld = tA.shape[1]
tQ, tZ = S.device_matrix(n, ld=ld), S.device_matrix(n, ld=ld)
S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
rc, st = S.hessenberg_triangular_device(tA, tB, tQ, tZ, n=n)
T = to_host(tB)[:n, :n]
d = np.sort(np.abs(np.diag(T))) / np.linalg.norm(T) / 2.0 ** -52
print(f"n={n} two_stage={st['two_stage']} prescribed infinite {ninf}; |T_ii| / (u ||T||_F), smallest {ninf + 4}:")
print(np.array2string(d[:ninf + 4], precision=2, max_line_width=200))
rc, ar, ai, be, st2 = S.gep_schur_device(tA, tB, tQ, tZ, n=n)
nb = np.linalg.norm(T) * 2.0 ** -52
nz = np.where(be != 0)[0]
order = nz[np.argsort(np.abs(be[nz]))][:4]
print("smallest nonzero |beta| / (u ||B||_F):", [(int(i), float(abs(be[i]) / nb), float(ar[i]), float(ai[i])) for i in order], "zeros at", np.where(be == 0)[0][:60])
print({k: st2[k] for k in st2 if k in ("sweeps", "aeds", "infinite", "pushed_infinite", "total_ms")})
print("QZ rc", rc, "beta == 0:", int((be == 0).sum()), "largest finite |lambda|:", np.sort(np.hypot(ar[be != 0], ai[be != 0]) / be[be != 0])[-3:])
S.node_finalize()
