"""Elementwise distance of the single-GPU Hessenberg form from the oracle at n = 1500 / 2000: where it is largest,
how far two GPU runs are from each other, cached against streaming gemv (SN_HESS_CACHE_MB)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
import starneig_amd as S
import oracle as O
from helpers import elementwise_tolerance
S.node_init(4, 1, S.NO_MESSAGES)
for n in (777, 1500, 2000):
    A0 = O.random_fullpos(n)
    Ao = A0.copy(order="F"); Qo = O.identity(n)
    O.hessenberg(Ao, Qo)
    nrm = np.linalg.norm(A0[:n]); tol = elementwise_tolerance(n)
    runs = []
    for rep in range(4):
        A = A0.copy(order="F"); Q = O.identity(n)
        assert S.SEP_SM_Hessenberg(n, A, A.shape[0], Q, Q.shape[0]) == 0
        runs.append(A[:n].copy())
        D = np.abs(A[:n] - Ao[:n]) / nrm / tol
        i, j = np.unravel_index(np.argmax(D), D.shape)
        colmax = D.max(axis=0)
        print(f"n={n} rep {rep}: max err {D.max():.3f} tol at ({i},{j}) |H|={abs(Ao[i,j]):.3g}; by column quartile: "
              f"{[round(float(colmax[k * n // 4:(k + 1) * n // 4].max()), 3) for k in range(4)]}; residual {O.residual_u(Q, A, A0):.1f} u", flush=True)
    print(f"n={n}: run-to-run max distance {max(np.abs(runs[0] - r).max() for r in runs[1:]) / nrm / tol:.3f} tol")
S.node_finalize()
