"""Two-stage Hessenberg-triangular reduction (csrc/ht_twostage.hip, SN_HT_TWOSTAGE=1) against the rotation path:
structure, residuals, orthogonality, time.  python scratch/r5_ht2.py n [n ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
import starneig_amd as S
S.node_init(S.USE_ALL, 1, S.NO_MESSAGES)
for n in [int(a) for a in sys.argv[1:]] or [300]:
    tA, tB = S.device_matrix(n), S.device_matrix(n)
    S.lcg_fill_device(tA, n, n, seed=2019); S.lcg_fill_device(tB, n, n, seed=77)
    tA0, tB0 = tA.clone(), tB.clone()
    tQ, tZ = S.device_matrix(n), S.device_matrix(n)
    S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
    torch.cuda.synchronize(); t0 = time.time()
    rc, st = S.hessenberg_triangular_device(tA, tB, tQ, tZ, n=n)
    torch.cuda.synchronize(); dt = time.time() - t0
    _, ca = S.check_pencil_device(tQ, tA, tZ, tA0, n=n)
    _, cb = S.check_pencil_device(tQ, tB, tZ, tB0, n=n)
    H = tA[:n, :n].T; T = tB[:n, :n].T
    below_h = float(torch.tril(H, -2).abs().max()); below_t = float(torch.tril(T, -1).abs().max())
    print(f"n={n} rc={rc} two_stage={st['two_stage']} {dt:.3f} s (QR {st['qr_ms']/1e3:.3f}, stage 1 {st['stage1_ms']/1e3:.3f}, rest {st['rotation_ms']/1e3 - st['stage1_ms']/1e3:.3f}) "
          f"below sub-diagonal {below_h:.2e} / diagonal {below_t:.2e}; residuals {ca['residual_u']:.1f} / {cb['residual_u']:.1f} u, "
          f"orthogonality {ca['orthogonality_q_u']:.1f} / {ca['orthogonality_z_u']:.1f} u", flush=True)
S.node_finalize()
