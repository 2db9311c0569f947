# The generalized chain on the device: Hessenberg-triangular reduction, then QZ (starneig_GEP_SM_Reduce's two steps)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(-1, 1, S.NO_MESSAGES)
n = int(sys.argv[1])
tA = S.device_matrix(n); tB = S.device_matrix(n)
S.lcg_fill_device(tA, n, n, seed=2019); S.lcg_fill_device(tB, n, n, seed=77)
tA0, tB0 = tA.clone(), tB.clone()
tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
tZ = S.device_matrix(n); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
torch.cuda.synchronize(); t0 = time.time()
rc, st = S.hessenberg_triangular_device(tA, tB, tQ, tZ, n=n)
torch.cuda.synchronize(); t1 = time.time()
rc2, ar, ai, be, st2 = S.gep_schur_device(tA, tB, tQ, tZ, n=n)
torch.cuda.synchronize(); t2 = time.time()
_, ca = S.check_pencil_device(tQ, tA, tZ, tA0, n=n)
_, cb = S.check_pencil_device(tQ, tB, tZ, tB0, n=n)
print(f"n={n}: Hessenberg-triangular {t1-t0:.2f}s rc={rc}; QZ {t2-t1:.2f}s rc={rc2} sweeps={st2['sweeps']} aeds={st2['aeds']}; "
      f"chain residuals {ca['residual_u']:.0f} / {cb['residual_u']:.0f} u, orthogonality {ca['orthogonality_q_u']:.0f} / {ca['orthogonality_z_u']:.0f} u, "
      f"infinite eigenvalues {int((be == 0).sum())}", flush=True)
