import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import starneig_amd as S, oracle as O
from helpers import to_host
S.node_init(1, 1, S.NO_MESSAGES)
torch.zeros(1, device='cuda')
for n in [int(a) for a in sys.argv[1:]] or [64, 150, 400, 1000]:
    H0, R0 = O.random_pencil_wellcond(n)
    H, R = H0.copy(order='F'), R0.copy(order='F')
    Q, Z = O.identity(n), O.identity(n)
    ar, ai, be = np.zeros(n), np.zeros(n), np.zeros(n)
    t = time.time()
    rc = S.GEP_SM_Schur(n, H, H.shape[0], R, R.shape[0], Q, Q.shape[0], Z, Z.shape[0], ar, ai, be)
    dt = time.time() - t
    print(n, 'rc', rc, 't', round(dt, 3), 'form', O.check_gep_schur_form(H, R),
          'resA', O.pencil_residual_u(Q, H, Z, H0), 'resB', O.pencil_residual_u(Q, R, Z, R0),
          'orth', O.orthogonality_u(Q), O.orthogonality_u(Z), flush=True)
for n in (3000, 6000, 12000):
    tH, tR = S.device_matrix(n), S.device_matrix(n)
    S.lcg_pencil_device(tH, tR, n)
    tH0, tR0 = tH.clone(), tR.clone()
    tQ, tZ = S.device_matrix(n), S.device_matrix(n)
    S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
    torch.cuda.synchronize(); t = time.time()
    rc, ar, ai, be, st = S.gep_schur_device(tH, tR, tQ, tZ, n=n)
    torch.cuda.synchronize(); dt = time.time() - t
    _, ca = S.check_pencil_device(tQ, tH, tZ, tH0, n=n)
    _, cb = S.check_pencil_device(tQ, tR, tZ, tR0, n=n)
    print(n, 'rc', rc, 't', round(dt, 3), st, ca, cb, flush=True)
