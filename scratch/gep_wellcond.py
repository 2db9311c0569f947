# QZ leg on the well-conditioned pencil family (host API), sizes from argv
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import starneig_amd as S, oracle as O
S.node_init(1, 1, S.NO_MESSAGES); torch.zeros(1, device='cuda')
for n in [int(a) for a in sys.argv[1:]]:
    H0, R0 = O.random_pencil_wellcond(n)
    H, R = H0.copy(order='F'), R0.copy(order='F')
    Q, Z = O.identity(n), O.identity(n)
    ar, ai, be = np.zeros(n), np.zeros(n), np.zeros(n)
    t = time.time()
    rc = S.GEP_SM_Schur(n, H, H.shape[0], R, R.shape[0], Q, Q.shape[0], Z, Z.shape[0], ar, ai, be)
    dt = time.time() - t
    print(n, 'rc', rc, 't %.2f' % dt, 'form', O.check_gep_schur_form(H, R),
          'resA %.0f resB %.0f' % (O.pencil_residual_u(Q, H, Z, H0), O.pencil_residual_u(Q, R, Z, R0)),
          'orth %.0f %.0f' % (O.orthogonality_u(Q), O.orthogonality_u(Z)), flush=True)
