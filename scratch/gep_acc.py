import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import starneig_amd as S, oracle as O
S.node_init(1, 1, S.NO_MESSAGES)
torch.zeros(1, device='cuda')
for kind, n in (("lcg2019", 400), ("lcg2019", 48)):
    g = np.load(f'/root/repo/tests/golden/gep_{kind}_n{n}.npz')
    gold = g['eig_real'] + 1j * g['eig_imag']
    H0, R0 = O.random_pencil_wellcond(n) if kind.startswith("well") else O.random_pencil(n)
    nrm = np.linalg.norm(H0)
    for scale in (1.0, 1e-2, 1e-4, 1e-6):
        for small in (96, 200, 1000):
            conf = S.schur_init_conf()
            conf.left_threshold = scale * 2.0 ** -52 * nrm
            conf.small_limit = small
            H, R = H0.copy(order='F'), R0.copy(order='F')
            Q, Z = O.identity(n), O.identity(n)
            ar, ai, be = np.zeros(n), np.zeros(n), np.zeros(n)
            rc = S.GEP_SM_Schur_expert(conf, n, H, H.shape[0], R, R.shape[0], Q, Q.shape[0], Z, Z.shape[0], ar, ai, be)
            ev = (ar + 1j * ai) / be
            print(kind, n, 'thres x', scale, 'small', small, 'rc', rc, 'eig err u', O.match_eigenvalues(ev, gold),
                  'resA', O.pencil_residual_u(Q, H, Z, H0), 'resB', O.pencil_residual_u(Q, R, Z, R0),
                  'sens', float(g['sens_u_per_u']), flush=True)
