# Where does the rounding error of the Hessenberg + Schur chain come from?
#   python scratch/acc_diag.py n [lapack]
# Prints, in units of u: the Hessenberg leg alone, the Schur leg alone (Q reset to I, measured
# against the Hessenberg matrix), the chain, and the column profile of ||Q^T q_j - e_j||.
# With "lapack": the same through scipy's LAPACK (dgehrd+dorghr, dhseqr) on the host.
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1])
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
torch.cuda.synchronize(); t = time.time()
S.hessenberg_device(tH, tQ, n=n)
torch.cuda.synchronize(); th = time.time() - t
_, c = S.check_device(tQ, tH, tA0, n=n)
print(f"n={n} hessenberg {th:.2f}s res={c['residual_u']:.1f}u orth={c['orthogonality_u']:.1f}u", flush=True)
tH0 = tH.clone(); tQh = tQ.clone()
# Schur leg alone
tQ2 = S.device_matrix(n); S.set_matrix_device(tQ2, n, n, 0.0, 1.0)
tS = tH0.clone()
torch.cuda.synchronize(); t = time.time()
rc, real, imag, st = S.schur_device(tS, tQ2, n=n)
torch.cuda.synchronize(); ts = time.time() - t
_, c = S.check_device(tQ2, tS, tH0, n=n)
print(f"n={n} schur alone {ts:.2f}s rc={rc} sweeps={st['sweeps']} aeds={st['aeds']} aed_host={st['aed_host_s']:.2f}s "
      f"res={c['residual_u']:.1f}u orth={c['orthogonality_u']:.1f}u", flush=True)
# column profile of the orthogonality error of the Schur factor
Q = tQ2[:, :n]                      # Q[c, r] = Q(r, c): rows of this tensor are columns of Q
G = Q @ Q.T                         # G[i, j] = q_i . q_j
G -= torch.eye(n, dtype=torch.float64, device="cuda")
colerr = torch.linalg.norm(G, dim=1) / 2.0 ** -52
k = 10
chunks = colerr.reshape(k, -1).mean(dim=1) if n % k == 0 else colerr[: n // k * k].reshape(k, -1).mean(dim=1)
print("  per-column ||Q^T q_j - e_j|| (u), mean over tenths of the columns:", " ".join(f"{x:.0f}" for x in chunks.tolist()), flush=True)
del G
# chain
tS2 = tH0.clone(); tQ3 = tQh.clone()
rc, real, imag, st = S.schur_device(tS2, tQ3, n=n)
_, c = S.check_device(tQ3, tS2, tA0, n=n)
print(f"n={n} chain res={c['residual_u']:.1f}u orth={c['orthogonality_u']:.1f}u", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "lapack":
    import scipy.linalg as sl
    from scipy.linalg import lapack
    A = np.asfortranarray(tA0[:, :n].cpu().numpy().T)
    t = time.time()
    H, Qh = sl.hessenberg(A, calc_q=True)
    t1 = time.time() - t
    u = 2.0 ** -52
    res = np.linalg.norm(Qh @ H @ Qh.T - A) / np.linalg.norm(A) / u
    orth = np.linalg.norm(Qh @ Qh.T - np.eye(n)) / np.sqrt(n) / u
    print(f"n={n} LAPACK dgehrd+dorghr {t1:.1f}s res={res:.1f}u orth={orth:.1f}u", flush=True)
    t = time.time()
    T, Z = sl.schur(H, output="real")
    t2 = time.time() - t
    res = np.linalg.norm(Z @ T @ Z.T - H) / np.linalg.norm(H) / u
    orth = np.linalg.norm(Z @ Z.T - np.eye(n)) / np.sqrt(n) / u
    print(f"n={n} LAPACK dhseqr (gees on H) {t2:.1f}s res={res:.1f}u orth={orth:.1f}u", flush=True)
