import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, scipy.linalg as sl
import starneig_amd as S, oracle as O
import torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(int(os.environ.get("SN_CORES", "1")),1,S.NO_MESSAGES)
def run(name, H0):
    n = H0.shape[0]
    H = np.asfortranarray(H0.copy()); Q = np.asfortranarray(np.eye(n))
    real = np.zeros(n); imag = np.zeros(n)
    t = time.time(); rc = S.SEP_SM_Schur(n, H, n, Q, n, real, imag); dt = time.time() - t
    res = np.linalg.norm(Q @ H @ Q.T - H0) / max(np.linalg.norm(H0), 1e-300) / 2.0**-52
    orth = np.linalg.norm(Q @ Q.T - np.eye(n)) / np.sqrt(n) / 2.0**-52
    form = O.check_schur_form(np.asfortranarray(H))
    print(f"{name:34s} n={n:5d} rc={rc} form={form} res={res:8.1f}u orth={orth:8.1f}u t={dt:.2f}s", flush=True)
rng = np.random.RandomState(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
run("orthogonal Hessenberg", sl.hessenberg(sl.qr(rng.randn(n, n))[0]))
run("companion (z^n - 1)", np.eye(n, k=-1) + np.eye(n)[:, [0]] @ np.eye(n)[[n-1], :])
c = np.eye(n, k=-1); c[0, :] = rng.randn(n) * 1e-3; run("companion random", c)
run("Toeplitz tridiagonal (-1,2,-1)", 2*np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1))
run("Jordan-like (1 on superdiag)", np.eye(n) * 3 + np.eye(n, k=1) + 1e-8*np.eye(n, k=-1))
run("graded 10^(-i/40)", sl.hessenberg(rng.randn(n, n) * np.logspace(0, -15, n)[:, None]))
run("symmetric random", sl.hessenberg((lambda M: M + M.T)(rng.randn(n, n))))
run("rank-1 + I", sl.hessenberg(np.eye(n) + np.outer(rng.randn(n), rng.randn(n))))
run("all ones Hessenberg", np.triu(np.ones((n, n)), -1))
run("zero matrix", np.zeros((n, n)))
run("large scale 1e150", sl.hessenberg(rng.randn(300, 300)) * 1e150)
run("tiny scale 1e-150", sl.hessenberg(rng.randn(300, 300)) * 1e-150)
run("random 3000", sl.hessenberg(rng.randn(3000, 3000)))
