# structured matrices at large n, device resident (residuals through check_device)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1])
def run(name, make):
    tH0 = S.device_matrix(n)
    M = tH0[:, :n]                  # M[c, r] = H(r, c)
    make(M)
    tH = tH0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    torch.cuda.synchronize(); t = time.time()
    rc, real, imag, st = S.schur_device(tH, tQ, n=n)
    torch.cuda.synchronize(); dt = time.time() - t
    _, c = S.check_device(tQ, tH, tH0, n=n)
    print(f"{name:28s} n={n} rc={rc} t={dt:.2f}s sweeps={st['sweeps']} aeds={st['aeds']} res={c['residual_u']:.0f}u orth={c['orthogonality_u']:.0f}u", flush=True)
def all_ones(M):
    M.copy_(torch.tril(torch.ones((n, n), dtype=torch.float64, device="cuda"), 1))      # H upper Hessenberg <=> M = H^T lower + 1 super
def toeplitz(M):
    M.zero_(); idx = torch.arange(n, device="cuda")
    M[idx, idx] = 2.0; M[idx[:-1], idx[1:]] = -1.0; M[idx[1:], idx[:-1]] = -1.0
def sym_random(M):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    M.zero_(); idx = torch.arange(n, device="cuda")
    d = torch.randn(n, generator=g, device="cuda", dtype=torch.float64); e = torch.randn(n - 1, generator=g, device="cuda", dtype=torch.float64)
    M[idx, idx] = d; M[idx[:-1], idx[1:]] = e; M[idx[1:], idx[:-1]] = e
run("all ones Hessenberg", all_ones)
run("Toeplitz tridiagonal", toeplitz)
run("symmetric tridiagonal random", sym_random)
