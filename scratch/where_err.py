# Where does the residual of a structured input live?  E = Q^T A Q - T split into: below the
# sub-diagonal (what deflations truncated), the sub-diagonal, diagonal band, far upper triangle.
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import starneig_amd as S
from test_gpu_baseline_configs import structured
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1]); kind = sys.argv[2]
tH0 = S.device_matrix(n)
if kind == "lcg":
    S.lcg_fill_device(tH0, n, n); tq = S.device_matrix(n); S.set_matrix_device(tq, n, n, 0.0, 1.0)
    S.hessenberg_device(tH0, tq, n=n); del tq
else:
    structured(kind, n, tH0[:, :n])
tH = tH0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
conf = None
if len(sys.argv) > 3:
    conf = S.schur_init_conf(); conf.aed_window_size, conf.shift_count = int(sys.argv[3]), int(sys.argv[4])
rc, real, imag, st = S.schur_device(tH, tQ, n=n, conf=conf)
_, c = S.check_device(tQ, tH, tH0, n=n)
u = 2.0 ** -52
A = tH0[:, :n].T.contiguous(); Q = tQ[:, :n].T.contiguous(); T = tH[:, :n].T.contiguous()
E = Q.T @ A @ Q - T
nA = torch.linalg.norm(A).item()
i = torch.arange(n, device="cuda")
d = i[None, :] - i[:, None]          # column - row
def part(mask): return (torch.linalg.norm(E * mask).item() / nA / u)
print(f"{kind} n={n} rc={rc} sweeps={st['sweeps']} aeds={st['aeds']} res={c['residual_u']:.0f}u orth={c['orthogonality_u']:.0f}u  |E|/|A|={torch.linalg.norm(E).item()/nA/u:.0f}u: "
      f"below-subdiag {part(d < -1):.0f}u, subdiag {part(d == -1):.0f}u, band 0..+96 {part((d >= 0) & (d <= 96)):.0f}u, "
      f"upper 97..1000 {part((d > 96) & (d <= 1000)):.0f}u, far upper {part(d > 1000):.0f}u", flush=True)
# row profile of the upper-triangle error (tenths)
Eu = E * (d >= 0)
rows = torch.linalg.norm(Eu, dim=1).reshape(10, -1).pow(2).sum(dim=1).sqrt() / nA / u
cols = torch.linalg.norm(Eu, dim=0).reshape(10, -1).pow(2).sum(dim=1).sqrt() / nA / u
print("  upper error by row tenths:", " ".join(f"{x:.0f}" for x in rows.tolist()))
print("  upper error by col tenths:", " ".join(f"{x:.0f}" for x in cols.tolist()))
