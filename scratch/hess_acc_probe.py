# Hessenberg rounding-error probes: orthogonality of Q after the first k panels (Q = I on entry), and
# against the panel width.   python scratch/hess_acc_probe.py n
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1])
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
def orth(tQ):
    Q = tQ[:, :n]
    G = Q @ Q.T - torch.eye(n, dtype=torch.float64, device="cuda")
    return float(torch.linalg.norm(G)) / n ** 0.5 / 2.0 ** -52
for pw in [int(x) for x in (sys.argv[2:] or ["-1"])]:
    tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    rc = S.hessenberg_device(tH, tQ, n=n, panel_width=pw)
    torch.cuda.synchronize()
    _, c = S.check_device(tQ, tH, tA0, n=n)
    print(f"n={n} panel_width={pw} rc={rc} orth={orth(tQ):.2f}u check: res={c['residual_u']:.1f} orth={c['orthogonality_u']:.1f}", flush=True)
