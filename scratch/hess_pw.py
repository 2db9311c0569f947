# Hessenberg reduction time against the panel width (default: the reference's formula, 312 at n = 20000)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1])
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
for pw in [int(a) for a in sys.argv[2:]]:
    ts = []
    for r in range(int(os.environ.get('REPS', '2'))):
        tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
        torch.cuda.synchronize(); t = time.time()
        rc, st = S.hessenberg_device(tH, tQ, n=n, panel_width=pw, stats=True, sample_every=16)
        torch.cuda.synchronize(); dt = time.time() - t; ts.append(dt)
    _, c = S.check_device(tQ, tH, tA0, n=n)
    print(f"panel width {pw}: rc={rc} {min(ts[1:] or ts):.3f}s (all {[round(x,3) for x in ts]}) residual {c['residual_u']:.0f}u orth {c['orthogonality_u']:.0f}u", flush=True)
