# time of the device-resident eigenvalue reordering (30 % / 50 % of the spectrum selected)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(int(os.environ.get("SN_CORES", "-1")), 1, S.NO_MESSAGES)
n = int(sys.argv[1]); frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
tS = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
S.hessenberg_device(tS, tQ, n=n)
rc, real, imag, _ = S.schur_device(tS, tQ, n=n)
thr = np.quantile(real, 1.0 - frac)
sel = (real > thr).astype(np.int32)
for i in range(n - 1):
    if imag[i] > 0.0: sel[i] = sel[i + 1] = max(sel[i], sel[i + 1])
k = int(sel.sum())
torch.cuda.synchronize(); t = time.time()
rc, r2, i2, st = S.reorder_schur_device(tS, tQ, sel, n=n)
torch.cuda.synchronize(); dt = time.time() - t
_, c = S.check_device(tQ, tS, tA0, n=n)
print(f"reorder n={n} selected {k} ({frac:.0%}): rc={rc} {dt:.2f}s windows={st['windows']} rounds={st['rounds']} gemm {st['gemm_flops']/1e12:.2f} TFLOP "
      f"res={c['residual_u']:.0f}u orth={c['orthogonality_u']:.0f}u placed={int(sel.sum())}", flush=True)
