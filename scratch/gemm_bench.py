import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1,1,S.NO_MESSAGES)
def bench(ta, tb, m, n, k, beta=1.0, reps=5, zero=False):
    ar, ac = (m, k) if ta == "N" else (k, m)
    br, bc = (k, n) if tb == "N" else (n, k)
    A = torch.rand((ac, ar), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((bc, br), dtype=torch.float64, device="cuda") - 0.5
    Cm = torch.rand((n, m), dtype=torch.float64, device="cuda") - 0.5
    if zero:
        A.zero_(); B.zero_(); Cm.zero_()
    S.dgemm_device(ta, tb, m, n, k, -1.0, A, ar, B, br, beta, Cm, m)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        S.dgemm_device(ta, tb, m, n, k, -1.0, A, ar, B, br, beta, Cm, m)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * m * n * k / ms / 1e9
    print(f"{ta}{tb} m={m:6d} n={n:6d} k={k:6d} beta={beta}: {ms:8.3f} ms  {tf:6.1f} TFLOP/s  ({tf/78.6*100:4.1f}% of 78.6)", flush=True)
bench("N", "N", 8192, 8192, 8192, 0.0, reps=10)
bench("N", "N", 8192, 8192, 8192, 0.0, reps=10, zero=True)
bench("N", "T", 20000, 20000, 312)
bench("N", "T", 20000, 20000, 312, zero=True)
bench("N", "T", 20000, 20000, 624)                  # fused trailing update A -= [Y V][V' W]^T
bench("N", "T", 10000, 10000, 624)
bench("T", "N", 20000, 312, 20000, 0.0)             # W = A^T (V T)  (split-K)
bench("N", "N", 20000, 312, 20000, 0.0)             # W = Q (V T)    (split-K)
bench("N", "T", 20000, 20000, 312)                  # Q -= W V^T
