"""The fused trailing update C -= A B^T (k = 624) over the sizes the Hessenberg reduction sees."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
tot_f = tot_t = 0.0
for m in (20000, 17500, 15000, 12500, 10000, 7500, 5000, 3500):
    k = 624
    A = torch.rand((k, m), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((k, m), dtype=torch.float64, device="cuda") - 0.5
    C = torch.rand((m, m), dtype=torch.float64, device="cuda") - 0.5
    S.dgemm_device("N", "T", m, m, k, -1.0, A, m, B, m, 1.0, C, m)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        S.dgemm_device("N", "T", m, m, k, -1.0, A, m, B, m, 1.0, C, m)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * m * m * k / ms / 1e9
    tiles = ((m + 127) // 128) ** 2
    print(f"m={m:6d}: {ms:7.3f} ms {tf:6.1f} TFLOP/s {tf / 78.6 * 100:5.1f} %  tiles {tiles} = {tiles / 512:.2f} rounds", flush=True)
    tot_f += 2.0 * m * m * k; tot_t += ms
print(f"flop-weighted: {tot_f / tot_t / 1e9 / 78.6 * 100:.1f} %")
