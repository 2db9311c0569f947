"""The fused update timed alone after (a) nothing, (b) 60 ms of idle, (c) 60 ms of an HBM-bound stream (what
the panel factorisation is): does the phase before it change its speed (clock ramp)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
m, k = 19000, 624
A = torch.rand((k, m), dtype=torch.float64, device="cuda") - 0.5
B = torch.rand((k, m), dtype=torch.float64, device="cuda") - 0.5
Cm = torch.rand((m, m), dtype=torch.float64, device="cuda") - 0.5
big = torch.rand((20000, 20000), dtype=torch.float64, device="cuda")
v = torch.rand((20000,), dtype=torch.float64, device="cuda")
def gemm(): S.dgemm_device("N", "T", m, m, k, -1.0, A, m, B, m, 1.0, Cm, m)
def timed(pre):
    ts = []
    for _ in range(6):
        pre()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); gemm(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sum(ts[1:]) / (len(ts) - 1)
    return ms, 2.0 * m * m * k / ms / 1e9 / 78.6 * 100
def idle(): torch.cuda.synchronize(); time.sleep(0.06)
def stream():
    for _ in range(90): torch.mv(big, v)      # ~0.6 ms each: 3.2 GB at ~5.5 TB/s
gemm(); torch.cuda.synchronize()
for name, pre in (("back to back", lambda: None), ("after 60 ms idle", idle), ("after 55 ms of HBM-bound gemv", stream), ("back to back", lambda: None)):
    ms, pct = timed(pre)
    print(f"{name:32s}: {ms:7.3f} ms {pct:5.1f} %", flush=True)
