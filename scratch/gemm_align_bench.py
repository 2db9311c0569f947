"""Does the row / column offset of the operands (the trailing block starts at row i + 1) cost the fused update anything?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
L = S.lib.load()
m, k, ld = 19000, 624, 20608
A = torch.rand((k, ld), dtype=torch.float64, device="cuda") - 0.5      # ld x k
B = torch.rand((k, ld), dtype=torch.float64, device="cuda") - 0.5
Cm = torch.rand((20000, 20000), dtype=torch.float64, device="cuda") - 0.5
def run(ro, co, reps=5):
    pa = A.data_ptr() + 8 * ro; pb = B.data_ptr() + 8 * co; pc = Cm.data_ptr() + 8 * (co * 20000 + ro)
    f = lambda: L.starneig_amd_dgemm_device(b"N", b"T", m, m, k, -1.0, pa, ld, pb, ld, 1.0, pc, 20000, None)
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"row offset {ro:3d} col offset {co:3d}: {ms:7.3f} ms {2.0 * m * m * k / ms / 1e9 / 78.6 * 100:5.1f} %", flush=True)
for ro, co in ((0, 0), (1, 0), (0, 1), (1, 1), (313, 624), (8, 8), (16, 16), (2, 2)):
    run(ro, co)
