"""One GEMM shape, a few launches (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
ta, tb, m, n, k = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
beta = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
ar, ac = (m, k) if ta == "N" else (k, m)
br, bc = (k, n) if tb == "N" else (n, k)
A = torch.rand((ac, ar), dtype=torch.float64, device="cuda") - 0.5
B = torch.rand((bc, br), dtype=torch.float64, device="cuda") - 0.5
C = torch.rand((n, m), dtype=torch.float64, device="cuda") - 0.5
for _ in range(4):
    S.dgemm_device(ta, tb, m, n, k, -1.0, A, ar, B, br, beta, C, m)
torch.cuda.synchronize()
