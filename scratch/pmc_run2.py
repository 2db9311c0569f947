"""The n = 20000 Hessenberg reduction, nothing else (no event sampling): for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tA = S.device_matrix(n); S.lcg_fill_device(tA, n, n)
tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
rc = S.hessenberg_device(tA, tQ, n=n)
torch.cuda.synchronize()
print("rc", rc)
