# Schur leg only on an n x n LCG matrix reduced by the device Hessenberg (for kernel traces)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(int(os.environ.get("SN_CORES", "16")), 1, S.NO_MESSAGES)
n = int(sys.argv[1])
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
S.hessenberg_device(tH, tQ, n=n)
torch.cuda.synchronize()
t = time.time()
rc, real, imag, st = S.schur_device(tH, tQ, n=n)
torch.cuda.synchronize()
print("schur", rc, "%.2fs" % (time.time() - t), st, flush=True)
