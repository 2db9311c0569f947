import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
for r in range(reps):
    tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    torch.cuda.synchronize(); t = time.time()
    rc, st = S.hessenberg_device(tH, tQ, n=n, stats=True, sample_every=16)
    torch.cuda.synchronize(); dt = time.time() - t
    bw = st["gemv_sampled_bytes"] / (st["gemv_sampled_ms"] * 1e-3) / 1e9 if st["gemv_sampled_ms"] else 0
    print("hess n=%d rc=%d %.3fs  gemv sampled %.0f GB/s" % (n, rc, dt, bw), flush=True)
