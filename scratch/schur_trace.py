"""One Hessenberg + one Schur reduction at n (for rocprofv3 --kernel-trace): usage schur_trace.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import starneig_amd as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(S.USE_ALL, 1, S.NO_MESSAGES)
tA = S.device_matrix(n); S.lcg_fill_device(tA, n, n, seed=2019, mode=0)
tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
rc, st = S.hessenberg_device(tA, tQ, n=n, stats=True); torch.cuda.synchronize()
tH0, tQ0 = tA.clone(), tQ.clone()
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    tA.copy_(tH0); tQ.copy_(tQ0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc, re, im, sst = S.schur_device(tA, tQ, n=n)
    torch.cuda.synchronize()
    print(f"schur {time.perf_counter() - t0:.3f} s wait {sst['gpu_wait_s']:.2f} aed {sst['aed_host_s']:.2f}", flush=True)
