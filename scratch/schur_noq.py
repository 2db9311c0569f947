import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1])
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
S.hessenberg_device(tH, tQ, n=n)
torch.cuda.synchronize()
for withq in (True, False, True, False):
    tS = tH.clone(); tQ2 = tQ.clone()
    torch.cuda.synchronize(); t = time.time()
    rc, real, imag, st = S.schur_device(tS, tQ2 if withq else None, n=n)
    torch.cuda.synchronize()
    print("schur with Q" if withq else "schur without Q", "%.2fs" % (time.time() - t), "aed_host %.2f wait %.2f" % (st["aed_host_s"], st["gpu_wait_s"]), flush=True)
