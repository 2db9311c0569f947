"""Hessenberg + Schur legs of one process (as bench.py times them): python scratch/hess_schur_pair.py n reps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(16, 1, S.NO_MESSAGES)
n = int(sys.argv[1]); reps = int(sys.argv[2])
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
for r in range(reps):
    tH = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    torch.cuda.synchronize(); t = time.time()
    rc, st = S.hessenberg_device(tH, tQ, n=n, stats=True, sample_every=16)
    torch.cuda.synchronize(); t1 = time.time()
    rc, re, im, st2 = S.schur_device(tH, tQ, n=n)
    torch.cuda.synchronize(); t2 = time.time()
    print("hess %.3f s  schur %.3f s  total %.3f s" % (t1 - t, t2 - t1, t2 - t), flush=True)
