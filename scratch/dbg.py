import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1,1,S.NO_MESSAGES)
def run(n, pw=-1, q=True):
    tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
    tA = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    rc = S.hessenberg_device(tA, tQ, n=n, panel_width=pw)
    rc2, chk = S.check_device(tQ, tA, tA0, n=n)
    print(n, pw, rc, {k: (round(v,1) if isinstance(v,float) else v) for k,v in chk.items()}, flush=True)
for n in (2000, 3000, 4000, 4200, 5000, 6000):
    run(n)
for pw in (32, 64, 128, 256):
    run(6000, pw)
