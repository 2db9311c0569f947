import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1,1,S.NO_MESSAGES)
def run(n, aed=-1, ns=-1, small=-1):
    tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
    tA = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    rc, hst = S.hessenberg_device(tA, tQ, n=n, stats=True)
    rc2, chk0 = S.check_device(tQ, tA, tA0, n=n)
    conf = S.schur_init_conf(); conf.aed_window_size=aed; conf.shift_count=ns; conf.small_limit=small
    rc, real, imag, st = S.schur_device(tA, tQ, n=n, conf=conf)
    torch.cuda.synchronize()
    rc2, chk = S.check_device(tQ, tA, tA0, n=n)
    print(n, aed, ns, small, "hess res %.1f"%chk0["residual_u"], "schur rc", rc, st["sweeps"], st["aeds"], "res %.3g orth %.3g"%(chk["residual_u"], chk["orthogonality_u"]), flush=True)
run(1500)
