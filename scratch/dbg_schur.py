import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1,1,S.NO_MESSAGES)
def run(n):
    tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
    tA = tA0.clone(); tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    rc, hst = S.hessenberg_device(tA, tQ, n=n, stats=True)
    t=time.time()
    rc, real, imag, st = S.schur_device(tA, tQ, n=n)
    torch.cuda.synchronize(); dt=time.time()-t
    rc2, chk = S.check_device(tQ, tA, tA0, n=n)
    print(n, "hess %.2fs"%(hst["total_ms"]/1e3), "schur rc", rc, "%.2fs"%dt, st, {k: (round(v,1) if isinstance(v,float) else v) for k,v in chk.items()}, "trace err %.2e"%abs(real.sum()-n/2), flush=True)
for n in [int(x) for x in sys.argv[1:]]:
    run(n)
