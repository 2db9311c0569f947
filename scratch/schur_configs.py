import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import starneig_amd as S
if os.environ.get("SN_USE_TEST_LIB"):       # the hooks build as the product (SN_AED_DUMP lives there)
    import starneig_amd.lib as _l
    _l.LIB_PATH = _l.TEST_LIB_PATH
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(int(os.environ.get("SN_CORES", "1")),1,S.NO_MESSAGES)
n = int(sys.argv[1])
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
tH = tA0.clone(); tQ0 = S.device_matrix(n); S.set_matrix_device(tQ0, n, n, 0.0, 1.0)
rc, hst = S.hessenberg_device(tH, tQ0, n=n, stats=True)
torch.cuda.synchronize()
print("hess %.2fs" % (hst["total_ms"]/1e3), flush=True)
for cfg in sys.argv[2:]:
    parts = [int(x) for x in cfg.split(",")]
    aed, ns, small = parts[:3]
    nib = parts[3] if len(parts) > 3 else -1
    tA = tH.clone(); tQ = tQ0.clone()
    conf = S.schur_init_conf(); conf.aed_window_size=aed; conf.shift_count=ns; conf.small_limit=small; conf.aed_nibble=nib
    t=time.time()
    rc, real, imag, st = S.schur_device(tA, tQ, n=n, conf=conf)
    torch.cuda.synchronize(); dt=time.time()-t
    rc2, chk = S.check_device(tQ, tA, tA0, n=n)
    print(cfg, "rc", rc, "%.2fs"%dt, "sweeps", st["sweeps"], "aeds", st["aeds"], "chase", st["chase_launches"], "gemmTF %.1f"%(st["gemm_flops"]/1e12), "aed_host %.2fs wait %.2fs"%(st["aed_host_s"], st["gpu_wait_s"]), "res %.0f orth %.0f"%(chk["residual_u"], chk["orthogonality_u"]), flush=True)
