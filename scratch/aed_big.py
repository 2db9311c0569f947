# Schur leg at n with the reference's default AED window / shift count (0.08 n / 0.06 n), which takes
# the blocked device AED (row S5), against this library's small-window default.
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(int(os.environ.get("SN_CORES", "16")), 1, S.NO_MESSAGES)
n = int(sys.argv[1])
tA0 = S.device_matrix(n); S.lcg_fill_device(tA0, n, n)
tH0 = tA0.clone(); tQ0 = S.device_matrix(n); S.set_matrix_device(tQ0, n, n, 0.0, 1.0)
S.hessenberg_device(tH0, tQ0, n=n)
for name, cfg in (("warm-up", None), ("default", None), ("reference sizes", (int(0.08 * n), int(0.06 * n) // 2 * 2))):
    conf = None
    if cfg:
        conf = S.schur_init_conf(); conf.aed_window_size, conf.shift_count = cfg
    tH, tQ = tH0.clone(), tQ0.clone()
    torch.cuda.synchronize(); t = time.time()
    rc, real, imag, st = S.schur_device(tH, tQ, n=n, conf=conf)
    torch.cuda.synchronize(); dt = time.time() - t
    _, c = S.check_device(tQ, tH, tA0, n=n)
    print(f"n={n} {name} {cfg}: rc={rc} {dt:.2f}s sweeps={st['sweeps']} aeds={st['aeds']} aed_s={st['aed_host_s']:.2f} "
          f"res={c['residual_u']:.0f}u orth={c['orthogonality_u']:.0f}u", flush=True)
