"""Two-stage Hessenberg-triangular reduction with and without the accumulation of Q and Z: what the side stream's
compact-WY applications cost the chase.  python scratch/r6_ht_noqz.py n"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
import starneig_amd as S
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1])
for withqz in (True, False, True, False):
    tA, tB = S.device_matrix(n), S.device_matrix(n)
    S.lcg_fill_device(tA, n, n, seed=2019); S.lcg_fill_device(tB, n, n, seed=77)
    tQ = tZ = None
    if withqz:
        tQ, tZ = S.device_matrix(n), S.device_matrix(n)
        S.set_matrix_device(tQ, n, n, 0.0, 1.0); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
    torch.cuda.synchronize(); t0 = time.time()
    rc, st = S.hessenberg_triangular_device(tA, tB, tQ, tZ, n=n)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"n={n} Q,Z={'yes' if withqz else 'no '}: {dt:.3f} s (QR {st['qr_ms']/1e3:.3f}, stage 1 {st['stage1_ms']/1e3:.3f}, stage 2 {st['rotation_ms']/1e3 - st['stage1_ms']/1e3:.3f})", flush=True)
S.node_finalize()
