# Hessenberg-triangular reduction on the GPU: time and backward error.
#   python scratch/ht_time.py n [lapack]
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1])
tA = S.device_matrix(n); tB = S.device_matrix(n)
S.lcg_fill_device(tA, n, n, seed=2019); S.lcg_fill_device(tB, n, n, seed=77)
tA0, tB0 = tA.clone(), tB.clone()
tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
tZ = S.device_matrix(n); S.set_matrix_device(tZ, n, n, 0.0, 1.0)
torch.cuda.synchronize(); t = time.time()
noqz = os.environ.get('HT_NOQZ') == '1'
rc, st = S.hessenberg_triangular_device(tA, tB, None if noqz else tQ, None if noqz else tZ, n=n)
torch.cuda.synchronize(); t = time.time() - t
if noqz:
    print(f'n={n} without Q, Z: rotations {st["rotation_ms"]/1e3:.3f}s'); sys.exit(0)
_, ca = S.check_pencil_device(tQ, tA, tZ, tA0, n=n)
_, cb = S.check_pencil_device(tQ, tB, tZ, tB0, n=n)
print(f"n={n} rc={rc} wall {t:.2f}s qr {st['qr_ms']/1e3:.3f}s rotations {st['rotation_ms']/1e3:.3f}s "
      f"({st['rotation_ms']*1e6/max(st['rotations']/2,1):.0f} ns per chain step) "
      f"resA={ca['residual_u']:.1f}u resB={cb['residual_u']:.1f}u orthQ={ca['orthogonality_q_u']:.1f}u "
      f"orthZ={ca['orthogonality_z_u']:.1f}u below={ca['below_subdiagonal']}", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "lapack":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    from make_golden_ht import lapack_ht
    A = np.asfortranarray(tA0[:, :n].cpu().numpy().T); B = np.asfortranarray(tB0[:, :n].cpu().numpy().T)
    for blocked in (True, False):
        t = time.time(); H, T, Q, Z = lapack_ht(A, B, blocked=blocked); t = time.time() - t
        u = 2.0 ** -52
        print(f"n={n} LAPACK dgeqrf+dormqr+{'dgghd3' if blocked else 'dgghrd'} {t:.2f}s "
              f"resA={np.linalg.norm(Q @ H @ Z.T - A) / np.linalg.norm(A) / u:.1f}u", flush=True)
