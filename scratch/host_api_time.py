# One call of the host-array API (pageable arrays, PCIe-inclusive) after a warm-up: where the time goes
# (STARNEIG_AMD_TUNING=1 SN_SCHUR_PROFILE=1 prints the shim's own breakdown).
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(-1, 1, S.NO_MESSAGES)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(1)
A0 = np.asfortranarray(rng.standard_normal((n, n)))
for rep in range(2):
    A = A0.copy(order="F"); Q = np.asfortranarray(np.eye(n))
    t = time.time(); rc = S.SEP_SM_Hessenberg(n, A, n, Q, n); t1 = time.time()
    real = np.zeros(n); imag = np.zeros(n)
    rc2 = S.SEP_SM_Schur(n, A, n, Q, n, real, imag); t2 = time.time()
    print(f"rep {rep}: Hessenberg {t1 - t:.3f} s rc={rc}, Schur {t2 - t1:.3f} s rc={rc2}", flush=True)
# the same matrix through the device-pointer entry for comparison
tA = S.device_matrix(n); tA[:, :n].copy_(torch.from_numpy(A0.T.copy())); tQ = S.device_matrix(n)
for rep in range(2):
    tH = tA.clone(); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
    torch.cuda.synchronize(); t = time.time()
    rc, st = S.hessenberg_device(tH, tQ, n=n, stats=True)
    torch.cuda.synchronize(); print(f"device entry rep {rep}: {time.time() - t:.3f} s (events {st['total_ms'] / 1e3:.3f} s)", flush=True)
