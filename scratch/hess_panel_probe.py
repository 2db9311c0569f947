# How orthogonal is the block reflector I - V (VT)^T of the FIRST panel, and where is its defect?
#   STARNEIG_AMD_TUNING=1 SN_HESS_MAX_PANELS=1 python scratch/hess_panel_probe.py n
# Uses the test-hook build (libstarneig_amd_test.so) for the whole run: it exports the product's C-ABI too.
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import starneig_amd as S
from starneig_amd import lib as L
L.LIB_PATH = L.TEST_LIB_PATH                      # one library instance: product entry points + hooks
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1])
tA = S.device_matrix(n); S.lcg_fill_device(tA, n, n)
tQ = S.device_matrix(n); S.set_matrix_device(tQ, n, n, 0.0, 1.0)
nb = S.default_panel_width(n)
rc = S.hessenberg_device(tA, tQ, n=n)
lib = L.load()
dp = C.POINTER(C.c_double)
ldp = C.c_int(0)
cap = (n + 1024) * nb
V = np.zeros(cap); VT = np.zeros(cap); scal = np.zeros(4 * nb)
assert lib.sn_internal_hess_panel_factors(0, nb, V.ctypes.data_as(dp), VT.ctypes.data_as(dp), scal.ctypes.data_as(dp), C.byref(ldp)) == 0
ld = ldp.value
V = V[: ld * nb].reshape(nb, ld).T[1:n].astype(np.longdouble)      # rows R0 = 1 .. n-1 (global row index)
VT = VT[: ld * nb].reshape(nb, ld).T[1:n].astype(np.longdouble)
tau = scal.reshape(nb, 4)[:, 1].astype(np.longdouble)
u = np.longdouble(2.0) ** -52
G = V.T @ V                                                          # extended precision
d = np.array([tau[j] * G[j, j] - 2 for j in range(nb)], dtype=np.longdouble)
print("tau_j v_j'v_j - 2 (in u): rms %.2f max %.2f" % (float(np.sqrt(np.mean(d ** 2)) / u), float(np.abs(d).max() / u)))
# T from VT = V T (least squares in extended precision), then Delta = V'V - T^-1 - T^-T
T = np.linalg.solve(G.astype(np.float64), (V.T @ VT).astype(np.float64)).astype(np.longdouble)
# refine once in extended precision
R = (V.T @ VT) - G @ T
T = T + np.linalg.solve(G.astype(np.float64), R.astype(np.float64)).astype(np.longdouble)
print("strictly lower part of T (should vanish): max |.| = %.2e" % float(np.abs(np.tril(T, -1)).max()))
Ti = np.linalg.inv(np.triu(T).astype(np.float64)).astype(np.longdouble)
Ti = Ti + Ti @ (np.eye(nb, dtype=np.longdouble) - np.triu(T) @ Ti)   # one Newton step
Delta = G - Ti - Ti.T
off = Delta - np.diag(np.diag(Delta))
print("Delta = V'V - T^-1 - T^-T: diagonal rms %.2f u, off-diagonal rms %.2f u, ||Delta||_F %.1f u" % (
    float(np.sqrt(np.mean(np.diag(Delta) ** 2)) / u), float(np.sqrt(np.mean(off ** 2)) / u), float(np.linalg.norm(Delta.astype(np.float64)) / float(u))))
D = np.eye(n - 1) - (V @ VT.T).astype(np.float64)
Dd = D.T @ D - np.eye(n - 1)
print("defect of I - V VT': ||.||_F = %.1f u  (/sqrt(n) = %.2f u)" % (np.linalg.norm(Dd) / float(u), np.linalg.norm(Dd) / float(u) / np.sqrt(n)))
# the same with VT recomputed from V'V in extended precision (what an exact T would give)
M = np.triu(G, 1) + np.diag(np.diag(G) / 2)
Tx = np.linalg.inv(M.astype(np.float64))
Dx = np.eye(n - 1) - V.astype(np.float64) @ Tx @ V.astype(np.float64).T
print("with T = (triu(V'V) with halved diagonal)^-1: %.2f u" % (np.linalg.norm(Dx.T @ Dx - np.eye(n - 1)) / float(u) / np.sqrt(n)))
# the Q the library produced for this panel (Q = I on entry) against I - VT V' from its own factors
Qd = np.asfortranarray(tQ.cpu().numpy().T)[:n, :n]
Qref = np.eye(n)
Qref[1:, 1:] -= (VT @ V.T).astype(np.float64)          # rows R0.. x cols R0..  (W2 = I VT)
# W2 = Q(:, R0:E) VT has n rows: row 0 of Q is e_0, column 0 outside R0:E -> nothing else changes
diff = np.abs(Qd - Qref)
print("max |Q_device - (I - VT V')| = %.2e at %s; ||.||_F/u = %.1f" % (diff.max(), np.unravel_index(diff.argmax(), diff.shape), np.linalg.norm(diff) / float(u)))
print("defect of Q_device: %.2f u" % (np.linalg.norm(Qd.T @ Qd - np.eye(n)) / float(u) / np.sqrt(n)))
rows = np.linalg.norm(diff, axis=1); cols = np.linalg.norm(diff, axis=0)
print("row profile of the difference (tenths): ", " ".join("%.1f" % (np.linalg.norm(r) / float(u)) for r in np.array_split(rows, 10)))
print("col profile of the difference (tenths): ", " ".join("%.1f" % (np.linalg.norm(c) / float(u)) for c in np.array_split(cols, 10)))
