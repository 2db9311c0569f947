"""The GEMM shapes of the QR step of the Hessenberg-triangular reduction (csrc/hess_tri.hip ht_qr_step), alone:
ours against torch.matmul (rocBLAS), microseconds and the rate of the bytes the product has to move.
python scratch/r6_qr_gemm.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starneig_amd as S
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
S.node_init(1, 1, S.NO_MESSAGES)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
nb = 64


def timed(f, reps=6):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(ta, tb, m, nn, k, beta, label):
    ar, ac = (m, k) if ta == "N" else (k, m)
    br, bc = (k, nn) if tb == "N" else (nn, k)
    A = torch.rand((ac, ar), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((bc, br), dtype=torch.float64, device="cuda") - 0.5
    C = torch.rand((nn, m), dtype=torch.float64, device="cuda") - 0.5
    ours = timed(lambda: S.dgemm_device(ta, tb, m, nn, k, -1.0, A, ar, B, br, beta, C, m))
    oa = A if ta == "N" else A.t()            # torch holds the column-major arrays as their transposes: C^T = op(B)^T op(A)^T
    ob = B if tb == "N" else B.t()
    ref = timed(lambda: torch.matmul(ob, oa))
    byts = 8.0 * (m * k + k * nn + (2 if beta else 1) * m * nn)
    print(f"{label:34s} {ta}{tb} m={m:6d} n={nn:6d} k={k:6d}: ours {ours:8.1f} us ({byts / ours / 1e6:6.2f} TB/s, {2.0 * m * nn * k / ours / 1e6:6.1f} TF/s)   rocBLAS {ref:8.1f} us")


for m in (n, n // 2, n // 8):
    case("T", "N", nb, n, m, 0.0, "W = VT' A      (64 x n x m)")
    case("N", "N", m, n, nb, 1.0, "A -= V W       (m x n x 64)")
    case("N", "N", n, nb, m, 0.0, "W = Q VT       (n x 64 x m)")
    case("N", "T", n, m, nb, 1.0, "Q -= W V'      (n x m x 64)")
    case("T", "N", nb, nb, m, 0.0, "G = V' V       (64 x 64 x m)")
    case("N", "N", m, nb, nb, 0.0, "VT = V T       (m x 64 x 64)")
