"""GPU: the fp64 MFMA GEMM (rows H4-H8, S3) through the C-ABI against a plain
fp64 reference (numpy on the host).  Tolerance: |C - C_ref| <= 4*k*u*|A||B| + |beta C|u
elementwise bound, checked as max-norm relative to (|A| |B|)."""
import numpy as np
import pytest

from helpers import U, to_device, to_host

pytestmark = pytest.mark.gpu

SHAPES = [
    # (transA, transB, m, n, k)
    ("N", "T", 300, 257, 40),       # H4/H6/H8 rank-nb update shape
    ("T", "N", 130, 40, 517),       # H5: W = A^T (V T), long k
    ("N", "N", 211, 48, 333),       # H7: W = X (V T)
    ("T", "N", 96, 500, 96),        # S3 left update  lQ^T X
    ("N", "N", 500, 96, 96),        # S3 right update X lQ
    ("T", "T", 65, 33, 17),
    ("N", "T", 1, 1, 1),
    ("N", "N", 16, 16, 4),
    ("N", "T", 1000, 1000, 312),    # 128x128 tile path
    ("T", "N", 312, 300, 5001),     # split-K path (few output tiles, long k; beta = 0 only)
    ("T", "N", 40, 33, 2500),
    ("N", "N", 700, 48, 2500),      # split-K, mn-contiguous row operand (W = Q (V T))
    ("T", "N", 1500, 100, 4100),    # split-K with 128x64 tiles
    ("N", "T", 700, 515, 624),      # fused trailing update shape (k = 2 nb), interior + edge tiles
]


@pytest.mark.parametrize("ta,tb,m,n,k", SHAPES)
@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (-1.0, 1.0), (0.5, -2.0)])
def test_dgemm_matches_fp64_reference(node, ta, tb, m, n, k, alpha, beta):
    rng = np.random.RandomState(m * 131 + n * 7 + k)
    a_shape = (m, k) if ta == "N" else (k, m)
    b_shape = (k, n) if tb == "N" else (n, k)
    lda, ldb, ldc = a_shape[0] + 3, b_shape[0] + 1, m + 5
    A = np.zeros((lda, a_shape[1]), order="F"); A[:a_shape[0]] = rng.uniform(-1, 1, a_shape)
    B = np.zeros((ldb, b_shape[1]), order="F"); B[:b_shape[0]] = rng.uniform(-1, 1, b_shape)
    C = np.zeros((ldc, n), order="F"); C[:m] = rng.uniform(-1, 1, (m, n))
    C[m:] = 777.0                                    # padding must stay untouched
    opA = A[:a_shape[0]] if ta == "N" else A[:a_shape[0]].T
    opB = B[:b_shape[0]] if tb == "N" else B[:b_shape[0]].T
    ref = alpha * (opA @ opB) + beta * C[:m]
    bound = (abs(alpha) * (np.abs(opA) @ np.abs(opB)) * 4 * k + abs(beta) * np.abs(C[:m]) * 2 + 1e-300) * U
    tA, tB, tC = to_device(A), to_device(B), to_device(C)
    assert node.dgemm_device(ta, tb, m, n, k, alpha, tA, lda, tB, ldb, beta, tC, ldc) == 0
    out = to_host(tC)
    assert np.all(np.abs(out[:m] - ref) <= bound)
    assert np.all(out[m:] == 777.0)


def test_dgemm_identity_asymmetric(node):
    """A = I with an asymmetric B catches a transposed C/D register map
    (cdna_hip_programming.md section 3)."""
    n = 64
    B = np.asfortranarray(np.arange(n * n, dtype=np.float64).reshape(n, n))
    I = np.asfortranarray(np.eye(n))
    C = np.asfortranarray(np.zeros((n, n)))
    tC = to_device(C)
    node.dgemm_device("N", "N", n, n, n, 1.0, to_device(I), n, to_device(B), n, 0.0, tC, n)
    assert np.array_equal(to_host(tC), B)
    node.dgemm_device("T", "T", n, n, n, 1.0, to_device(B), n, to_device(I), n, 0.0, tC, n)
    assert np.array_equal(to_host(tC), B.T)


@pytest.mark.parametrize("m,n", [(3000, 3000), (4200, 4200), (3300, 3300), (2900, 4500), (3007, 3001)])
def test_big_tile_launch_with_a_cut_last_round(node, m, n):
    """128 x 128 launches of more than 512 tiles cut the tiles of their last, partly filled round of
    workgroups into halves or quarters (csrc/dgemm_mfma.hip): 576 tiles -> 64 quartered, 1089 -> 65
    quartered, 676 -> 164 halved, 828 -> whole tiles, and one odd size.  Against torch.matmul in fp64."""
    import torch
    k = 312
    g = torch.Generator(device="cuda").manual_seed(m * 7 + n)
    A = torch.rand((k, m), dtype=torch.float64, device="cuda", generator=g) - 0.5       # column-major m x k
    B = torch.rand((k, n), dtype=torch.float64, device="cuda", generator=g) - 0.5       # column-major n x k
    C0 = torch.rand((n, m), dtype=torch.float64, device="cuda", generator=g) - 0.5      # column-major m x n
    C = C0.clone()
    assert node.dgemm_device("N", "T", m, n, k, -1.0, A, m, B, n, 1.0, C, m) == 0
    torch.cuda.synchronize()
    ref = C0 - B.T @ A                  # (A_cm B_cm^T)^T = B_t^T A_t in the tensors' layout
    bound = (torch.abs(B).T @ torch.abs(A)) * 4 * k * U + torch.abs(C0) * 2 * U
    assert bool((torch.abs(C - ref) <= bound).all())
