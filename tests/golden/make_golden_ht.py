"""Generates tests/golden/ht_lcg2019_n*.npz -- run in the build container (needs scipy).

Inputs: the reference test driver's generalized Hessenberg input -- two generate_random_fullpos
matrices (A, then B) on the one LCG stream of test/common/common.c:56-59, seed 2019
(test/hessenberg/experiment.c:102-106), written here independently in numpy.  Outputs: what the
LAPACK calls of starneig_GEP_SM_HessenbergTriangular (wrappers/lapack.c:143-163) produce, with
the unblocked dgghrd in place of dgghd3 (same rotations, same order as the oracle restates; dgghd3
groups them differently and is compared on the backward error only):
    dgeqrf(B); dormqr('L','T') on A; dormqr('R','N') on Q = I; B <- R; dgghrd('V','V').
The routines live in scipy's bundled OpenBLAS (LP64, symbols prefixed scipy_) and are reached
through ctypes because scipy wraps no dgghrd.

Only data is stored: the pencil (H, T), the first rows of Q and Z, norms, and a few input
columns to pin the generator.
"""
import ctypes as C
import glob
import os

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))


def lcg_fullpos_pair(n, seed=2019):
    s = seed
    out = []
    for _ in range(2):
        M = np.zeros((n, n), order="F")
        for j in range(n):
            for i in range(n):
                s = (s * 1103515245 + 12345) & 0x7FFFFFFF
                M[i, j] = s / 0x7FFFFFFF
        out.append(M)
    return out


def openblas():
    libs = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
    return C.CDLL(libs[0])


def lapack_ht(A, B, blocked=False):
    """The LAPACK sequence of wrappers/lapack.c:143-163 on copies; returns H, T, Q, Z."""
    L = openblas()
    n = A.shape[0]
    A = np.asfortranarray(A.copy()); B = np.asfortranarray(B.copy())
    Q = np.asfortranarray(np.eye(n)); Z = np.asfortranarray(np.eye(n))
    ci, vp = C.c_int, C.c_void_p
    p = lambda a: a.ctypes.data_as(vp)
    N, ONE, info = ci(n), ci(1), ci(0)
    tau = np.zeros(n)
    lwork = 64 * n + 1024
    work = np.zeros(lwork)
    LW = ci(lwork)
    L.scipy_dgeqrf_(C.byref(N), C.byref(N), p(B), C.byref(N), p(tau), p(work), C.byref(LW), C.byref(info))
    assert info.value == 0
    L.scipy_dormqr_(C.c_char_p(b"L"), C.c_char_p(b"T"), C.byref(N), C.byref(N), C.byref(N), p(B), C.byref(N),
                    p(tau), p(A), C.byref(N), p(work), C.byref(LW), C.byref(info), C.c_size_t(1), C.c_size_t(1))
    assert info.value == 0
    L.scipy_dormqr_(C.c_char_p(b"R"), C.c_char_p(b"N"), C.byref(N), C.byref(N), C.byref(N), p(B), C.byref(N),
                    p(tau), p(Q), C.byref(N), p(work), C.byref(LW), C.byref(info), C.c_size_t(1), C.c_size_t(1))
    assert info.value == 0
    B[:] = np.triu(B)
    if blocked:
        L.scipy_dgghd3_(C.c_char_p(b"V"), C.c_char_p(b"V"), C.byref(N), C.byref(ONE), C.byref(N), p(A), C.byref(N),
                        p(B), C.byref(N), p(Q), C.byref(N), p(Z), C.byref(N), p(work), C.byref(LW), C.byref(info),
                        C.c_size_t(1), C.c_size_t(1))
    else:
        L.scipy_dgghrd_(C.c_char_p(b"V"), C.c_char_p(b"V"), C.byref(N), C.byref(ONE), C.byref(N), p(A), C.byref(N),
                        p(B), C.byref(N), p(Q), C.byref(N), p(Z), C.byref(N), C.byref(info),
                        C.c_size_t(1), C.c_size_t(1))
    assert info.value == 0
    return A, B, Q, Z


def main():
    u = 2.0 ** -52
    for n in (6, 40, 150):
        A, B = lcg_fullpos_pair(n)
        H, T, Q, Z = lapack_ht(A, B)
        assert np.count_nonzero(np.tril(H, -2)) == 0 and np.count_nonzero(np.tril(T, -1)) == 0
        ra = np.linalg.norm(Q @ H @ Z.T - A) / np.linalg.norm(A) / u
        rb = np.linalg.norm(Q @ T @ Z.T - B) / np.linalg.norm(B) / u
        H3, T3, Q3, Z3 = lapack_ht(A, B, blocked=True)
        ra3 = np.linalg.norm(Q3 @ H3 @ Z3.T - A) / np.linalg.norm(A) / u
        print(f"n={n}: dgghrd residuals {ra:.1f} / {rb:.1f} u, dgghd3 residual {ra3:.1f} u")
        np.savez_compressed(
            os.path.join(HERE, f"ht_lcg2019_n{n}.npz"),
            n=n, H=H, T=T, Q_head=Q[:4].copy(), Z_head=Z[:4].copy(),
            A_head=A[:, :2].copy(), B_head=B[:, :2].copy(),
            normA=np.linalg.norm(A), normB=np.linalg.norm(B),
            # the singular values of T^-1 H are invariant under the choice of rotations
            lapack_residual_u=np.array([ra, rb]))


if __name__ == "__main__":
    main()
