"""Generates tests/golden/dlag2_cases.npz -- run in the build container (needs scipy).

LAPACK DLAG2 (scipy's bundled OpenBLAS, through ctypes) on 2 x 2 pencils (A, B), B upper triangular:
the routine the reference calls for every 2 x 2 block of a generalized Schur form
(common/math.c:148-176).  Stored: the inputs and the five outputs (scale1, scale2, wr1, wr2, wi).
Cases: random pencils, complex pairs, close real pairs (relative gap 1e-3 ... 1e-15: the discriminant
nearly vanishes), standardised blocks (B diagonal and positive), graded scales, near-singular B.
Pins oracle/gep_oracle.c:oracle_dlag2 (tests/test_oracle_gep.py) and, through the test library,
the product's host kernel (tests/test_schur_host_gep.py).
"""
import ctypes as C
import glob
import os

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))


def lapack_dlag2():
    libs = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
    f = C.CDLL(libs[0]).scipy_dlag2_
    f.restype = None

    def call(A, B):
        a = np.asfortranarray(A, dtype=np.float64); b = np.asfortranarray(B, dtype=np.float64)
        out = [C.c_double(0.0) for _ in range(5)]
        two = C.c_int(2)
        safmin = C.c_double(np.finfo(np.float64).tiny)
        f(a.ctypes.data_as(C.c_void_p), C.byref(two), b.ctypes.data_as(C.c_void_p), C.byref(two),
          C.byref(safmin), *[C.byref(o) for o in out])
        return [o.value for o in out]
    return call


def cases(rng):
    out = []
    for _ in range(200):                                    # random
        A = rng.standard_normal((2, 2)); B = np.triu(rng.standard_normal((2, 2)))
        out.append((A, B))
    for _ in range(200):                                    # complex pairs, standardised form
        A = rng.standard_normal((2, 2)); A[1, 0] = -abs(A[1, 0]) - 0.1; A[0, 1] = abs(A[0, 1]) + 0.1
        B = np.diag(rng.random(2) + 0.1)
        out.append((A, B))
    for k in range(300):                                    # close real pairs / nearly double eigenvalues
        lam = rng.standard_normal() * 10.0 ** rng.integers(-3, 4)
        gap = 10.0 ** (-rng.integers(3, 16))
        D = np.diag([lam, lam * (1.0 + gap)])
        Q, _ = np.linalg.qr(rng.standard_normal((2, 2))); Z, _ = np.linalg.qr(rng.standard_normal((2, 2)))
        T = np.triu(rng.standard_normal((2, 2))) + 2.0 * np.eye(2)
        A = Q @ (T @ D) @ Z; Bm = Q @ T @ Z
        # bring B back to upper triangular form by a rotation from the left
        r = np.hypot(Bm[0, 0], Bm[1, 0]); c, s_ = Bm[0, 0] / r, Bm[1, 0] / r
        G = np.array([[c, s_], [-s_, c]])
        A = G @ A; Bm = G @ Bm; Bm[1, 0] = 0.0
        out.append((A, Bm))
    for _ in range(100):                                    # graded scales
        sa, sb = 10.0 ** rng.integers(-140, 140), 10.0 ** rng.integers(-140, 140)
        A = rng.standard_normal((2, 2)) * sa; B = np.triu(rng.standard_normal((2, 2))) * sb
        out.append((A, B))
    for _ in range(50):                                     # near-singular B
        A = rng.standard_normal((2, 2)); B = np.triu(rng.standard_normal((2, 2)))
        B[rng.integers(0, 2), :] *= 10.0 ** -rng.integers(150, 300)
        B[1, 0] = 0.0
        out.append((A, B))
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(2019)
    dlag2 = lapack_dlag2()
    cs = cases(rng)
    A = np.array([c[0] for c in cs]); B = np.array([c[1] for c in cs])
    out = np.array([dlag2(a, b) for a, b in cs])
    np.savez_compressed(os.path.join(HERE, "dlag2_cases.npz"), A=A, B=B, out=out)
    print(len(cs), "cases;", int((out[:, 4] != 0).sum()), "complex")
