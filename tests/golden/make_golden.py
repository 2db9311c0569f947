"""Generates tests/golden/*.npz -- run in the build container (needs scipy).

Inputs are the reference test driver's deterministic matrices (LCG of
test/common/common.c:56-59, seed 2019, `fullpos` fill of test/common/init.c:
108-120, written here independently in numpy), outputs come from LAPACK
(scipy's OpenBLAS): dgehrd+dorghr -- the comparator the reference test driver
offers as `--solver lapack` (test/hessenberg/solvers.c:231-283) -- and dhseqr /
eigvals for the Schur leg (test/schur/solvers.c:135-169).

Only data is stored (inputs are regenerated from the seed): for each n the
sub-diagonal, diagonal, first row and last column of H, ||A||_F, and the
eigenvalues sorted by (real, imag).
"""
import os

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))


def lcg_fullpos(n, seed=2019):
    vals = np.empty(n * n, dtype=np.float64)
    s = seed
    for k in range(n * n):
        s = (s * 1103515245 + 12345) & 0x7FFFFFFF
        vals[k] = s / 0x7FFFFFFF
    return vals.reshape((n, n), order="F")      # column by column


def main():
    for n in (64, 200, 512, 2000):
        A = lcg_fullpos(n)
        H, Q = sl.hessenberg(A, calc_q=True)
        ev = np.linalg.eigvals(A)
        order = np.lexsort((ev.imag, ev.real))
        ev = ev[order]
        np.savez_compressed(
            os.path.join(HERE, f"hessenberg_lcg2019_n{n}.npz"),
            n=n, seed=2019,
            a_first_col=A[:, 0], a_last_col=A[:, -1], a_fro=np.linalg.norm(A),
            h_subdiag=np.diag(H, -1), h_diag=np.diag(H), h_first_row=H[0, :],
            h_last_col=H[:, -1], h_fro=np.linalg.norm(H),
            eig_real=ev.real, eig_imag=ev.imag)
        print(n, "ok", np.linalg.norm(Q @ H @ Q.T - A) / np.linalg.norm(A))


if __name__ == "__main__":
    main()
