"""Generates tests/golden/gep_general_lcg2019_n{2000,1600}.npz -- run in the build container (needs scipy; a few minutes).

The fixtures behind tests/test_gpu_ht_twostage.py::test_reduce_of_a_general_pencil_against_lapack: the
two-stage Hessenberg-triangular path (the product path from n = 1500 on) has no LAPACK counterpart to agree with
entry by entry, so the chain GEP_SM_Reduce = HessenbergTriangular + Schur is pinned on what LAPACK computes for the
SAME general pencil:

  n = 2000: the test driver's generalized Hessenberg input (two generate_random_fullpos matrices on one LCG stream,
            test/hessenberg/experiment.c:102-106, seed 2019; restated here in numpy, independently of oracle/):
            generalized eigenvalues by scipy.linalg.eigvals -> LAPACK dggev (dgeqrf + dgghrd + dhgeqz), their spread
            against a second LAPACK route (complex QZ) and their sensitivity to perturbations of one u -- the
            tolerance of the comparison is built from these two stored numbers, as for the small fixtures of
            make_golden_gep.py.
  n = 1600: the same generator with rows 10, 700 and 1599 of B set to zero (a rank deficiency of three): the number
            of infinite eigenvalues LAPACK returns (beta == 0 exactly) and its finite eigenvalues.

Only data is stored: the first and last column of A and B (to pin the generator), norms, eigenvalues.
"""
import os

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))


def fullpos_pair(n, seed=2019):
    """generate_random_fullpos twice on one stream (test/common/init.c:93-104: column by column, prand() / PRAND_MAX;
    test/common/common.c:56-59: the LCG)."""
    m = 0x7FFFFFFF
    s = seed
    out = np.empty(2 * n * n)
    for k in range(2 * n * n):
        s = (s * 1103515245 + 12345) & m
        out[k] = s / m
    A = out[:n * n].reshape((n, n), order="F").copy(order="F")
    B = out[n * n:].reshape((n, n), order="F").copy(order="F")
    return A, B


def spread_u(a, b):
    """largest relative distance between two eigenvalue sets after greedy matching, in u (oracle.match_eigenvalues)"""
    b = list(b)
    worst = 0.0
    scale = np.abs(a).max()
    for x in a[np.argsort(-np.abs(a))]:
        d = np.abs(np.array(b) - x)
        k = int(np.argmin(d))
        worst = max(worst, d[k] / max(abs(x), 1e-3 * scale))
        b.pop(k)
    return worst / 2.0 ** -52


def main():
    n = 2000
    A, B = fullpos_pair(n)
    ev = sl.eigvals(A, B)
    AA, BB, _, _ = sl.qz(A, B, output="complex")
    ev2 = np.diag(AA) / np.diag(BB)
    rng = np.random.default_rng(2019)
    sens = 0.0
    for _ in range(2):
        Ap = A * (1.0 + 2.0 ** -52 * rng.standard_normal(A.shape))
        Bp = B * (1.0 + 2.0 ** -52 * rng.standard_normal(B.shape))
        sens = max(sens, spread_u(ev, sl.eigvals(Ap, Bp)))
    order = np.lexsort((ev.imag, ev.real))
    ev = ev[order]
    np.savez_compressed(os.path.join(HERE, f"gep_general_lcg2019_n{n}.npz"), n=n, seed=2019,
        a_col0=A[:, 0], a_last_col=A[:, -1], b_col0=B[:, 0], b_last_col=B[:, -1],
        a_fro=np.linalg.norm(A), b_fro=np.linalg.norm(B), eig_real=ev.real, eig_imag=ev.imag,
        lapack_spread_u=spread_u(ev, ev2), sens_u_per_u=sens,
        b_singular_values=np.linalg.svd(B, compute_uv=False))
    print(n, "general pencil: |ev| in", np.abs(ev).min(), np.abs(ev).max(), "LAPACK spread (u)", spread_u(ev, ev2), "sensitivity (u per u)", sens)

    n = 1600
    A, B = fullpos_pair(n)
    for r in (10, 700, 1599):
        B[r, :] = 0.0
    w = sl.eigvals(A, B, homogeneous_eigvals=True)
    alpha, beta = w[0], w[1]
    inf = beta == 0.0
    fin = (alpha[~inf] / beta[~inf])
    order = np.lexsort((fin.imag, fin.real))
    fin = fin[order]
    np.savez_compressed(os.path.join(HERE, f"gep_general_lcg2019_n{n}.npz"), n=n, seed=2019, zero_rows=np.array([10, 700, 1599]),
        a_col0=A[:, 0], b_last_col=B[:, -1], n_infinite=int(inf.sum()), eig_real=fin.real, eig_imag=fin.imag,
        small_beta=np.sort(np.abs(beta))[:8])
    print(n, "singular B: infinite eigenvalues", int(inf.sum()), "smallest |beta|", np.sort(np.abs(beta))[:6], "largest finite |ev|", np.abs(fin).max())


if __name__ == "__main__":
    main()
