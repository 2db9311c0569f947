"""Generates tests/golden/gep_lcg2019_n*.npz -- run in the build container (needs scipy).

Inputs: the reference test driver's random Hessenberg-triangular pencil
(test/schur/experiment.c:203-207: generate_random_hessenberg then generate_random_uptriag
on the one LCG stream of test/common/common.c:56-59, seed 2019; test/common/init.c:122-138,
159-175), written here independently in numpy.  Outputs: the generalized eigenvalues LAPACK
computes (scipy.linalg.eigvals(H, R) -> dggev -> dhgeqz, the routine the reference itself
calls on its small pencils, schur/cpu_utils.c:2287) and the pencil norms.

Only data is stored: a few columns of the inputs (to pin the generator) and the
eigenvalues sorted by (real, imag).
"""
import os

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))


class Lcg:
    def __init__(self, seed=2019):
        self.s = seed

    def draw(self):
        self.s = (self.s * 1103515245 + 12345) & 0x7FFFFFFF
        return 2.0 * (self.s / 0x7FFFFFFF) - 1.0


def pencil(n, seed=2019):
    g = Lcg(seed)
    H = np.zeros((n, n))
    R = np.zeros((n, n))
    for j in range(n):
        for i in range(min(n, j + 2)):
            H[i, j] = g.draw()
    for j in range(n):
        for i in range(min(n, j + 1)):
            R[i, j] = g.draw()
    return H, R


def well_conditioned(R):
    """Random triangular matrices are exponentially ill-conditioned in n; this variant of the
    same LCG data (strict upper part / sqrt(n), diagonal 1 + |r_ii|) keeps the eigenvalues of
    the pencil well defined so that they can be compared to a few thousand u."""
    n = R.shape[0]
    return np.triu(R, 1) / np.sqrt(n) + np.diag(1.0 + np.abs(np.diag(R)))


def spread_u(a, b):
    """largest relative distance between two eigenvalue sets after greedy matching, in u"""
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
    for kind in ("lcg2019", "wellcond2019"):
        for n in (48, 150, 400):
            H, R = pencil(n)
            if kind == "wellcond2019":
                R = well_conditioned(R)
            ev = sl.eigvals(H, R)                       # dggev -> dhgeqz
            AA, BB, _, _ = sl.qz(H, R, output="complex")  # zgges -> zhgeqz: a second LAPACK route
            ev2 = np.diag(AA) / np.diag(BB)
            # sensitivity: movement of the eigenvalues under elementwise relative perturbations of
            # one u (what a backward error of 1 u may do), worst of 5 trials
            rng = np.random.default_rng(2019)
            sens = 0.0
            for _ in range(5):
                Hp = H * (1.0 + 2.0 ** -52 * rng.standard_normal(H.shape))
                Rp = R * (1.0 + 2.0 ** -52 * rng.standard_normal(R.shape))
                sens = max(sens, spread_u(ev, sl.eigvals(Hp, Rp)))
            order = np.lexsort((ev.imag, ev.real))
            ev = ev[order]
            np.savez_compressed(
                os.path.join(HERE, f"gep_{kind}_n{n}.npz"),
                n=n, seed=2019, h_col0=H[:, 0], h_last_col=H[:, -1], r_col1=R[:, 1], r_last_col=R[:, -1],
                h_fro=np.linalg.norm(H), r_fro=np.linalg.norm(R),
                eig_real=ev.real, eig_imag=ev.imag,
                lapack_spread_u=spread_u(ev, ev2), sens_u_per_u=sens)
            print(kind, n, "ok", np.abs(ev).min(), np.abs(ev).max(), "LAPACK real-vs-complex QZ spread (u):",
                  spread_u(ev, ev2), "sensitivity (u per u):", sens)


if __name__ == "__main__":
    main()
