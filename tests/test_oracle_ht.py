"""The Hessenberg-triangular oracle (oracle/ht_oracle.c) against LAPACK golden vectors.

tests/golden/ht_lcg2019_n*.npz: dgeqrf + dormqr + dgghrd outputs for the reference test
driver's input (test/hessenberg/experiment.c:102-106), see make_golden_ht.py.  The oracle
applies the same rotations in the same order, so the comparison is elementwise.
"""
import os

import numpy as np
import pytest

import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
U = 2.0 ** -52


def load(n):
    return np.load(os.path.join(GOLDEN, f"ht_lcg2019_n{n}.npz"))


def run_oracle(n):
    A, B = oracle.random_fullpos_pair(n)
    A0, B0 = A[:n].copy(), B[:n].copy()
    Q, Z = oracle.identity(n), oracle.identity(n)
    oracle.hessenberg_triangular(A, B, Q, Z)
    return A0, B0, A[:n], B[:n], Q[:n], Z[:n]


@pytest.mark.parametrize("n", [6, 40, 150])
def test_generator_matches_fixture(n):
    g = load(n)
    A, B = oracle.random_fullpos_pair(n)
    assert np.array_equal(A[:n, :2], g["A_head"])
    assert np.array_equal(B[:n, :2], g["B_head"])
    assert np.linalg.norm(A[:n]) == pytest.approx(float(g["normA"]), rel=1e-14)


@pytest.mark.parametrize("n", [6, 40, 150])
def test_oracle_matches_lapack_elementwise(n):
    g = load(n)
    A0, B0, H, T, Q, Z = run_oracle(n)
    tol = 200 * n * U
    assert np.abs(H - g["H"]).max() <= tol * np.abs(g["H"]).max()
    assert np.abs(T - g["T"]).max() <= tol * np.abs(g["T"]).max()
    assert np.abs(Q[:4] - g["Q_head"]).max() <= tol
    assert np.abs(Z[:4] - g["Z_head"]).max() <= tol


@pytest.mark.parametrize("n", [1, 2, 3, 6, 40, 150])
def test_oracle_structure_and_backward_error(n):
    A0, B0, H, T, Q, Z = run_oracle(n)
    assert np.count_nonzero(np.tril(H, -2)) == 0
    assert oracle.count_below_diagonal(T) == 0
    lim = 40 * max(1.0, np.sqrt(n))
    assert np.linalg.norm(Q @ H @ Z.T - A0) / np.linalg.norm(A0) / U < lim
    assert np.linalg.norm(Q @ T @ Z.T - B0) / np.linalg.norm(B0) / U < lim
    assert np.linalg.norm(Q @ Q.T - np.eye(n)) / U < lim
    assert np.linalg.norm(Z @ Z.T - np.eye(n)) / U < lim


def test_lartg_conventions():
    # LAPACK 3.10+ dlartg: c >= 0, r carries the sign of f; f = 0 -> (0, sign(g), |g|)
    from scipy.linalg.lapack import dlartg
    for f, g in [(3.0, 4.0), (-3.0, 4.0), (3.0, -4.0), (-3.0, -4.0), (0.0, 2.0), (0.0, -2.0), (5.0, 0.0), (-5.0, 0.0)]:
        c, s, r = oracle.lartg(f, g)
        cl, sl_, rl = dlartg(f, g)
        assert (c, s, r) == pytest.approx((cl, sl_, rl), abs=1e-15)


def test_qr_step_alone():
    n = 60
    A, B = oracle.random_fullpos_pair(n)
    A0, B0 = A[:n].copy(), B[:n].copy()
    Q = oracle.identity(n)
    oracle.ht_qr(A, B, Q)
    Qm = Q[:n]
    assert oracle.count_below_diagonal(B) == 0
    assert np.linalg.norm(Qm @ B[:n] - B0) / np.linalg.norm(B0) / U < 50
    assert np.linalg.norm(Qm @ A[:n] - A0) / np.linalg.norm(A0) / U < 50
