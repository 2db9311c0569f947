"""CPU tests of oracle/testdriver_oracle.c: the reference test driver's `--init known` generator,
its `random` Schur input with `--decouple`, and its two eigenvalue hooks, exercised with LAPACK
(scipy) as the solver -- the comparator the reference's own driver ships (`--solver lapack`)."""
import numpy as np
import scipy.linalg as sl

import oracle as O


def lapack_schur_eigenvalues(A, n):
    T, _ = sl.schur(A[:n], output="real")
    Tp = np.zeros_like(A); Tp[:n] = T
    return O.extract_eigenvalues(np.asfortranarray(Tp))


def test_known_generator_structure_and_hook_with_lapack():
    n = 300
    A, _, kr, ki, kb = O.known_pencil(n, zero_ratio=0.0, inf_ratio=0.0)
    # complex_ratio 0.5: n/4 conjugate pairs, positive imaginary part first (dlanv2 order)
    assert int((ki != 0).sum()) == 2 * int(0.5 * n / 2)
    idx = np.nonzero(ki)[0]
    assert np.all(ki[idx[::2]] > 0) and np.all(ki[idx[1::2]] == -ki[idx[::2]]) and np.all(kr[idx[::2]] == kr[idx[1::2]])
    assert np.abs(kr).max() <= n and np.all(kb == 1.0)
    # a similarity transformation of the generating form: same trace
    assert abs(np.trace(A[:n]) - kr.sum()) < 1e-9 * n * n
    wr, wi = lapack_schur_eigenvalues(A, n)
    hook = O.known_eigenvalues_check((wr, wi, np.ones(n)), (kr, ki, kb))
    # (a warning -- above 1e4 u RELATIVE to the eigenvalue -- is what an eigenvalue close to the origin
    # earns with any solver; the hook passes with warnings, it fails above 1e6 u)
    assert hook["failures"] == 0 and hook["warnings"] <= 3 and hook["max_u"] < 1e6


def test_known_generator_default_ratios_zero_cluster_fails_for_lapack_too():
    """the reason tests/test_gpu_testdriver.py treats the prescribed zeros separately"""
    n = 200
    A, _, kr, ki, kb = O.known_pencil(n)
    zeros = int(((kr == 0) & (ki == 0)).sum())
    assert zeros >= 2
    wr, wi = lapack_schur_eigenvalues(A, n)
    hook = O.known_eigenvalues_check((wr, wi, np.ones(n)), (kr, ki, kb))
    assert hook["failures"] == zeros and hook["warnings"] <= 3


def test_known_generator_generalized_with_lapack_qz():
    n = 200
    A, B, kr, ki, kb = O.known_pencil(n, generalized=True, zero_ratio=0.0)
    ninf = int((kb == 0).sum())
    assert ninf >= 1
    AA, BB, _, _ = sl.qz(A[:n], B[:n], output="real")
    S = np.zeros_like(A); S[:n] = AA
    T = np.zeros_like(A); T[:n] = BB
    ar, ai, be = O.gep_extract_eigenvalues(np.asfortranarray(S), np.asfortranarray(T))
    assert int((be == 0).sum()) == ninf
    hook = O.known_eigenvalues_check((ar, ai, be), (kr, ki, kb))
    assert hook["failures"] == 0 and hook["warnings"] <= 3


def test_random_schur_input_and_decouple():
    n = 120
    H, Q, B, Z = O.schur_random_input(n, decouple=3)
    assert B is None and Z is None
    assert O.count_below_subdiagonal(H) == 0
    assert int((np.diag(H[:n], -1) == 0).sum()) == 3
    assert O.orthogonality_u(Q) < 10 and np.allclose(Q[:n], Q[:n].T)      # a Householder matrix
    # same LCG stream as the plain random Hessenberg matrix
    assert np.array_equal(np.triu(H[:n]), np.triu(O.random_hessenberg(n)[:n]))
    H2, Q2, B2, Z2 = O.schur_random_input(n, generalized=True, decouple=2, set_to_inf=4)
    assert int((np.diag(H2[:n], -1) == 0).sum()) == 2 and int((np.diag(B2[:n]) == 0).sum()) == 4
    assert O.count_below_diagonal(B2) == 0 and O.orthogonality_u(Z2) < 10


def test_eigenvalues_hook_positions_and_thresholds():
    n = 6
    r = np.array([1.0, 2.0, 0.0, 4.0, 5.0, 6.0]); i = np.zeros(n); b = np.ones(n)
    ok = O.eigenvalues_check((r, i, b), (r, i, b))
    assert ok["failures"] == 0 and ok["warnings"] == 0 and ok["max_u"] == 0.0
    r2 = r.copy(); r2[1] *= 1.0 + 2e3 * 2.0 ** -52; r2[4] *= 1.0 + 2e4 * 2.0 ** -52
    bad = O.eigenvalues_check((r, i, b), (r2, i, b))
    assert bad["warnings"] == 1 and bad["failures"] == 1
    b2 = b.copy(); b2[3] = 0.0                  # infinite where the solver returned a finite one
    assert O.eigenvalues_check((r, i, b2), (r, i, b))["failures"] == 1
