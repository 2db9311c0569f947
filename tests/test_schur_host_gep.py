"""CPU: host-side kernels of the generalized (QZ) path through internal hooks, against
scipy/LAPACK (generalized eigenvalues of the same pencil) and the invariants of a generalized
Schur decomposition Q S Z^T = A, Q T Z^T = B."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sl

import oracle as O
import starneig_amd as S
from helpers import U

dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


def lib():
    L = S.lib.load_test_hooks()
    L.sn_internal_gep_small_schur.argtypes = [C.c_int] + [dp, C.c_int] * 4 + [dp, dp, dp]
    L.sn_internal_gep_ht_reduce.argtypes = [C.c_int, C.c_int, C.c_int] + [dp, C.c_int] * 4
    L.sn_internal_gep_aed_window.argtypes = [C.c_int] + [dp, C.c_int] * 4 + [C.c_double, C.c_double, dp, dp, dp,
                                                                              C.POINTER(C.c_int)]
    return L


def random_ht_pencil(n, seed=2019):
    """test driver's generalized Schur input: random Hessenberg + random upper triangular
    (test/common/init.c:122-138,159-175), LCG stream"""
    O.lib().oracle_init_prand(seed)
    A = np.zeros((n, n), order="F"); B = np.zeros((n, n), order="F")
    O.lib().oracle_fill_random_hessenberg(n, P(A), n)
    rng = np.random.RandomState(seed)
    B[:] = np.triu(rng.uniform(-1, 1, (n, n)))
    return A, B


def check_gschur(A0, B0, Sm, Tm, Q, Z, tol=500):
    n = A0.shape[0]
    assert np.all(np.tril(Sm, -2) == 0.0) and np.all(np.tril(Tm, -1) == 0.0)
    sub = np.diag(Sm, -1)
    assert not np.any((sub[:-1] != 0) & (sub[1:] != 0))
    for i in np.nonzero(sub)[0]:                         # standardised 2x2 blocks (hooks.c:571-620)
        assert Tm[i, i + 1] == 0.0 and Tm[i, i] > 0 and Tm[i + 1, i + 1] > 0
    assert np.linalg.norm(Q @ Sm @ Z.T - A0) <= tol * U * np.linalg.norm(A0)
    assert np.linalg.norm(Q @ Tm @ Z.T - B0) <= tol * U * np.linalg.norm(B0)
    assert np.linalg.norm(Q @ Q.T - np.eye(n)) <= tol * U * np.sqrt(n)
    assert np.linalg.norm(Z @ Z.T - np.eye(n)) <= tol * U * np.sqrt(n)


def eig_match(ar, ai, be, A0, B0):
    ev = (ar + 1j * ai) / be
    ref = sl.eigvals(A0, B0)
    return O.match_eigenvalues(ev, ref)


@pytest.mark.parametrize("n", [2, 3, 4, 11, 40, 120])
def test_small_qz_against_scipy(n):
    A0, B0 = random_ht_pencil(n)
    Sm, Tm = A0.copy(order="F"), B0.copy(order="F")
    Q = np.asfortranarray(np.eye(n)); Z = np.asfortranarray(np.eye(n))
    ar = np.zeros(n); ai = np.zeros(n); be = np.zeros(n)
    assert lib().sn_internal_gep_small_schur(n, P(Sm), n, P(Tm), n, P(Q), n, P(Z), n, P(ar), P(ai), P(be)) == 0
    check_gschur(A0, B0, Sm, Tm, Q, Z)
    assert np.all(be > 0)
    assert eig_match(ar, ai, be, A0, B0) < 1e7           # pencils with tiny |T_ii| are ill-conditioned
    i = 0
    while i < n:
        if ai[i] != 0:
            assert ai[i] > 0 and ai[i + 1] == -ai[i]; i += 2
        else:
            i += 1


def test_ht_reduce_restores_structure():
    n = 30
    rng = np.random.RandomState(1)
    A0 = np.asfortranarray(rng.randn(n, n)); B0 = np.asfortranarray(np.triu(rng.randn(n, n)))
    A, B = A0.copy(order="F"), B0.copy(order="F")
    Q = np.asfortranarray(np.eye(n)); Z = np.asfortranarray(np.eye(n))
    lib().sn_internal_gep_ht_reduce(n, 0, n - 1, P(A), n, P(B), n, P(Q), n, P(Z), n)
    assert np.abs(np.tril(A, -2)).max() == 0.0 and np.abs(np.tril(B, -1)).max() == 0.0
    assert np.linalg.norm(Q @ A @ Z.T - A0) <= 200 * U * np.linalg.norm(A0)
    assert np.linalg.norm(Q @ B @ Z.T - B0) <= 200 * U * np.linalg.norm(B0)


@pytest.mark.parametrize("nw,sub", [(40, 1e-6), (90, 1e-2)])
def test_gep_aed_window_invariants(nw, sub):
    A0, B0 = random_ht_pencil(nw, seed=7)
    thres = U * np.linalg.norm(A0) * 1e4
    A, B = A0.copy(order="F"), B0.copy(order="F")
    Q = np.zeros((nw, nw), order="F"); Z = np.zeros((nw, nw), order="F")
    spike = np.zeros(nw); sr = np.zeros(nw); si = np.zeros(nw); out = (C.c_int * 3)()
    lib().sn_internal_gep_aed_window(nw, P(A), nw, P(B), nw, P(Q), nw, P(Z), nw, sub, thres,
                                     P(spike), P(sr), P(si), out)
    nd, nsh, failed = out[0], out[1], out[2]
    assert failed == 0
    assert np.linalg.norm(Q @ Q.T - np.eye(nw)) <= 500 * U * np.sqrt(nw)
    assert np.linalg.norm(Z @ Z.T - np.eye(nw)) <= 500 * U * np.sqrt(nw)
    if nd == 0:
        return
    ns = nw - nd
    assert np.linalg.norm(Q @ A @ Z.T - A0) <= 1000 * U * np.linalg.norm(A0)
    assert np.linalg.norm(Q @ B @ Z.T - B0) <= 1000 * U * np.linalg.norm(B0)
    assert np.all(np.tril(A, -2) == 0.0) and np.all(np.tril(B, -1) == 0.0)
    assert np.all(A[ns:, :ns] == 0.0)
    full = sub * Q[0, :]
    assert np.all(spike[1:] == 0.0)
    assert abs(abs(spike[0]) - np.linalg.norm(full[:ns])) <= 1e3 * U * abs(sub)
    assert np.all(np.abs(full[ns:]) < thres * 1.001)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gep_move_block_up_reorders_eigenvalues(seed):
    """every block of a generalized Schur form can be bubbled to the top; the eigenvalue moves
    with it and the decomposition stays intact (LAPACK dtgexc semantics)"""
    n = 24
    A0, B0 = random_ht_pencil(n, seed=seed)
    L = lib()
    L.sn_internal_gep_move_block_up.argtypes = [C.c_int] + [dp, C.c_int] * 4 + [C.c_int, C.c_int]
    L.sn_internal_gep_move_block_up.restype = C.c_int
    Sm, Tm = A0.copy(order="F"), B0.copy(order="F")
    Q = np.asfortranarray(np.eye(n)); Z = np.asfortranarray(np.eye(n))
    ar = np.zeros(n); ai = np.zeros(n); be = np.zeros(n)
    assert L.sn_internal_gep_small_schur(n, P(Sm), n, P(Tm), n, P(Q), n, P(Z), n, P(ar), P(ai), P(be)) == 0
    ev0 = np.sort_complex((ar + 1j * ai) / be)
    moved = 0
    for trial in range(12):
        sub = np.diag(Sm, -1)
        starts = [i for i in range(n) if i == 0 or sub[i - 1] == 0.0]
        src = starts[(5 * trial + 3) % len(starts)]
        if src == 0:
            continue
        two = src + 1 < n and Sm[src + 1, src] != 0.0
        lam = (Sm[src, src] / Tm[src, src]) if not two else None
        at = L.sn_internal_gep_move_block_up(n, P(Sm), n, P(Tm), n, P(Q), n, P(Z), n, src, 0)
        check_gschur(A0, B0, Sm, Tm, Q, Z)
        if at == 0:
            moved += 1
            if lam is not None:
                assert abs(Sm[0, 0] / Tm[0, 0] - lam) <= 1e-9 * max(1.0, abs(lam))
        er = np.zeros(n); ei = np.zeros(n); eb = np.zeros(n)
        O.lib().oracle_gep_extract_eigenvalues(n, P(Sm), n, P(Tm), n, P(er), P(ei), P(eb))
        assert O.match_eigenvalues((er + 1j * ei) / eb, ev0) < 1e7
    assert moved >= 6


def test_gep_aed_window_reorders_to_deflate_more():
    """with reordering the AED deflates converged eigenvalues from anywhere in the window: a
    pencil that is already triangular with a spike that is tiny except for a few entries must
    deflate all the others"""
    nw = 30
    rng = np.random.RandomState(5)
    A0 = np.asfortranarray(np.triu(rng.uniform(-1, 1, (nw, nw))) + np.diag(np.arange(1.0, nw + 1.0)))
    B0 = np.asfortranarray(np.triu(rng.uniform(-1, 1, (nw, nw)) * 0.1) + np.eye(nw))
    A, B = A0.copy(order="F"), B0.copy(order="F")
    Q = np.zeros((nw, nw), order="F"); Z = np.zeros((nw, nw), order="F")
    spike = np.zeros(nw); sr = np.zeros(nw); si = np.zeros(nw); out = (C.c_int * 3)()
    # triangular input: Q = I after the (trivial) Schur step, so spike = sub * e_1: only the first
    # eigenvalue is coupled and all others deflate -- but it sits at the TOP, the scan from the
    # bottom passes 29 deflatable eigenvalues first
    lib().sn_internal_gep_aed_window(nw, P(A), nw, P(B), nw, P(Q), nw, P(Z), nw, 1e-3, 1e-10,
                                     P(spike), P(sr), P(si), out)
    assert out[2] == 0 and out[0] == nw - 1
    # and with a spike hitting a middle eigenvalue (rotate the pencil so that Q(0,:) has two entries)
    A, B = A0.copy(order="F"), B0.copy(order="F")
    G = np.eye(nw); c, s = np.cos(0.3), np.sin(0.3); k = 17
    # swap roles by an explicit equivalence that keeps triangularity: not possible with a plain
    # rotation, so instead check the general invariants on a random window with a loose threshold
    A0r, B0r = random_ht_pencil(60, seed=11)
    A, B = A0r.copy(order="F"), B0r.copy(order="F")
    Q = np.zeros((60, 60), order="F"); Z = np.zeros((60, 60), order="F")
    spike = np.zeros(60); sr = np.zeros(60); si = np.zeros(60)
    thres = 0.02 * 1e-3          # deflate whatever has |Q(0,i)| < 0.02
    lib().sn_internal_gep_aed_window(60, P(A), 60, P(B), 60, P(Q), 60, P(Z), 60, 1e-3, thres,
                                     P(spike), P(sr), P(si), out)
    nd = out[0]; ns = 60 - nd
    assert out[2] == 0 and nd > 0
    full = 1e-3 * Q[0, :]
    assert np.all(np.abs(full[ns:]) < thres * 1.001)
    assert np.linalg.norm(Q @ Q.T - np.eye(60)) <= 500 * U * np.sqrt(60)
    # what is dropped is exactly the deflated spike entries
    E = Q @ A @ Z.T - A0r
    assert np.linalg.norm(E) <= 1000 * U * np.linalg.norm(A0r)
    assert np.all(np.tril(A, -2) == 0.0) and np.all(np.tril(B, -1) == 0.0) and np.all(A[ns:, :ns] == 0.0)


def test_close_real_pair_is_split_and_reported_accurately():
    """A 2x2 block with two CLOSE real eigenvalues (-4423.496 / -4424.034, relative gap 1e-4), found
    by the reference's `--init known --generalized` experiment at n = 4000: the quadratic formula on
    (sum, product) lost the difference to cancellation, the block stayed unsplit and both
    eigenvalues were reported as their mean (an error of 6e-5 relative = 3e11 u).  The eigenvalues
    now come from the shifted form LAPACK dlag2 uses (reference common/math.c:148-176)."""
    L = lib()
    A0 = np.array([[-3403.648820837277, 0.32292784219672926], [-0.013751781294212357, -1382.9661188795437]])
    B0 = np.array([[0.7694533042009347, 0.0], [0.0, 0.31260067874299907]])
    A, B = np.asfortranarray(A0.copy()), np.asfortranarray(B0.copy())
    Q, Z = np.asfortranarray(np.eye(2)), np.asfortranarray(np.eye(2))
    ar, ai, be = np.zeros(2), np.zeros(2), np.zeros(2)
    assert L.sn_internal_gep_small_schur(2, P(A), 2, P(B), 2, P(Q), 2, P(Z), 2, P(ar), P(ai), P(be)) == 0
    assert A[1, 0] == 0.0 and B[1, 0] == 0.0 and not ai.any()
    ref = np.sort(sl.eigvals(A0, B0).real)
    assert np.abs(np.sort(ar / be) - ref).max() <= 100 * U * np.abs(ref).max()
    check_gschur(A0, B0, A, B, Q, Z, tol=50)


def test_returned_eigenvalues_of_2x2_blocks_are_dlag2_bit_for_bit():
    """The product returns (alpha, beta) of a 2 x 2 block as LAPACK dlag2's (wr, +-wi, scale), like the
    reference (common/math.c:148-176) -- the same bits as LAPACK itself and as the oracle's extraction, so
    that the reference's `eigenvalues` hook holds at its own thresholds (10^3 / 10^4 u, hooks.c:787-788)
    even on blocks whose discriminant nearly vanishes."""
    import os
    import oracle as O
    hooks = S.lib.load_test_hooks()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dlag2_cases.npz"))
    dp = C.POINTER(C.c_double)
    checked = 0
    for A, B, want in zip(g["A"], g["B"], g["out"]):
        if A[1, 0] == 0.0:
            continue
        a, b = np.asfortranarray(A), np.asfortranarray(B)
        ar, ai, be = np.zeros(2), np.zeros(2), np.zeros(2)
        hooks.sn_internal_gep_extract_eigenvalues(2, a.ctypes.data_as(dp), 2, b.ctypes.data_as(dp), 2,
                                                  ar.ctypes.data_as(dp), ai.ctypes.data_as(dp), be.ctypes.data_as(dp))
        s1, s2, w1, w2, wi = want
        assert (ar[0], ar[1], ai[0], ai[1], be[0], be[1]) == (w1, w2, wi, -wi, s1, s2)
        oar, oai, obe = O.gep_extract_eigenvalues(a, b)
        assert np.array_equal(oar, ar) and np.array_equal(oai, ai) and np.array_equal(obe, be)
        checked += 1
    assert checked > 800
