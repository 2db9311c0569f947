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


def _schur_window(w, seed):
    """a w x w generalized Schur form (S, T) with its factors, from the small QZ kernel"""
    rng = np.random.RandomState(seed)
    A0 = np.asfortranarray(np.triu(rng.randn(w, w), -1)); B0 = np.asfortranarray(np.triu(rng.randn(w, w)) + 2.0 * np.eye(w))
    Sm, Tm = A0.copy(order="F"), B0.copy(order="F")
    Q = np.asfortranarray(np.eye(w)); Z = np.asfortranarray(np.eye(w))
    ar = np.zeros(w); ai = np.zeros(w); be = np.zeros(w)
    assert lib().sn_internal_gep_small_schur(w, P(Sm), w, P(Tm), w, P(Q), w, P(Z), w, P(ar), P(ai), P(be)) == 0
    return Sm, Tm


def _blocks(Sm):
    """[(first row, size)] of the diagonal blocks of a quasi-triangular matrix"""
    out, i, w = [], 0, Sm.shape[0]
    while i < w:
        bs = 2 if i + 1 < w and Sm[i + 1, i] != 0.0 else 1
        out.append((i, bs)); i += bs
    return out


def _deflate(w, S0, T0, spike0, sub, thres, carried):
    L = S.lib.load_test_hooks()
    ip = C.POINTER(C.c_int)
    L.sn_internal_gep_deflate_window.argtypes = [C.c_int] + [dp, C.c_int] * 4 + [dp, C.c_double, C.c_double, C.c_int, ip]
    L.sn_internal_gep_deflate_window.restype = C.c_int
    Sm, Tm = S0.copy(order="F"), T0.copy(order="F")
    Q = np.asfortranarray(np.eye(w)); Z = np.asfortranarray(np.eye(w))
    spike = spike0.copy()
    und = C.c_int(-1)
    rc = L.sn_internal_gep_deflate_window(w, P(Sm), w, P(Tm), w, P(Q), w, P(Z), w, P(spike), sub, thres, carried,
                                          C.byref(und))
    return rc, Sm, Tm, Q, Z, spike, und.value


@pytest.mark.parametrize("w,carried", [(40, 0), (64, 5), (96, 17)])
def test_gep_deflate_window_on_a_diagonal_pencil_is_a_permutation(w, carried):
    """On a pencil of two diagonal matrices every exchange is a signed permutation, so the outcome of
    host::gep_deflate_window (the pencil twin of deflate_window; reference schur/cpu.c:638-1006 with B != NULL)
    is known exactly: the carried rows first, in order, then the undeflatable rows in the order they were found
    (from the bottom up), then the deflatable ones; the spike entries travel with their rows."""
    rng = np.random.RandomState(w)
    d = np.arange(1, w + 1) + rng.rand(w)
    S0 = np.asfortranarray(np.diag(d)); T0 = np.asfortranarray(np.diag(1.0 + rng.rand(w)))
    sub, thres = 1.0, 1e-8
    big = np.zeros(w, dtype=bool)
    big[rng.choice(w - carried, size=(w - carried) // 3, replace=False)] = True
    big[w - carried:] = True
    spike0 = np.where(big, 0.1 + rng.rand(w), 1e-12 * rng.rand(w))
    rc, Sm, Tm, Q, Z, spike, und = _deflate(w, S0, T0, spike0, sub, thres, carried)
    assert rc == 0 and und == int(big.sum())
    check_gschur(S0, T0, Sm, Tm, Q, Z)
    unchecked_big = [i for i in range(w - carried - 1, -1, -1) if big[i]]          # found from the bottom up
    order = list(range(w - carried, w)) + unchecked_big
    lam0 = d / np.diag(T0)
    lam = np.diag(Sm) / np.diag(Tm)
    assert np.allclose(lam[:und], lam0[order], rtol=1e-13)
    assert np.allclose(np.abs(spike[:und]), np.abs(spike0[order]), rtol=1e-13)
    assert np.all(np.abs(spike[und:]) < thres)


@pytest.mark.parametrize("w,carried_blocks,seed", [(40, 0, 1), (64, 3, 2), (96, 5, 3)])
def test_gep_deflate_window_of_the_blocked_aed(w, carried_blocks, seed):
    """the same on a general window with 2 x 2 blocks: the decomposition stays intact, the spike follows the
    left factor, the carried blocks sit at the top in their order, everything below `undeflated` is below the
    threshold and no eigenvalue is lost"""
    S0, T0 = _schur_window(w, seed)
    blocks = _blocks(S0)
    carried = sum(bs for _, bs in blocks[len(blocks) - carried_blocks:]) if carried_blocks else 0
    rng = np.random.RandomState(100 + seed)
    sub, thres = 1.0, 1e-8
    spike0 = 1e-12 * rng.rand(w)
    for i, bs in blocks[:3]:                                # the top three blocks cannot be deflated
        spike0[i:i + bs] = 0.1 + rng.rand(bs)
    if carried:
        spike0[w - carried:] = 0.5
    rc, Sm, Tm, Q, Z, spike, und = _deflate(w, S0, T0, spike0, sub, thres, carried)
    assert rc == 0
    check_gschur(S0, T0, Sm, Tm, Q, Z)
    assert np.allclose(spike, spike0 @ Q, rtol=0, atol=1e-13)
    assert carried <= und <= w
    assert np.all(np.abs(spike[und:]) < thres)
    if carried:
        want = sl.eigvals(S0[w - carried:, w - carried:], T0[w - carried:, w - carried:])
        got = sl.eigvals(Sm[:carried, :carried], Tm[:carried, :carried])
        assert O.match_eigenvalues(np.asarray(got), np.asarray(want)) < 1e7
    e0 = sl.eigvals(S0, T0); e1 = sl.eigvals(Sm, Tm)
    assert O.match_eigenvalues(np.asarray(e1), np.asarray(e0)) < 1e7


@pytest.mark.parametrize("w,seed", [(48, 4), (128, 5)])
def test_gep_reorder_window_moves_the_marked_blocks_to_the_top(w, seed):
    L = S.lib.load_test_hooks()
    ip = C.POINTER(C.c_int)
    L.sn_internal_gep_reorder_window.argtypes = [C.c_int] + [dp, C.c_int] * 4 + [ip, ip]
    L.sn_internal_gep_reorder_window.restype = C.c_int
    S0, T0 = _schur_window(w, seed)
    blocks = _blocks(S0)
    sel = np.zeros(w, dtype=np.int32)
    marked = blocks[len(blocks) // 2:][::2]                 # every other block of the lower half
    rows = 0
    want = []
    for i, bs in marked:
        sel[i:i + bs] = 1; rows += bs
        want.extend(sl.eigvals(S0[i:i + bs, i:i + bs], T0[i:i + bs, i:i + bs]))
    Sm, Tm = S0.copy(order="F"), T0.copy(order="F")
    Q = np.asfortranarray(np.eye(w)); Z = np.asfortranarray(np.eye(w))
    failed = C.c_int(0)
    placed = L.sn_internal_gep_reorder_window(w, P(Sm), w, P(Tm), w, P(Q), w, P(Z), w,
                                              sel.ctypes.data_as(ip), C.byref(failed))
    assert failed.value == 0 and placed == rows
    check_gschur(S0, T0, Sm, Tm, Q, Z)
    assert np.array_equal(sel, (np.arange(w) < rows).astype(np.int32))
    got = sl.eigvals(Sm[:rows, :rows], Tm[:rows, :rows])
    assert O.match_eigenvalues(np.asarray(got), np.asarray(want)) < 1e7
