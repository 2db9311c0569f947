"""GPU tests of starneig_SEP_SM_ReorderSchur / starneig_SEP_SM_Reduce with a predicate (SURVEY 8f
rows f1, f3; reference reorder/interface.c:210-263, common/combined.c:46-98, the sequence of
examples/sep_sm_full_chain.c:88-121), through the C-ABI.  A reordered Schur form is not unique;
parity is on the reference's own acceptance checks: Schur form (hooks.c:535-714), residual and
orthogonality (checks.c:180-208), the selected eigenvalues lead the diagonal
(test/reorder: the `reordering` hook), returned eigenvalues consistent with the diagonal blocks."""
import numpy as np
import pytest

import oracle as O
from helpers import U, WARN_U, to_host

pytestmark = pytest.mark.gpu


def schur_of_lcg(node, n):
    A0 = O.random_fullpos(n)
    S = A0.copy(order="F"); Q = O.identity(n)
    real = np.zeros(n); imag = np.zeros(n)
    assert node.SEP_SM_Reduce(n, S, S.shape[0], Q, Q.shape[0], real, imag) == 0
    return A0, S, Q, real, imag


def check_reordered(A0, S, Q, real, imag, sel_in_eigs, sel_out, n):
    assert O.check_schur_form(S) == 0
    assert O.residual_u(Q, S, A0) < WARN_U and O.orthogonality_u(Q) < WARN_U
    k = len(sel_in_eigs)
    assert np.array_equal(sel_out, (np.arange(n) < k).astype(np.int32))
    er, ei = O.extract_eigenvalues(S)
    assert O.match_eigenvalues(real + 1j * imag, er + 1j * ei) < 1e3
    lead = real[:k] + 1j * imag[:k]
    scale = np.abs(real + 1j * imag).max()
    # the selected multiset leads the diagonal (each matched within 1e5 u of the spectral radius)
    rest = list(sel_in_eigs)
    for z in lead:
        d = np.abs(np.array(rest) - z)
        j = int(np.argmin(d))
        assert d[j] <= 1e5 * U * scale, (z, d[j] / (U * scale))
        rest.pop(j)


def threshold_in_a_gap(ev0, frac):
    """a real-part threshold near the (1 - frac) quantile that sits in the middle of the widest gap
    between neighbouring real parts there, so that two solvers agree on which eigenvalues lie
    above it"""
    re = np.sort(np.unique(np.round(ev0.real, 12)))
    c = int(len(re) * (1.0 - frac))
    lo, hi = max(1, c - 8), min(len(re) - 1, c + 8)
    j = lo + int(np.argmax(re[lo:hi] - re[lo - 1:hi - 1]))
    return 0.5 * (re[j - 1] + re[j])


@pytest.mark.parametrize("n,frac", [(50, 0.5), (300, 0.3), (1000, 0.5), (2000, 0.5)])
def test_select_and_reorder_after_schur(node, n, frac):
    """The selection is judged against an INDEPENDENT eigen-solve: the eigenvalues of the leading
    k x k block of the reordered form (numpy / LAPACK on that block) must be those members of
    numpy.linalg.eigvals(A0) that satisfy the predicate -- same count, matched within 1e4 u (the
    reference's `eigenvalues` fail threshold, test/common/hooks.c:787)."""
    A0, S, Q, real, imag = schur_of_lcg(node, n)
    ev0 = np.linalg.eigvals(A0[:n])
    thr = threshold_in_a_gap(ev0, frac)
    rc, sel, cnt = node.SEP_SM_Select(n, S, S.shape[0], lambda re, im: re > thr)
    assert rc == 0 and cnt > 0
    members = ev0[ev0.real > thr]
    assert cnt == len(members)
    eigs = [real[i] + 1j * imag[i] for i in range(n) if sel[i]]
    assert len(eigs) == cnt
    r2 = np.zeros(n); i2 = np.zeros(n)
    assert node.SEP_SM_ReorderSchur(n, sel, S, S.shape[0], Q, Q.shape[0], r2, i2) == 0
    check_reordered(A0, S, Q, r2, i2, eigs, sel, n)
    lead = np.linalg.eigvals(S[:cnt, :cnt])
    assert O.match_eigenvalues(lead, members) < 1e4
    assert O.match_eigenvalues(np.linalg.eigvals(S[cnt:n, cnt:n]), ev0[ev0.real <= thr]) < 1e4


def test_reduce_with_predicate_runs_the_example_chain(node):
    """examples/sep_sm_full_chain.c:88-121: Reduce = Hessenberg + Schur + Select + ReorderSchur"""
    n = 700
    A0 = O.random_fullpos(n)
    A = A0.copy(order="F"); Q = O.identity(n)
    real = np.zeros(n); imag = np.zeros(n)
    rc, sel, cnt = node.SEP_SM_Reduce(n, A, A.shape[0], Q, Q.shape[0], real, imag,
                                      predicate=lambda re, im: re > 0.0)
    assert rc == 0 and cnt > 0
    assert O.check_schur_form(A) == 0
    assert O.residual_u(Q, A, A0) < WARN_U and O.orthogonality_u(Q) < WARN_U
    assert np.all(real[:cnt] > 0.0) and np.all(real[cnt:] <= 0.0)
    assert np.array_equal(sel, (np.arange(n) < cnt).astype(np.int32))
    ev0 = np.linalg.eigvals(A0[:n])
    assert cnt == int((ev0.real > 0.0).sum())
    assert O.match_eigenvalues(np.linalg.eigvals(A[:cnt, :cnt]), ev0[ev0.real > 0.0]) < 1e4


@pytest.mark.parametrize("which", ["none", "all", "leading", "last_one", "half_of_a_pair"])
def test_reorder_edge_selections(node, which):
    n = 200
    A0, S, Q, real, imag = schur_of_lcg(node, n)
    sel = np.zeros(n, dtype=np.int32)
    pair = next(i for i in range(n - 1) if imag[i] != 0.0)
    if which == "all":
        sel[:] = 1
    elif which == "leading":
        k = 50 + (1 if imag[49] > 0 else 0)
        sel[:k] = 1
    elif which == "last_one":
        sel[n - 1] = 1
        if imag[n - 1] != 0.0:
            sel[n - 2] = 1
    elif which == "half_of_a_pair":
        sel[pair + 1] = 1                       # only the second row of a 2x2 block is marked
    S0 = S.copy(order="F")
    eigs = [real[i] + 1j * imag[i] for i in range(n) if sel[i]]
    if which == "half_of_a_pair":
        eigs = [real[pair] + 1j * imag[pair], real[pair + 1] + 1j * imag[pair + 1]]
    r2 = np.zeros(n); i2 = np.zeros(n)
    assert node.SEP_SM_ReorderSchur(n, sel, S, S.shape[0], Q, Q.shape[0], r2, i2) == 0
    check_reordered(A0, S, Q, r2, i2, eigs, sel, n)
    if which in ("none", "all", "leading"):
        assert np.array_equal(S, S0)            # nothing to move: the form is untouched


def test_reorder_conf_and_argument_checks(node):
    n = 64
    A0, S, Q, real, imag = schur_of_lcg(node, n)
    ld = S.shape[0]
    sel = np.zeros(n, dtype=np.int32); sel[n // 2:] = 1
    if imag[n // 2 - 1] > 0:
        sel[n // 2 - 1] = 1
    L = node.lib.load()
    v = real.ctypes.data
    assert L.starneig_SEP_SM_ReorderSchur(0, sel.ctypes.data, S.ctypes.data, ld, Q.ctypes.data, ld, v, v) == -1
    assert L.starneig_SEP_SM_ReorderSchur(n, None, S.ctypes.data, ld, Q.ctypes.data, ld, v, v) == -2
    assert L.starneig_SEP_SM_ReorderSchur(n, sel.ctypes.data, None, ld, Q.ctypes.data, ld, v, v) == -3
    assert L.starneig_SEP_SM_ReorderSchur(n, sel.ctypes.data, S.ctypes.data, n - 1, Q.ctypes.data, ld, v, v) == -4
    assert L.starneig_SEP_SM_ReorderSchur(n, sel.ctypes.data, S.ctypes.data, ld, None, ld, v, v) == -5
    assert L.starneig_SEP_SM_ReorderSchur(n, sel.ctypes.data, S.ctypes.data, ld, Q.ctypes.data, n - 1, v, v) == -6
    conf = node.reorder_init_conf()
    assert (conf.plan, conf.blueprint, conf.window_size, conf.values_per_chain) == (1, 1, -1, -1)
    conf.plan = 9
    assert node.SEP_SM_ReorderSchur(n, sel.copy(), S.copy(order="F"), ld, Q.copy(order="F"), ld, None, None,
                                    conf=conf) == node.INVALID_CONFIGURATION
    conf = node.reorder_init_conf(); conf.window_size = 3
    assert node.SEP_SM_ReorderSchur(n, sel.copy(), S.copy(order="F"), ld, Q.copy(order="F"), ld, None, None,
                                    conf=conf) == node.INVALID_CONFIGURATION
    # small windows and short chains give the same invariants
    conf = node.reorder_init_conf(); conf.window_size = 16; conf.values_per_chain = 3
    eigs = [real[i] + 1j * imag[i] for i in range(n) if sel[i]]
    r2 = np.zeros(n); i2 = np.zeros(n)
    assert node.SEP_SM_ReorderSchur(n, sel, S, ld, Q, ld, r2, i2, conf=conf) == 0
    check_reordered(A0, S, Q, r2, i2, eigs, sel, n)


def test_device_resident_reorder_n5000(node):
    """device-pointer twin on a device-resident chain; 30 % of the spectrum selected"""
    import torch
    n = 5000
    tA0 = node.device_matrix(n)
    assert node.lcg_fill_device(tA0, n, n, seed=2019, mode=0) == 0
    tS = tA0.clone(); tQ = node.device_matrix(n)
    node.set_matrix_device(tQ, n, n, 0.0, 1.0)
    assert node.hessenberg_device(tS, tQ, n=n) == 0
    rc, real, imag, _ = node.schur_device(tS, tQ, n=n)
    assert rc == 0
    thr = np.quantile(real, 0.7)
    sel = (real > thr).astype(np.int32)
    for i in range(n - 1):                      # whole blocks
        if imag[i] > 0.0:
            sel[i] = sel[i + 1] = max(sel[i], sel[i + 1])
    k = int(sel.sum())
    eigs = sorted(real[sel == 1])
    rc, r2, i2, st = node.reorder_schur_device(tS, tQ, sel, n=n)
    torch.cuda.synchronize()
    assert rc == 0 and st["windows"] > 0
    rc, chk = node.check_device(tQ, tS, tA0, n=n)
    assert rc == 0 and chk["residual_u"] < WARN_U and chk["orthogonality_u"] < WARN_U
    assert chk["below_subdiagonal"] == 0
    assert O.check_schur_form(to_host(tS)) == 0
    assert np.array_equal(sel, (np.arange(n) < k).astype(np.int32))
    assert np.all(r2[:k] > thr - 1e-8) and np.all(r2[k:] <= thr + 1e-8)
    assert np.abs(np.array(sorted(r2[:k])) - np.array(eigs)).max() <= 1e-7 * np.abs(real).max()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_selections_many_windows_per_round(node, seed):
    # scattered selections: every round holds many disjoint windows (reordered by several host
    # threads), 2x2 blocks land on window boundaries
    n = 1500
    A0, S, Q, real, imag = schur_of_lcg(node, n)
    rng = np.random.default_rng(seed)
    sel = (rng.random(n) < 0.4).astype(np.int32)
    for i in range(n - 1):
        if imag[i] > 0.0:
            sel[i] = sel[i + 1] = max(sel[i], sel[i + 1])
    eigs = [real[i] + 1j * imag[i] for i in range(n) if sel[i]]
    r2 = np.zeros(n); i2 = np.zeros(n)
    assert node.SEP_SM_ReorderSchur(n, sel, S, S.shape[0], Q, Q.shape[0], r2, i2) == 0
    check_reordered(A0, S, Q, r2, i2, eigs, sel, n)


def test_values_per_chain_one_with_selected_complex_pairs(node):
    """conf->values_per_chain = 1 and selected 2x2 blocks: the first selected block of a window is
    admitted whatever its size (a round loop that could not place a 2x2 block never ended)"""
    n = 300
    A0, S, Q, real, imag = schur_of_lcg(node, n)
    assert (imag > 0).any()
    sel = np.zeros(n, dtype=np.int32)
    pairs = [i for i in range(n - 1) if imag[i] > 0.0]
    for i in pairs[len(pairs) // 2:]:
        sel[i] = sel[i + 1] = 1
    sel[n - 1] = 1 if imag[n - 1] == 0.0 else sel[n - 1]
    eigs = [real[i] + 1j * imag[i] for i in range(n) if sel[i]]
    conf = node.reorder_init_conf(); conf.window_size = 16; conf.values_per_chain = 1
    r2 = np.zeros(n); i2 = np.zeros(n)
    assert node.SEP_SM_ReorderSchur(n, sel, S, S.shape[0], Q, Q.shape[0], r2, i2, conf=conf) == 0
    check_reordered(A0, S, Q, r2, i2, eigs, sel, n)
