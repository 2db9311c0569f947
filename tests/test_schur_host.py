"""CPU suite, part 3: the host-side sequential kernels of the Schur path (small Schur
reduction, block reordering, shift extraction, AED window) through internal hooks of the
library -- no GPU involved.  Checked against numpy/LAPACK and the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
import starneig_amd as S
from helpers import U

dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


def lib():
    L = S.lib.load_test_hooks()
    L.sn_internal_small_schur.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, dp, dp]
    L.sn_internal_move_block_up.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int]
    L.sn_internal_extract_shifts.argtypes = [C.c_int, dp, C.c_int, dp, dp]
    L.sn_internal_aed_window.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, C.c_double, C.c_double,
                                         dp, dp, dp, C.POINTER(C.c_int)]
    return L


def hess_input(n, seed=2019):
    return np.asfortranarray(O.random_hessenberg(n, seed=seed, ld=n))


def is_quasi_triangular(T):
    return O.check_schur_form(np.asfortranarray(T)) == 0


@pytest.mark.parametrize("n", [1, 2, 3, 10, 75, 200])
def test_small_schur_matches_oracle_and_numpy(n):
    H0 = hess_input(n)
    T = H0.copy(order="F"); Z = np.asfortranarray(np.eye(n))
    wr = np.zeros(n); wi = np.zeros(n)
    assert lib().sn_internal_small_schur(n, P(T), n, P(Z), n, P(wr), P(wi)) == 0
    assert is_quasi_triangular(T)
    assert np.linalg.norm(Z @ T @ Z.T - H0) <= 200 * U * np.linalg.norm(H0)
    assert np.linalg.norm(Z @ Z.T - np.eye(n)) <= 200 * U * np.sqrt(n)
    ev = np.linalg.eigvals(H0)
    assert O.match_eigenvalues(wr + 1j * wi, ev) < 1e6
    # same algorithm as the oracle: same eigenvalues to rounding
    To = H0.copy(order="F"); Zo = np.asfortranarray(np.eye(n))
    wro, wio = O.schur(To, Zo)
    assert O.match_eigenvalues(wr + 1j * wi, wro + 1j * wio) < 1e6


def test_move_block_up_preserves_similarity_and_order():
    n = 40
    H0 = hess_input(n, seed=5)
    T = H0.copy(order="F"); Z = np.asfortranarray(np.eye(n))
    wr = np.zeros(n); wi = np.zeros(n)
    lib().sn_internal_small_schur(n, P(T), n, P(Z), n, P(wr), P(wi))
    ev_before = wr + 1j * wi
    # move the last block to the top
    frm = n - 2 if T[n - 1, n - 2] != 0 else n - 1
    moving = ev_before[frm]
    at = lib().sn_internal_move_block_up(n, P(T), n, P(Z), n, frm, 0)
    assert at == 0
    assert is_quasi_triangular(T)
    assert np.linalg.norm(Z @ T @ Z.T - H0) <= 500 * U * np.linalg.norm(H0)
    wr2, wi2 = O.extract_eigenvalues(np.asfortranarray(T))
    ev_after = wr2 + 1j * wi2
    assert abs(ev_after[0].real - moving.real) <= 1e-8 * max(1, abs(moving))
    assert O.match_eigenvalues(ev_after, ev_before) < 1e8


def test_extract_shifts_order_and_pairing():
    """schur/cpu_utils.c:3522-3594: ascending |re|+|im|, zero/inf dropped, conjugates adjacent."""
    blocks = [np.array([[3.0]]), np.array([[1.0, 2.0], [-2.0, 1.0]]), np.array([[0.0]]),
              np.array([[-0.5]]), np.array([[4.0, 1.0], [-1.0, 4.0]]), np.array([[2.0]])]
    n = sum(b.shape[0] for b in blocks)
    T = np.zeros((n, n), order="F"); k = 0
    for b in blocks:
        T[k:k + b.shape[0], k:k + b.shape[0]] = b; k += b.shape[0]
    wr = np.zeros(n); wi = np.zeros(n)
    cnt = lib().sn_internal_extract_shifts(n, P(T), n, P(wr), P(wi))
    assert cnt == n - 1                                   # the zero eigenvalue is dropped
    for i in range(0, cnt - 1, 2):
        assert wi[i] == -wi[i + 1]
    mags = np.abs(wr[:cnt]) + np.abs(wi[:cnt])
    assert mags[0] == 0.5
    assert np.allclose(sorted(mags), [0.5, 2.0, 3.0, 3.0, 3.0, 5.0, 5.0], rtol=1e-14)


@pytest.mark.parametrize("nw,sub_scale", [(30, 1e-3), (60, 1e-1), (100, 1e-8)])
def test_aed_window_invariants(nw, sub_scale):
    """After AED: T = [Hessenberg | *; 0 | Schur], Z orthogonal, Z T Z^T + spike consistent
    with the original window, deflated spike entries were below the threshold."""
    W0 = hess_input(nw, seed=11)
    sub = sub_scale
    thres = U * np.linalg.norm(W0) * 50           # generous threshold so something deflates
    T = W0.copy(order="F"); Z = np.zeros((nw, nw), order="F")
    spike = np.zeros(nw); sr = np.zeros(nw); si = np.zeros(nw)
    out = (C.c_int * 3)()
    lib().sn_internal_aed_window(nw, P(T), nw, P(Z), nw, sub, thres, P(spike), P(sr), P(si), out)
    nd, nshift, failed = out[0], out[1], out[2]
    assert failed == 0
    assert np.linalg.norm(Z @ Z.T - np.eye(nw)) <= 500 * U * np.sqrt(nw)
    if nd == 0:
        return
    ns = nw - nd
    # similarity on the window
    assert np.linalg.norm(Z @ T @ Z.T - W0) <= 1000 * U * np.linalg.norm(W0)
    # structure: leading ns x ns Hessenberg, trailing nd x nd quasi-triangular, zero coupling
    assert np.all(np.tril(T[:ns, :ns], -2) == 0.0)
    assert np.all(T[ns:, :ns] == 0.0)
    assert is_quasi_triangular(T[ns:, ns:])
    # the spike: sub * e1^T Z, compressed to one entry on the Hessenberg part, zero on the rest
    full = sub * Z[0, :]
    assert np.all(spike[1:] == 0.0)
    assert abs(abs(spike[0]) - np.linalg.norm(full[:ns])) <= 100 * U * abs(sub)
    assert np.all(np.abs(full[ns:]) < thres)
    assert nshift <= nw


@pytest.mark.parametrize("n,seed", [(12, 1), (60, 2), (128, 3)])
def test_reorder_window_moves_selected_blocks_to_the_top(n, seed):
    """host::reorder_window (the window kernel of starneig_SEP_SM_ReorderSchur; reference
    reorder/cpu.c, LAPACK dtrsen semantics): selected blocks at the top in their original order,
    similarity and orthogonality preserved, marks follow the rows, against scipy's sorted Schur."""
    L = S.lib.load_test_hooks()
    ip = C.POINTER(C.c_int)
    L.sn_internal_reorder_window.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, ip, ip]
    H0 = hess_input(n, seed=seed)
    T = H0.copy(order="F"); Z = np.asfortranarray(np.eye(n))
    wr = np.zeros(n); wi = np.zeros(n)
    assert lib().sn_internal_small_schur(n, P(T), n, P(Z), n, P(wr), P(wi)) == 0
    T0 = T.copy(order="F")
    sel = (wr > np.median(wr)).astype(np.int32)
    # one mark of a pair is enough to select the block
    for i in range(n - 1):
        if T[i + 1, i] != 0.0 and sel[i]:
            sel[i + 1] = 0
    want = [(wr[i], wi[i]) for i in range(n) if wr[i] > np.median(wr)]
    Zw = np.asfortranarray(np.eye(n)); failed = C.c_int(0)
    placed = L.sn_internal_reorder_window(n, P(T), n, P(Zw), n, sel.ctypes.data_as(ip), C.byref(failed))
    assert failed.value == 0 and placed == len(want)
    assert is_quasi_triangular(T)
    assert np.linalg.norm(Zw @ T @ Zw.T - T0) <= 500 * U * np.linalg.norm(T0)
    assert np.linalg.norm(Zw @ Zw.T - np.eye(n)) <= 200 * U * np.sqrt(n)
    assert np.array_equal(sel, (np.arange(n) < placed).astype(np.int32))
    er, ei = O.extract_eigenvalues(np.asfortranarray(T))
    # the selected eigenvalues, in their original order, now lead the diagonal
    got = np.array(er[:placed]) + 1j * np.array(ei[:placed])
    exp = np.array([a for a, b in want]) + 1j * np.array([b for a, b in want])
    assert np.abs(got - exp).max() <= 1e5 * U * np.abs(exp).max()
    # and nothing selected is left below
    assert np.all(np.array(er[placed:]) <= np.median(wr) + 1e5 * U * np.abs(wr).max())


@pytest.mark.parametrize("n", [48, 97, 160, 300])
def test_helper_threads_reproduce_the_serial_kernels_bit_for_bit(n):
    """small_schur / aed_window with Z, the far columns of the active block and the columns right of it
    handed to the helper team (schur_host_team.h): every entry sees the same operations in the same
    order, so T, Z and the eigenvalues equal the serial results exactly."""
    L = lib()
    L.sn_internal_helper_session.argtypes = [C.c_int]
    H0 = hess_input(n, seed=11)
    out = []
    for on in (0, 1, 1, 0):
        L.sn_internal_helper_session(on)
        T = H0.copy(order="F"); Z = np.asfortranarray(np.eye(n))
        wr = np.zeros(n); wi = np.zeros(n)
        assert L.sn_internal_small_schur(n, P(T), n, P(Z), n, P(wr), P(wi)) == 0
        out.append((T, Z, wr, wi))
    L.sn_internal_helper_session(0)
    for T, Z, wr, wi in out[1:]:
        assert np.array_equal(T, out[0][0]) and np.array_equal(Z, out[0][1])
        assert np.array_equal(wr, out[0][2]) and np.array_equal(wi, out[0][3])
    # the AED window kernel on top of it
    res = []
    for on in (0, 1):
        L.sn_internal_helper_session(on)
        T = H0.copy(order="F"); Z = np.asfortranarray(np.zeros((n, n)))
        spike = np.zeros(n); sr = np.zeros(n); si = np.zeros(n); o3 = (C.c_int * 3)()
        L.sn_internal_aed_window(n, P(T), n, P(Z), n, 0.37, 1e-13 * np.linalg.norm(H0), P(spike), P(sr), P(si), o3)
        res.append((T, Z, spike, tuple(o3)))
    L.sn_internal_helper_session(0)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], res[1][2]) and res[0][3] == res[1][3]


@pytest.mark.parametrize("which", [0, 1])
def test_aed_window_on_windows_captured_from_the_baseline_reduction(which):
    """Two AED windows as they arrive during the n = 20000 reduction (tests/golden/make_aed_windows.py):
    unlike a random Hessenberg matrix they deflate about half of their eigenvalues, so the swaps of the
    deflation phase and the re-reduction to Hessenberg form carry real work.  The invariants of
    test_aed_window_invariants, and the serial kernel against the helper team (schur_host_team.h: far
    columns, rows above, Z and the re-Hessenberg updates on other threads) bit for bit."""
    L = lib()
    L.sn_internal_helper_session.argtypes = [C.c_int]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "aed_windows_lcg20000.npz"))
    W0 = np.asfortranarray(g["windows"][which]); sub = float(g["subs"][which]); thres = float(g["thres"][which])
    nw = W0.shape[0]
    res = []
    for helpers in (0, 5, 3):
        L.sn_internal_helper_session(helpers)
        ld = nw + 8                                   # the driver's leading dimension
        Tb = np.zeros((ld, nw), order="F"); Tb[:nw] = W0
        Zb = np.zeros((ld, nw), order="F")
        spike = np.zeros(nw); sr = np.zeros(nw); si = np.zeros(nw); out = (C.c_int * 3)()
        L.sn_internal_aed_window(nw, P(Tb), ld, P(Zb), ld, sub, thres, P(spike), P(sr), P(si), out)
        L.sn_internal_helper_session(0)
        res.append((Tb[:nw].copy(), Zb[:nw].copy(), spike, tuple(out)))
    T, Z, spike, (nd, nshift, failed) = res[0]
    assert failed == 0 and 40 <= nd <= nw - 40
    ns = nw - nd
    assert np.linalg.norm(Z @ Z.T - np.eye(nw)) <= 500 * U * np.sqrt(nw)
    assert np.linalg.norm(Z @ T @ Z.T - W0) <= 1000 * U * np.linalg.norm(W0)
    assert np.all(np.tril(T[:ns, :ns], -2) == 0.0) and np.all(T[ns:, :ns] == 0.0)
    assert is_quasi_triangular(T[ns:, ns:])
    full = sub * Z[0, :]
    assert np.all(spike[1:] == 0.0) and np.all(np.abs(full[ns:]) < thres)
    assert abs(abs(spike[0]) - np.linalg.norm(full[:ns])) <= 100 * U * abs(sub)
    for Tt, Zt, st, ot in res[1:]:
        assert ot == res[0][3] and np.array_equal(st, spike)
        assert np.array_equal(Tt, T) and np.array_equal(Zt, Z)


def test_reorder_window_rejected_swap_keeps_a_valid_decomposition():
    """A selected 2x2 block below an unselected 2x2 block with (almost) the same eigenvalue pair:
    the Sylvester equation of the exchange is (nearly) singular, dlaexc's acceptance test rejects
    the swap (reorder/cpu.c -> STARNEIG_PARTIAL_REORDERING).  The kernel must then report `failed`,
    leave a valid similarity behind and keep the marks on the rows where the blocks now are."""
    L = S.lib.load_test_hooks()
    ip = C.POINTER(C.c_int)
    L.sn_internal_reorder_window.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, ip, ip]
    n = 8
    rng = np.random.default_rng(3)
    T = np.asfortranarray(np.triu(rng.standard_normal((n, n))))
    blk = np.array([[1.0, 2.0], [-0.5, 1.0]])                   # standardised 2x2 block, eigenvalues 1 +- i
    T[0:2, 0:2] = [[3.0, 0.1], [0.0, 4.0]]                      # two real eigenvalues, unselected
    T[2:4, 2:4] = blk                                           # unselected pair
    T[4:6, 4:6] = blk * (1.0 + 1e-15)                           # selected pair: the same eigenvalues up to rounding
    T[6:8, 6:8] = [[-2.0, 0.3], [0.0, -3.0]]
    for i in (0, 1, 3, 5, 6):                                   # clean the sub-diagonal outside the two pairs
        if i + 1 < n and i not in (2, 4):
            T[i + 1, i] = 0.0
    T[3, 2] = blk[1, 0]; T[5, 4] = blk[1, 0] * (1.0 + 1e-15)
    T0 = T.copy(order="F")
    sel = np.zeros(n, dtype=np.int32); sel[4] = sel[5] = 1
    Z = np.asfortranarray(np.eye(n)); failed = C.c_int(0)
    placed = L.sn_internal_reorder_window(n, P(T), n, P(Z), n, sel.ctypes.data_as(ip), C.byref(failed))
    assert is_quasi_triangular(T)
    assert np.linalg.norm(Z @ T @ Z.T - T0) <= 500 * U * np.linalg.norm(T0)
    assert np.linalg.norm(Z @ Z.T - np.eye(n)) <= 200 * U * np.sqrt(n)
    if failed.value:
        # nothing reached the top; the selected pair still carries its marks, wherever it stopped
        assert placed == 0 and sel.sum() == 2
        i = int(np.argmax(sel))
        assert sel[i + 1] == 1 and T[i + 1, i] != 0.0
    else:
        assert placed == 2 and np.array_equal(sel, (np.arange(n) < 2).astype(np.int32))
