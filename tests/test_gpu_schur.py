"""GPU parity tests of the Schur leg: the HIP multi-shift QR path, called through the C-ABI,
against the CPU oracle (double-shift QR restatement) on the same seeded inputs, plus the
reference's own acceptance checks (test/common/hooks.c:535-714 Schur form, :891-991 eigenvalue
self-consistency, checks.c:180-208 residuals).  Schur forms are not unique, so parity is on
invariants: eigenvalue multisets, structure, residual, orthogonality.

Tolerances (units of u = 2^-52, relative to max(|lambda|, 1e-3 max|lambda|)):
  eigenvalues vs oracle: 1e4 u for the dense LCG matrices (the reference's warn threshold),
  random Hessenberg input: those spectra are so ill-conditioned that two backward-stable
  solvers disagree by 1e6..1e9 u (the oracle itself differs from LAPACK by ~1e6 u at n=300),
  so for them the checks are the residual, the structure, the trace and the self-consistency
  of the returned eigenvalues -- no oracle comparison."""
import os
import numpy as np
import pytest

import oracle as O
from helpers import U, WARN_U, eig_backward_error_u, to_device, to_host

pytestmark = pytest.mark.gpu


def hessenberg_of_lcg(n):
    """Schur input = oracle Hessenberg form of the LCG matrix (test/schur 'lapack' initializer)."""
    A0 = O.random_fullpos(n)
    H = A0.copy(order="F"); Q = O.identity(n)
    O.hessenberg(H, Q)
    return A0, H, Q


def check_result(S, H0, Hs, Q0, Qs, real, imag, eig_tol_u):
    n = H0.shape[1]
    assert O.check_schur_form(Hs) == 0
    # Qs S Qs^T == Q0 H0 Q0^T
    X = Q0[:n] @ H0[:n] @ Q0[:n].T
    R = Qs[:n] @ Hs[:n] @ Qs[:n].T - X
    assert np.linalg.norm(R) / np.linalg.norm(X) < WARN_U * U
    assert O.orthogonality_u(Qs) < WARN_U
    # returned eigenvalues == eigenvalues of the diagonal blocks (reference 'eigenvalues' hook)
    wr, wi = O.extract_eigenvalues(Hs)
    assert O.match_eigenvalues(real + 1j * imag, wr + 1j * wi) < 1e3
    # conjugate pairs adjacent, positive imaginary part first (sep_sm.h:100-120)
    i = 0
    while i < n:
        if imag[i] != 0.0:
            assert imag[i] > 0 and imag[i + 1] == -imag[i] and real[i + 1] == real[i]
            i += 2
        else:
            i += 1
    assert abs(real.sum() - np.trace(H0[:n])) <= 1e-9 * max(1.0, np.abs(np.diag(H0[:n])).sum())
    if eig_tol_u is None:
        return
    # eigenvalue multiset against the oracle
    Ho = H0.copy(order="F"); Zo = O.identity(n, ld=H0.shape[0])
    wro, wio = O.schur(Ho, Zo)
    assert O.match_eigenvalues(real + 1j * imag, wro + 1j * wio) < eig_tol_u


@pytest.mark.parametrize("n", [1, 2, 3, 7, 50, 128, 129, 200, 400, 1000])
def test_schur_of_lcg_hessenberg_matches_oracle(node, n):
    A0, H0, Q0 = hessenberg_of_lcg(n)
    H = H0.copy(order="F"); Q = Q0.copy(order="F")
    real = np.zeros(n); imag = np.zeros(n)
    assert node.SEP_SM_Schur(n, H, H.shape[0], Q, Q.shape[0], real, imag) == 0
    check_result(node, H0, H, Q0, Q, real, imag, 1e4)
    # the whole chain reproduces the original dense matrix
    assert O.residual_u(Q, H, A0) < WARN_U


@pytest.mark.parametrize("n", [150, 300, 700])
def test_schur_of_random_hessenberg(node, n):
    """the test driver's `schur --init random`: random Hessenberg, 2*prand-1 (init.c:159-175)"""
    H0 = O.random_hessenberg(n)
    H = H0.copy(order="F"); Q0 = O.identity(n); Q = Q0.copy(order="F")
    real = np.zeros(n); imag = np.zeros(n)
    assert node.SEP_SM_Schur(n, H, H.shape[0], Q, Q.shape[0], real, imag) == 0
    check_result(node, H0, H, Q0, Q, real, imag, None)
    # conditioning-independent eigenvalue check: each one is an exact eigenvalue of a matrix
    # within 500 u of the input
    eye = np.zeros_like(H0); eye[np.arange(n), np.arange(n)] = 1.0
    assert eig_backward_error_u(H0, eye, real + 1j * imag, np.ones(n), sample=24) < WARN_U


@pytest.mark.parametrize("aed,shifts,small", [(50, 20, 100), (100, 60, 128), (200, 120, 150), (24, 8, 100)])
def test_expert_configurations(node, aed, shifts, small):
    """reference CTest sweep over AED sizes / shift counts (test/CMakeLists.txt:419-444)"""
    n = 500
    A0, H0, Q0 = hessenberg_of_lcg(n)
    conf = node.schur_init_conf()
    conf.aed_window_size = aed; conf.shift_count = shifts; conf.small_limit = small
    H = H0.copy(order="F"); Q = Q0.copy(order="F")
    real = np.zeros(n); imag = np.zeros(n)
    assert node.SEP_SM_Schur_expert(conf, n, H, H.shape[0], Q, Q.shape[0], real, imag) == 0
    check_result(node, H0, H, Q0, Q, real, imag, 1e4)


def test_lapack_threshold_and_null_eigenvalue_arrays(node):
    n = 300
    A0, H0, Q0 = hessenberg_of_lcg(n)
    conf = node.schur_init_conf()
    conf.left_threshold = -3.0          # STARNEIG_SCHUR_LAPACK_THRESHOLD
    H = H0.copy(order="F"); Q = Q0.copy(order="F")
    assert node.SEP_SM_Schur_expert(conf, n, H, H.shape[0], Q, Q.shape[0], None, None) == 0
    assert O.check_schur_form(H) == 0
    assert O.residual_u(Q, H, A0) < WARN_U
    conf.left_threshold = -7.0
    assert node.SEP_SM_Schur_expert(conf, n, H, H.shape[0], Q, Q.shape[0], None, None) == node.INVALID_CONFIGURATION
    conf = node.schur_init_conf(); conf.aed_window_size = 10; conf.shift_count = 20
    assert node.SEP_SM_Schur_expert(conf, n, H, H.shape[0], Q, Q.shape[0], None, None) == node.INVALID_ARGUMENTS


def test_already_triangular_and_block_diagonal_inputs(node):
    n = 260
    rng = np.random.RandomState(3)
    T0 = np.asfortranarray(np.triu(rng.uniform(-1, 1, (n, n))))
    T = T0.copy(order="F"); Q = np.asfortranarray(np.eye(n))
    real = np.zeros(n); imag = np.zeros(n)
    assert node.SEP_SM_Schur(n, T, n, Q, n, real, imag) == 0
    assert np.array_equal(T, T0) and np.array_equal(Q, np.eye(n))
    assert np.array_equal(real, np.diag(T0)) and not imag.any()
    # two decoupled Hessenberg blocks (exact zero on the sub-diagonal)
    H0 = O.random_hessenberg(n, ld=n); H0[130, 129] = 0.0
    H = H0.copy(order="F"); Q = np.asfortranarray(np.eye(n))
    assert node.SEP_SM_Schur(n, H, n, Q, n, real, imag) == 0
    check_result(node, H0, H, np.asfortranarray(np.eye(n)), Q, real, imag, None)


def test_full_chain_reduce_config1(node):
    """BASELINE config 1: Hessenberg + Schur of the 2000 x 2000 matrix, via starneig_SEP_SM_Reduce
    (common/combined.c:46-98) -- checks of examples/validate.c:63-65,99-101 (1000 u limit)."""
    n = 2000
    A0 = O.random_fullpos(n)
    A = A0.copy(order="F"); Q = O.identity(n)
    real = np.zeros(n); imag = np.zeros(n)
    assert node.SEP_SM_Reduce(n, A, A.shape[0], Q, Q.shape[0], real, imag) == 0
    assert O.check_schur_form(A) == 0
    tA, tQ, tA0 = to_device(A), to_device(Q), to_device(A0)
    rc, chk = node.check_device(tQ, tA, tA0, n=n)
    assert rc == 0 and chk["residual_u"] < 1000 and chk["orthogonality_u"] < 1000
    # trace and the dominant (Perron) eigenvalue of the positive matrix are preserved
    assert abs(real.sum() - np.trace(A0[:n])) <= 1e-10 * abs(np.trace(A0[:n]))
    assert abs(real.max() - np.abs(np.linalg.eigvals(A0[:n])).max()) <= 1e-10 * n


@pytest.mark.parametrize("n", [64, 200, 512, 2000])
def test_eigenvalues_match_lapack_golden(node, n):
    """committed numpy/LAPACK eigenvalues of the LCG matrices (tests/golden)"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"hessenberg_lcg2019_n{n}.npz"))
    A0 = O.random_fullpos(n, seed=int(g["seed"]))
    A = A0.copy(order="F"); Q = O.identity(n)
    real = np.zeros(n); imag = np.zeros(n)
    assert node.SEP_SM_Reduce(n, A, A.shape[0], Q, Q.shape[0], real, imag) == 0
    assert O.match_eigenvalues(real + 1j * imag, g["eig_real"] + 1j * g["eig_imag"]) < 1e4
    assert O.check_schur_form(A) == 0 and O.residual_u(Q, A, A0) < WARN_U


def test_select_on_schur_form(node):
    """starneig_SEP_SM_Select (common/helpers.c:47-101): pairs are selected together."""
    n = 200
    A0, H0, Q0 = hessenberg_of_lcg(n)
    H = H0.copy(order="F"); Q = Q0.copy(order="F")
    real = np.zeros(n); imag = np.zeros(n)
    assert node.SEP_SM_Schur(n, H, H.shape[0], Q, Q.shape[0], real, imag) == 0
    rc, sel, cnt = node.SEP_SM_Select(n, H, H.shape[0], lambda re, im: re > 0.0)
    assert rc == 0 and cnt == int((real > 0).sum()) and np.array_equal(sel.astype(bool), real > 0)
    for i in range(n - 1):
        if imag[i] > 0:
            assert sel[i] == sel[i + 1]


def _special_matrices(n):
    import scipy.linalg as sl
    rng = np.random.RandomState(0)
    comp = np.eye(n, k=-1); comp[0, n - 1] = 1.0
    comp2 = np.eye(n, k=-1); comp2[0, :] = rng.randn(n) * 1e-3
    return {
        "orthogonal": sl.hessenberg(sl.qr(rng.randn(n, n))[0]),
        "companion_unit_circle": comp,            # needs exceptional shifts (zero trailing block)
        "companion_random": comp2,
        "toeplitz_tridiagonal": 2 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1),
        "jordan_like": 3 * np.eye(n) + np.eye(n, k=1) + 1e-8 * np.eye(n, k=-1),
        "graded": sl.hessenberg(rng.randn(n, n) * np.logspace(0, -15, n)[:, None]),
        "symmetric": sl.hessenberg((lambda M: M + M.T)(rng.randn(n, n))),
        "all_ones": np.triu(np.ones((n, n)), -1),
        "zero": np.zeros((n, n)),
        "scaled_1e150": sl.hessenberg(rng.randn(n, n)) * 1e150,
        "scaled_1e-150": sl.hessenberg(rng.randn(n, n)) * 1e-150,
    }


@pytest.mark.parametrize("name", ["orthogonal", "companion_unit_circle", "companion_random",
                                  "toeplitz_tridiagonal", "jordan_like", "graded", "symmetric",
                                  "all_ones", "zero", "scaled_1e150", "scaled_1e-150"])
def test_special_matrices_converge(node, name):
    """hard inputs for QR iterations: equal-modulus spectra, defective and graded matrices,
    extreme scaling (the chase kernel's scaled reflectors, exceptional shifts)"""
    n = 400
    H0 = np.asfortranarray(_special_matrices(n)[name])
    H = H0.copy(order="F"); Q = np.asfortranarray(np.eye(n))
    real = np.zeros(n); imag = np.zeros(n)
    assert node.SEP_SM_Schur(n, H, n, Q, n, real, imag) == 0
    assert O.check_schur_form(H) == 0
    nrm = np.linalg.norm(H0)
    if nrm > 0:
        assert np.linalg.norm(Q @ H @ Q.T - H0) / nrm < WARN_U * U
    assert np.linalg.norm(Q @ Q.T - np.eye(n)) / np.sqrt(n) < WARN_U * U


@pytest.mark.parametrize("n,conf_vals", [(3500, None), (5000, (96, 60, 128)), (4000, (200, 120, 256))])
def test_device_resident_chain_exercises_update_zones(node, n, conf_vals):
    """Sizes beyond 8 AED windows: rows above the chain band and above the guard row, the
    deflated columns and Q are updated on the lazy streams; the device-resident
    chain must still pass the reference's acceptance checks and reproduce itself bit for bit
    (the zones only reorder commuting updates; a race would show up as run-to-run differences
    in H -- Q and H are compared after an identical Hessenberg input)."""
    import torch
    tA0 = node.device_matrix(n)
    assert node.lcg_fill_device(tA0, n, n, seed=2019, mode=0) == 0
    tH0 = tA0.clone(); tQ0 = node.device_matrix(n)
    node.set_matrix_device(tQ0, n, n, 0.0, 1.0)
    assert node.hessenberg_device(tH0, tQ0, n=n) == 0
    conf = None
    if conf_vals:
        conf = node.schur_init_conf()
        conf.aed_window_size, conf.shift_count, conf.small_limit = conf_vals
    results = []
    for rep in range(2):
        tH, tQ = tH0.clone(), tQ0.clone()
        rc, real, imag, st = node.schur_device(tH, tQ, n=n, conf=conf)
        torch.cuda.synchronize()
        assert rc == 0 and st["sweeps"] > 0
        rc, chk = node.check_device(tQ, tH, tA0, n=n)
        assert rc == 0
        assert chk["residual_u"] < WARN_U and chk["orthogonality_u"] < WARN_U
        assert chk["below_subdiagonal"] == 0
        results.append((tH, tQ, real, imag))
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
    S = to_host(results[0][0])
    assert O.check_schur_form(S) == 0
    assert abs(results[0][2].sum() - float(torch.diagonal(tA0[:, :n]).sum())) <= 1e-9 * n


@pytest.mark.parametrize("n,conf_vals", [(4000, (200, 120, 256)), (6000, (256, 160, 256))])
def test_helper_team_of_the_window_kernel_in_situ(node, n, conf_vals):
    """starneig_node_init with six cores or more switches the host window kernel to a serial chain plus
    helper threads (csrc/schur_host_team.h; what bench.py runs).  Every element still sees the same factors
    in the same order, so with the SAME configuration the whole reduction -- T, Q, the eigenvalues -- must
    equal the one-core run bit for bit, and it must do so twice (the helpers' timing may not matter)."""
    import torch
    tA0 = node.device_matrix(n)
    assert node.lcg_fill_device(tA0, n, n, seed=2019, mode=0) == 0
    tH0 = tA0.clone(); tQ0 = node.device_matrix(n)
    node.set_matrix_device(tQ0, n, n, 0.0, 1.0)
    assert node.hessenberg_device(tH0, tQ0, n=n) == 0
    conf = node.schur_init_conf()
    conf.aed_window_size, conf.shift_count, conf.small_limit = conf_vals
    results = []
    try:
        for cores in (1, 8, 8):
            node.node_finalize()
            node.node_init(cores, 1, node.NO_MESSAGES)
            tH, tQ = tH0.clone(), tQ0.clone()
            rc, real, imag, st = node.schur_device(tH, tQ, n=n, conf=conf)
            torch.cuda.synchronize()
            assert rc == 0 and st["aeds"] > 0
            results.append((tH, tQ, real, imag, st["aeds"], st["sweeps"]))
    finally:
        node.node_finalize()
        node.node_init(1, 1, node.NO_MESSAGES)
    rc, chk = node.check_device(results[1][1], results[1][0], tA0, n=n)
    assert rc == 0 and chk["residual_u"] < WARN_U and chk["orthogonality_u"] < WARN_U
    for other in results[1:]:
        assert other[4:] == results[0][4:]
        assert torch.equal(other[0], results[0][0]) and torch.equal(other[1], results[0][1])
        assert np.array_equal(other[2], results[0][2]) and np.array_equal(other[3], results[0][3])


def test_device_schur_without_q(node):
    """dQ = NULL: no accumulation (the lazy Q stream is idle); eigenvalues equal the run with Q"""
    import torch
    n = 3200
    tA0 = node.device_matrix(n)
    assert node.lcg_fill_device(tA0, n, n, seed=2019, mode=0) == 0
    tH0 = tA0.clone(); tQ0 = node.device_matrix(n)
    node.set_matrix_device(tQ0, n, n, 0.0, 1.0)
    assert node.hessenberg_device(tH0, tQ0, n=n) == 0
    tH1, tQ1 = tH0.clone(), tQ0.clone()
    rc, real1, imag1, _ = node.schur_device(tH1, tQ1, n=n)
    assert rc == 0
    tH2 = tH0.clone()
    rc, real2, imag2, _ = node.schur_device(tH2, None, n=n)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(tH1, tH2)
    assert np.array_equal(real1, real2) and np.array_equal(imag1, imag2)


@pytest.mark.parametrize("n,nw,ns", [(2500, 400, 300), (4000, 640, 480)])
def test_blocked_aed_for_windows_above_the_hard_limit(node, n, nw, ns):
    """Row S5 (schur/core.c:1423-1551, :1070-1252, :783-1052): AED windows above
    aed_parallel_hard_limit (default 300) take the blocked device path -- recursive Schur
    reduction of the private window, windowed deflation checks with carried / flushed
    undeflatable blocks, device re-Hessenberg.  The reference's own window rule 0.08 n gives
    320 at n = 4000; same acceptance checks, same spectrum as the default configuration."""
    import torch
    tA0 = node.device_matrix(n)
    assert node.lcg_fill_device(tA0, n, n, seed=2019, mode=0) == 0
    tH0 = tA0.clone(); tQ0 = node.device_matrix(n)
    node.set_matrix_device(tQ0, n, n, 0.0, 1.0)
    assert node.hessenberg_device(tH0, tQ0, n=n) == 0
    tH, tQ = tH0.clone(), tQ0.clone()
    rc, real0, imag0, st0 = node.schur_device(tH, tQ, n=n)
    assert rc == 0
    conf = node.schur_init_conf()
    conf.aed_window_size, conf.shift_count = nw, ns
    tH, tQ = tH0.clone(), tQ0.clone()
    rc, real, imag, st = node.schur_device(tH, tQ, n=n, conf=conf)
    torch.cuda.synchronize()
    assert rc == 0 and st["aeds"] > 0 and st["aeds"] < st0["aeds"]
    rc, chk = node.check_device(tQ, tH, tA0, n=n)
    assert rc == 0 and chk["below_subdiagonal"] == 0
    assert chk["residual_u"] < WARN_U and chk["orthogonality_u"] < WARN_U
    assert O.check_schur_form(to_host(tH)) == 0
    assert O.match_eigenvalues(real + 1j * imag, real0 + 1j * imag0) < 1e4
    # a hard limit above the window sends the same window through the sequential host kernel
    conf.aed_parallel_hard_limit = 1000
    tH, tQ = tH0.clone(), tQ0.clone()
    rc, real2, imag2, st2 = node.schur_device(tH, tQ, n=n, conf=conf)
    assert rc == 0
    assert O.match_eigenvalues(real2 + 1j * imag2, real0 + 1j * imag0) < 1e4
    print(f"n={n} window {nw}: blocked {st['total_ms']:.0f} ms ({st['aeds']} AEDs, {st['sweeps']} sweeps), "
          f"host kernel {st2['total_ms']:.0f} ms ({st2['aeds']} AEDs), default {st0['total_ms']:.0f} ms ({st0['aeds']} AEDs)")


@pytest.mark.parametrize("world,n", [(2, 3500), (3, 5000)])
def test_owner_only_tiles_of_the_deflated_columns(node, world, n):
    """Sharded Schur leg (include/starneig_amd.h, starneig_amd_schur_sharded_device): rank r of `world`
    keeps only the 128-column tiles T of the deflated part with T % world == r up to date.  Each rank's
    replica must (a) agree with the single-GPU reduction in its own tiles, on and below the diagonal
    blocks and in the eigenvalues, (b) really skip work -- differ from it in tiles of the others --
    and (c) the tiles taken from their owners give the single-GPU Schur form bit for bit."""
    import torch
    tA0 = node.device_matrix(n)
    assert node.lcg_fill_device(tA0, n, n, seed=2019, mode=0) == 0
    tH0 = tA0.clone(); tQ0 = node.device_matrix(n)
    node.set_matrix_device(tQ0, n, n, 0.0, 1.0)
    assert node.hessenberg_device(tH0, tQ0, n=n) == 0
    tH1, tQ1 = tH0.clone(), tQ0.clone()
    rc, real1, imag1, _ = node.schur_device(tH1, tQ1, n=n)
    assert rc == 0
    tile = torch.arange(n, device=tH1.device) // 128
    assembled = torch.empty_like(tH1)
    for rank in range(world):
        tH, tQ = tH0.clone(), tQ0.clone()
        rc, real, imag, _ = node.schur_sharded_device(tH, tQ, n, rank, world, n=n)
        torch.cuda.synchronize()
        assert rc == 0
        assert np.array_equal(real, real1) and np.array_equal(imag, imag1)
        assert torch.equal(tQ, tQ1)                                   # all rows of Q were this rank's here
        own = (tile % world) == rank                                  # tH[c] is column c of H
        assert torch.equal(tH[own], tH1[own])
        differs = (tH[~own] != tH1[~own])
        assert differs.any(), "no tile was skipped: the reduction was not sharded"
        # tile 0 is never owner-only; what differs lies above the tile's first row: a tile becomes
        # owner-only when the active block ends at or above its first column
        assert torch.equal(tH[tile == 0], tH1[tile == 0])
        cols = torch.nonzero(~own).flatten()[torch.nonzero(differs)[:, 0]]
        rows = torch.nonzero(differs)[:, 1]
        assert bool((rows < cols // 128 * 128).all())
        assembled[own] = tH[own]
    assert torch.equal(assembled, tH1)
