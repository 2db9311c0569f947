"""GPU parity tests of the generalized Schur (QZ) leg -- BASELINE config 5: the HIP multi-shift
QZ path through the C-ABI (starneig_GEP_SM_Schur and its device-pointer twin) against the CPU
oracle (oracle/gep_oracle.c), the committed LAPACK golden eigenvalues and the reference's
acceptance checks (test/common/checks.c residuals, hooks.c Schur-form structure).

Generalized Schur forms are not unique, so parity is on invariants.  Tolerances in u = 2^-52:
residuals / orthogonality < 500 u (the reference's warn threshold); eigenvalues against the
oracle and against the LAPACK golden vectors within max(1e4 u, 20 x lapack_spread_u,
50 x sens_u_per_u), both stored in the fixture: lapack_spread_u is the distance between LAPACK's
own real and complex QZ answers for that pencil, sens_u_per_u the measured movement of the
eigenvalues under elementwise perturbations of one u (so 50 x allows a backward error of 50 u,
a tenth of the reference's warn threshold).  Random triangular factors are exponentially ill
conditioned in n, which is why the LCG pencils need this and why the `wellcond` variant exists."""
import os

import numpy as np
import pytest

import oracle as O
from helpers import U, WARN_U, eig_backward_error_u, to_device, to_host

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check_pencil(H0, R0, S, T, Q, Z, ar, ai, be):
    n = H0.shape[1]
    assert O.check_gep_schur_form(S, T) == 0
    assert O.pencil_residual_u(Q, S, Z, H0) < WARN_U
    assert O.pencil_residual_u(Q, T, Z, R0) < WARN_U
    assert O.orthogonality_u(Q) < WARN_U and O.orthogonality_u(Z) < WARN_U
    # returned eigenvalues == eigenvalues of the diagonal blocks
    er, ei, eb = O.gep_extract_eigenvalues(S, T)
    # the reference's `eigenvalues` hook at its own thresholds (warn 10^3 u, fail 10^4 u, hooks.c:787-788):
    # both sides evaluate LAPACK dlag2's algorithm on the same 2 x 2 blocks, as in the reference
    hook = O.eigenvalues_check((er, ei, eb), (ar, ai, be))
    assert hook["failures"] == 0 and hook["warnings"] == 0, hook
    assert np.all(be >= 0.0)
    k = 0
    while k < n:
        if ai[k] != 0.0:
            assert ai[k] > 0.0 and ai[k + 1] < 0.0 and S[k + 1, k] != 0.0
            k += 2
        else:
            assert k + 1 >= n or S[k + 1, k] == 0.0
            k += 1


def run_host(node, H0, R0, conf=None):
    n = H0.shape[1]
    H, R = H0.copy(order="F"), R0.copy(order="F")
    Q, Z = O.identity(n, ld=H.shape[0]), O.identity(n, ld=H.shape[0])
    ar, ai, be = np.zeros(n), np.zeros(n), np.zeros(n)
    if conf is None:
        rc = node.GEP_SM_Schur(n, H, H.shape[0], R, R.shape[0], Q, Q.shape[0], Z, Z.shape[0], ar, ai, be)
    else:
        rc = node.GEP_SM_Schur_expert(conf, n, H, H.shape[0], R, R.shape[0], Q, Q.shape[0], Z, Z.shape[0],
                                      ar, ai, be)
    assert rc == 0
    return H, R, Q, Z, ar, ai, be


@pytest.mark.parametrize("threshold", ["default", "lapack"])
@pytest.mark.parametrize("kind", ["lcg2019", "wellcond2019"])
@pytest.mark.parametrize("n", [48, 150, 400])
def test_qz_against_oracle_and_lapack_golden(node, kind, n, threshold):
    """Eigenvalue parity.  (1) Conditioning-independent: every returned pair (alpha, beta) must be
    an exact eigenvalue of a pencil within 500 u of the input (sigma_min(beta A - alpha B)
    relative to |beta| ||A|| + |alpha| ||B||) -- the same bar as the residual.  (2) Forward
    comparison with the LAPACK golden vectors and the oracle where the conditioning allows it
    (n <= 150): `lapack` = STARNEIG_SCHUR_LAPACK_THRESHOLD (local deflation criteria,
    cpu_utils.c:2937-2988) within 50 x sens_u_per_u; `default` = the reference's norm-stable
    criterion |spike| < u ||H||_F (schur/core.c:2425-2436, cpu_utils.c:2891-2917), which
    discards far larger entries than LAPACK's and moves ill-conditioned eigenvalues
    accordingly (measured 250 x sens on the well-conditioned family -> 2000 x sens; on the LCG
    pencils, whose triangular factor has cond 1e18-1e20, no forward comparison)."""
    g = np.load(os.path.join(GOLD, f"gep_{kind}_n{n}.npz"))
    H0, R0 = O.random_pencil(n) if kind == "lcg2019" else O.random_pencil_wellcond(n)
    conf = None
    if threshold == "lapack":
        conf = node.schur_init_conf()
        conf.left_threshold = -3.0          # STARNEIG_SCHUR_LAPACK_THRESHOLD
    S, T, Q, Z, ar, ai, be = run_host(node, H0, R0, conf)
    check_pencil(H0, R0, S, T, Q, Z, ar, ai, be)
    assert eig_backward_error_u(H0, R0, ar + 1j * ai, be) < WARN_U
    if n > 150 or (kind == "lcg2019" and threshold == "default"):
        return
    ev = (ar + 1j * ai) / be
    tol = max(1e4, 20.0 * float(g["lapack_spread_u"]),
              (50.0 if threshold == "lapack" else 2000.0) * float(g["sens_u_per_u"]))
    assert O.match_eigenvalues(ev, g["eig_real"] + 1j * g["eig_imag"]) < tol
    Ho, Ro = H0.copy(order="F"), R0.copy(order="F")
    info, oar, oai, obe = O.gep_schur(Ho, Ro, O.identity(n, ld=Ho.shape[0]), O.identity(n, ld=Ho.shape[0]))
    assert info == 0
    assert O.match_eigenvalues(ev, (oar + 1j * oai) / obe) < tol


@pytest.mark.parametrize("n", [1, 2, 3, 7, 63, 64, 65, 97, 129, 257, 700, 1500])
def test_qz_sizes(node, n):
    """edge sizes around the 64-row LDS window, the small-block limit and several chains"""
    H0, R0 = O.random_pencil_wellcond(n, seed=11 + n)
    S, T, Q, Z, ar, ai, be = run_host(node, H0, R0)
    check_pencil(H0, R0, S, T, Q, Z, ar, ai, be)
    if n <= 700:
        assert eig_backward_error_u(H0, R0, ar + 1j * ai, be, sample=16) < WARN_U
    if n <= 129:
        Ho, Ro = H0.copy(order="F"), R0.copy(order="F")
        info, oar, oai, obe = O.gep_schur(Ho, Ro, O.identity(n, ld=Ho.shape[0]), O.identity(n, ld=Ho.shape[0]))
        assert info == 0
        # random Hessenberg spectra are ill conditioned (cf. tests/test_gpu_schur.py): beyond
        # n ~ 130 two backward-stable solvers disagree by 1e8 u and more, so larger sizes are
        # checked through the decomposition only
        assert O.match_eigenvalues((ar + 1j * ai) / be, (oar + 1j * oai) / obe) < 1e7


@pytest.mark.parametrize("aed,shifts,small", [(40, 20, 96), (96, 60, 128), (200, 120, 150), (24, 8, 100)])
def test_qz_expert_configurations(node, aed, shifts, small):
    n = 600
    H0, R0 = O.random_pencil_wellcond(n, seed=5)
    conf = node.schur_init_conf()
    conf.aed_window_size = aed; conf.shift_count = shifts; conf.small_limit = small
    S, T, Q, Z, ar, ai, be = run_host(node, H0, R0, conf)
    check_pencil(H0, R0, S, T, Q, Z, ar, ai, be)


def test_qz_nontrivial_q_and_z(node):
    """Q and Z are updated from the right: start from random orthogonal factors"""
    n = 300
    H0, R0 = O.random_pencil_wellcond(n, seed=3)
    rng = np.random.default_rng(0)
    Q0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, n)))[0])
    Z0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, n)))[0])
    H, R, Q, Z = H0[:n].copy(order="F"), R0[:n].copy(order="F"), Q0.copy(order="F"), Z0.copy(order="F")
    ar, ai, be = np.zeros(n), np.zeros(n), np.zeros(n)
    assert node.GEP_SM_Schur(n, H, n, R, n, Q, n, Z, n, ar, ai, be) == 0
    A0 = Q0 @ H0[:n] @ Z0.T
    B0 = Q0 @ R0[:n] @ Z0.T
    assert np.linalg.norm(Q @ H @ Z.T - A0) / np.linalg.norm(A0) < WARN_U * U
    assert np.linalg.norm(Q @ R @ Z.T - B0) / np.linalg.norm(B0) < WARN_U * U
    assert O.check_gep_schur_form(H, R) == 0


def test_qz_argument_errors(node):
    n = 8
    H0, R0 = O.random_pencil(n)
    Q, Z = O.identity(n), O.identity(n)
    v = np.zeros(n)
    ld = H0.shape[0]
    assert node.GEP_SM_Schur(0, H0, ld, R0, ld, Q, ld, Z, ld, v, v, v) == -1
    assert node.GEP_SM_Schur(n, None, ld, R0, ld, Q, ld, Z, ld, v, v, v) == -2
    assert node.GEP_SM_Schur(n, H0, n - 1, R0, ld, Q, ld, Z, ld, v, v, v) == -3
    assert node.GEP_SM_Schur(n, H0, ld, None, ld, Q, ld, Z, ld, v, v, v) == -4
    assert node.GEP_SM_Schur(n, H0, ld, R0, n - 1, Q, ld, Z, ld, v, v, v) == -5
    assert node.GEP_SM_Schur(n, H0, ld, R0, ld, None, ld, Z, ld, v, v, v) == -6
    assert node.GEP_SM_Schur(n, H0, ld, R0, ld, Q, ld, None, ld, v, v, v) == -8
    # real/imag/beta are not argument-checked (schur/interface.c:286-294 stops at -9)


def test_qz_device_pencil_generator_bit_exact(node):
    """the in-HBM LCG pencil generator equals the oracle's (reference init.c) bit for bit"""
    for n in (5, 64, 333):
        tH, tR = node.device_matrix(n), node.device_matrix(n)
        assert node.lcg_pencil_device(tH, tR, n, seed=2019) == 0
        H, R = O.random_pencil(n, ld=tH.shape[1])
        assert np.array_equal(to_host(tH), H) and np.array_equal(to_host(tR), R)


@pytest.mark.parametrize("n", [3000])
def test_qz_device_resident_lcg_pencil(node, n):
    """BASELINE config 5 shape at a size the suite affords: device-resident LCG pencil, Q = Z = I,
    acceptance checks computed on the GPU"""
    import torch
    tH, tR = node.device_matrix(n), node.device_matrix(n)
    assert node.lcg_pencil_device(tH, tR, n, seed=2019) == 0
    tH0, tR0 = tH.clone(), tR.clone()
    tQ, tZ = node.device_matrix(n), node.device_matrix(n)
    node.set_matrix_device(tQ, n, n, 0.0, 1.0); node.set_matrix_device(tZ, n, n, 0.0, 1.0)
    rc, ar, ai, be, st = node.gep_schur_device(tH, tR, tQ, tZ, n=n)
    assert rc == 0
    torch.cuda.synchronize()
    rc, ca = node.check_pencil_device(tQ, tH, tZ, tH0, n=n)
    assert rc == 0
    rc, cb = node.check_pencil_device(tQ, tR, tZ, tR0, n=n)
    assert rc == 0
    assert ca["residual_u"] < WARN_U and cb["residual_u"] < WARN_U
    assert ca["orthogonality_q_u"] < WARN_U and ca["orthogonality_z_u"] < WARN_U
    assert ca["below_subdiagonal"] == 0
    S, T = to_host(tH), to_host(tR)
    assert O.check_gep_schur_form(S, T) == 0
    assert st["sweeps"] > 0 and st["aeds"] > 0


def test_qz_device_without_q_and_z(node):
    """dQ = dZ = NULL (and only one of them): the pencil result does not depend on the accumulation"""
    import torch
    n = 1500
    tH0, tR0 = node.device_matrix(n), node.device_matrix(n)
    assert node.lcg_pencil_device(tH0, tR0, n, seed=7) == 0
    ref = None
    for use_q, use_z in ((True, True), (False, False), (True, False), (False, True)):
        tH, tR = tH0.clone(), tR0.clone()
        tQ = node.device_matrix(n) if use_q else None
        tZ = node.device_matrix(n) if use_z else None
        if tQ is not None:
            node.set_matrix_device(tQ, n, n, 0.0, 1.0)
        if tZ is not None:
            node.set_matrix_device(tZ, n, n, 0.0, 1.0)
        rc, ar, ai, be, _ = node.gep_schur_device(tH, tR, tQ, tZ, n=n)
        assert rc == 0
        torch.cuda.synchronize()
        if ref is None:
            ref = (tH, tR, ar, ai, be)
            rc, ca = node.check_pencil_device(tQ, tH, tZ, tH0, n=n)
            assert rc == 0 and ca["residual_u"] < WARN_U
        else:
            assert torch.equal(ref[0], tH) and torch.equal(ref[1], tR)
            assert np.array_equal(ref[2], ar) and np.array_equal(ref[4], be)


@pytest.mark.parametrize("n,zeros", [(40, [7]), (300, [0, 5, 100, 101, 299]), (700, [350, 351, 352, 698])])
def test_qz_infinite_eigenvalues_are_deflated(node, n, zeros):
    """Row S9: exact zeros on the diagonal of B (singular B) -- the infinite eigenvalues are chased
    to the top of their block and deflated with beta = 0 (schur/cpu_utils.c:360-425, :605-681,
    schur/core.c:475-552), not perturbed away; the finite ones agree with LAPACK's QZ."""
    import scipy.linalg as sl
    H0, R0 = O.random_pencil_wellcond(n)
    for k in zeros:
        R0[k, k] = 0.0
    H, R, Q, Z, ar, ai, be = run_host(node, H0, R0)
    assert O.check_gep_schur_form(H, R) == 0
    assert O.pencil_residual_u(Q, H, Z, H0) < WARN_U and O.pencil_residual_u(Q, R, Z, R0) < WARN_U
    assert O.orthogonality_u(Q) < WARN_U and O.orthogonality_u(Z) < WARN_U
    inf = be == 0.0
    # (consecutive zeros on B's diagonal share infinite eigenvalues: [[0, b], [0, 0]] has rank 1 --
    # the count is LAPACK's on the same pencil: 1, 4 and 3 for the three cases)
    w = sl.eig(H0[:n], R0[:n], right=False, homogeneous_eigvals=True)
    n_inf = int((np.abs(w[1]) < 1e-10 * np.abs(w[0])).sum())
    assert 1 <= n_inf <= len(zeros)
    assert int(inf.sum()) == n_inf, (int(inf.sum()), n_inf, np.sort(np.abs(be))[:8])
    # an infinite eigenvalue is a 1x1 block with T(i,i) exactly zero and a real alpha != 0
    for i in np.nonzero(inf)[0]:
        assert R[i, i] == 0.0 and ai[i] == 0.0 and ar[i] == H[i, i] and ar[i] != 0.0
        assert (i + 1 >= n or H[i + 1, i] == 0.0) and (i == 0 or H[i, i - 1] == 0.0)
    # finite eigenvalues against LAPACK (scipy.linalg.eig on the pencil, homogeneous form)
    fin = np.abs(w[1]) >= 1e-10 * np.abs(w[0])
    fin_ref = w[0][fin] / w[1][fin]
    lam = (ar[~inf] + 1j * ai[~inf]) / be[~inf]
    assert lam.size == fin_ref.size
    # forward comparison where the infinite eigenvalues are simple; a shared (defective) infinite
    # eigenvalue makes its finite neighbours ill-conditioned (1e-6 relative between LAPACK and any
    # other backward-stable QZ) -- there the conditioning-independent check decides: every
    # returned pair is an exact eigenvalue of a pencil within 500 u of the input
    if all(b - a > 1 for a, b in zip(zeros, zeros[1:])):
        assert O.match_eigenvalues(lam, fin_ref) < 1e7
    assert eig_backward_error_u(H0, R0, ar + 1j * ai, be, sample=32) < WARN_U


def test_qz_infinite_eigenvalues_device_n3000(node):
    """the same on a device-resident pencil with 1 % of B's diagonal zeroed"""
    import torch
    n = 3000
    H0, R0 = O.random_pencil_wellcond(n)
    zeros = list(range(17, n, 100))
    for k in zeros:
        R0[k, k] = 0.0
    tH, tR = to_device(H0), to_device(R0)
    tH0, tR0 = tH.clone(), tR.clone()
    tQ, tZ = node.device_matrix(n, ld=H0.shape[0]), node.device_matrix(n, ld=H0.shape[0])
    node.set_matrix_device(tQ, n, n, 0.0, 1.0); node.set_matrix_device(tZ, n, n, 0.0, 1.0)
    rc, ar, ai, be, st = node.gep_schur_device(tH, tR, tQ, tZ, n=n)
    torch.cuda.synchronize()
    assert rc == 0
    rc, ca = node.check_pencil_device(tQ, tH, tZ, tH0, n=n)
    assert rc == 0
    rc, cb = node.check_pencil_device(tQ, tR, tZ, tR0, n=n)
    assert rc == 0
    assert ca["residual_u"] < WARN_U and cb["residual_u"] < WARN_U
    assert ca["orthogonality_q_u"] < WARN_U and ca["orthogonality_z_u"] < WARN_U
    assert ca["below_subdiagonal"] == 0 and cb["below_subdiagonal"] == 0
    assert int((be == 0.0).sum()) == len(zeros)
    assert O.check_gep_schur_form(to_host(tH), to_host(tR)) == 0


@pytest.mark.parametrize("n", [150, 700])
def test_right_threshold_decides_b_side_deflation(node, n):
    """conf->right_threshold (expert.h:337-349; thres_b of schur/core.c:2438-2449): the magnitude
    below which an entry of R is negligible.  A diagonal entry d = 1e-9 at the bottom of R is far
    above every default threshold (u ||R||_F ~ 1e-15): with the defaults the pencil has n finite
    eigenvalues, one of them ~1e9; with right_threshold = 1e-6 the entry is set to zero and split
    off as an infinite eigenvalue (beta = 0 exactly), at a backward error of d / ||R||; the
    norm-stable setting (-2) and the LAPACK setting (-3) behave like the default here.
    n = 150: the small-block kernel; n = 700: the AED window kernel of the device path."""
    H0, R0 = O.random_pencil_wellcond(n)
    d = 1e-9
    R0[n - 1, n - 1] = d
    counts = {}
    for label, thr in (("default", -1.0), ("norm_stable", -2.0), ("lapack", -3.0), ("explicit", 1e-6)):
        conf = node.schur_init_conf()
        conf.right_threshold = thr
        S, T, Q, Z, ar, ai, be = run_host(node, H0, R0, conf)
        assert O.check_gep_schur_form(S, T) == 0
        assert O.pencil_residual_u(Q, S, Z, H0) < WARN_U
        assert O.orthogonality_u(Q) < WARN_U and O.orthogonality_u(Z) < WARN_U
        resb = O.pencil_residual_u(Q, T, Z, R0)
        counts[label] = int((be == 0.0).sum())
        if label == "explicit":
            # the dropped entry IS the backward error: d / ||R||_F, nothing more
            assert resb * U <= 1.5 * d / np.linalg.norm(R0[:n]) + WARN_U * U, resb
            i = int(np.nonzero(be == 0.0)[0][0])
            assert T[i, i] == 0.0 and ai[i] == 0.0 and ar[i] == S[i, i]
        else:
            assert resb < WARN_U
            assert np.abs((ar + 1j * ai) / be).max() > 1e7      # the huge finite eigenvalue
    assert counts == {"default": 0, "norm_stable": 0, "lapack": 0, "explicit": 1}, counts
    conf = node.schur_init_conf()
    conf.right_threshold = -5.0
    H, R = H0.copy(order="F"), R0.copy(order="F")
    Q, Z = O.identity(n, ld=H.shape[0]), O.identity(n, ld=H.shape[0])
    assert node.GEP_SM_Schur_expert(conf, n, H, H.shape[0], R, R.shape[0], Q, Q.shape[0], Z, Z.shape[0],
                                    None, None, None) == node.INVALID_CONFIGURATION


def test_general_pencil_chain_with_shift_multiplicity(node):
    """A GENERAL pencil (two LCG matrices) through both generalized steps on the device: on its
    Hessenberg-triangular form the QZ sweeps, not only the AED windows, carry the reduction, and every shift pair
    of an AED drives several bulges from the fifth sweep on (csrc/schur_gep.hip; one bulge per pair took 13 sweeps
    at n = 3000 with a multiplicity of 2 and more without).  The reference's acceptance checks on the whole chain."""
    import torch
    n = 3000
    tA, tB = node.device_matrix(n), node.device_matrix(n)
    assert node.lcg_fill_device(tA, n, n, seed=2019) == 0 and node.lcg_fill_device(tB, n, n, seed=77) == 0
    tA0, tB0 = tA.clone(), tB.clone()
    tQ, tZ = node.device_matrix(n), node.device_matrix(n)
    node.set_matrix_device(tQ, n, n, 0.0, 1.0); node.set_matrix_device(tZ, n, n, 0.0, 1.0)
    rc, _ = node.hessenberg_triangular_device(tA, tB, tQ, tZ, n=n)
    assert rc == 0
    rc, ar, ai, be, st = node.gep_schur_device(tA, tB, tQ, tZ, n=n)
    torch.cuda.synchronize()
    assert rc == 0
    assert 4 < st["sweeps"] <= 10, st              # 4 sweeps with one bulge per pair, then few with eight
    rc, ca = node.check_pencil_device(tQ, tA, tZ, tA0, n=n)
    assert rc == 0
    rc, cb = node.check_pencil_device(tQ, tB, tZ, tB0, n=n)
    assert rc == 0
    assert ca["residual_u"] < WARN_U and cb["residual_u"] < WARN_U
    assert ca["orthogonality_q_u"] < WARN_U and ca["orthogonality_z_u"] < WARN_U
    assert ca["below_subdiagonal"] == 0
    assert O.check_gep_schur_form(to_host(tA), to_host(tB)) == 0
    assert int((be == 0).sum()) == 0
