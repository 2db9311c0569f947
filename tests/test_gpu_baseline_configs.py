"""GPU tests at the sizes BASELINE.json names (configs 2, 3 and 5) and on structured inputs at
n >= 8000, through the device-pointer C-ABI, with the reference's own acceptance checks
(test/common/hooks.c:52,57: warn 500 u / fail 10000 u on residual and orthogonality;
hooks.c:434-456 exact zeros below the sub-diagonal; hooks.c:535-714 Schur form; trace).

Config 1 (n = 2000, host arrays) is tests/test_gpu_hessenberg.py / test_gpu_schur.py against the
golden fixtures; config 4 (8 GPUs) cannot run on a one-GPU box -- its code path is covered by
tests/test_gpu_distributed.py (2-3 processes on one device) and tests/test_distributed_cpu.py."""
import numpy as np
import pytest

import oracle as O
from helpers import WARN_U, to_host, torch_check

pytestmark = pytest.mark.gpu


def schur_form_ok_device(tS, n):
    """Quasi-triangular with standardised 2x2 blocks, checked on the device (a 20000^2 matrix
    would take a 3.2 GB host copy): nothing below the sub-diagonal, no two consecutive non-zero
    sub-diagonal entries, and for every 2x2 block equal diagonal entries and off-diagonal entries
    of opposite sign (LAPACK dlanv2 standard form; hooks.c:535-714)."""
    import torch
    M = tS[:, :n]                                   # M[c, r] = S(r, c)
    sub = torch.diagonal(M, offset=1)               # S(i+1, i)
    dg = torch.diagonal(M)
    sup = torch.diagonal(M, offset=-1)              # S(i, i+1)
    nz = sub != 0
    ok = not bool((nz[:-1] & nz[1:]).any())
    ok &= bool((dg[:-1][nz] == dg[1:][nz]).all())
    ok &= bool((sub[nz] * sup[nz] < 0).all())
    return ok


def test_config3_hessenberg_schur_n20000(node):
    """BASELINE config 3: Hessenberg + multi-shift QR Schur, n = 20000, Q accumulated, the test
    driver's LCG matrix (seed 2019), everything resident in HBM."""
    import torch
    n = 20000
    tA0 = node.device_matrix(n)
    assert node.lcg_fill_device(tA0, n, n, seed=2019, mode=0) == 0
    tH = tA0.clone(); tQ = node.device_matrix(n)
    node.set_matrix_device(tQ, n, n, 0.0, 1.0)
    assert node.hessenberg_device(tH, tQ, n=n) == 0
    rc, chk = node.check_device(tQ, tH, tA0, n=n)
    assert rc == 0 and chk["below_subdiagonal"] == 0
    # 1.5 x the reference's published Hessenberg residuals at n = 4000 (15 u / 11 u); no allowance
    # for the five times larger n
    assert chk["residual_u"] < 1.5 * 15 and chk["orthogonality_u"] < 1.5 * 11
    # the same two numbers from torch.matmul in fp64 (independent of the library's check kernels)
    res_t, orth_t = torch_check(tQ, tH, tA0, n)
    print(f"Hessenberg n={n}: library {chk['residual_u']:.2f} / {chk['orthogonality_u']:.2f} u, "
          f"torch {res_t:.2f} / {orth_t:.2f} u")
    assert abs(res_t - chk["residual_u"]) <= 0.2 * res_t and abs(orth_t - chk["orthogonality_u"]) <= 0.2 * orth_t
    assert res_t < 1.5 * 15 and orth_t < 1.5 * 11
    trace = float(torch.diagonal(tA0[:, :n]).sum())
    assert abs(float(torch.diagonal(tH[:, :n]).sum()) - trace) <= 1e-9 * n
    rc, real, imag, st = node.schur_device(tH, tQ, n=n)
    torch.cuda.synchronize()
    assert rc == 0 and st["sweeps"] > 0 and st["aeds"] > 0
    rc, chk = node.check_device(tQ, tH, tA0, n=n)
    assert rc == 0
    assert chk["residual_u"] < WARN_U and chk["orthogonality_u"] < WARN_U
    assert chk["below_subdiagonal"] == 0
    res_t, orth_t = torch_check(tQ, tH, tA0, n)
    print(f"Schur n={n}: library {chk['residual_u']:.2f} / {chk['orthogonality_u']:.2f} u, "
          f"torch {res_t:.2f} / {orth_t:.2f} u")
    assert abs(res_t - chk["residual_u"]) <= 0.2 * res_t and abs(orth_t - chk["orthogonality_u"]) <= 0.2 * orth_t
    assert res_t < WARN_U and orth_t < WARN_U
    assert schur_form_ok_device(tH, n)
    assert abs(real.sum() - trace) <= 1e-9 * n
    # eigenvalues returned == eigenvalues of the diagonal blocks; pairs adjacent, +imag first
    dg = torch.diagonal(tH[:, :n]).cpu().numpy()
    assert np.array_equal(real, dg)
    cplx = imag != 0
    idx = np.nonzero(cplx)[0]
    assert idx.size % 2 == 0
    first = idx[::2]
    assert np.all(imag[first] > 0) and np.all(imag[first + 1] == -imag[first]) and np.all(np.diff(idx)[::2] == 1)


def test_config5_qz_n12000(node):
    """BASELINE config 5: generalized Schur (QZ) reduction of the test driver's random
    Hessenberg-triangular pencil, n = 12000, Q = Z = I (test/schur/experiment.c:203-207)."""
    import torch
    n = 12000
    tH, tR = node.device_matrix(n), node.device_matrix(n)
    assert node.lcg_pencil_device(tH, tR, n, seed=2019) == 0
    tH0, tR0 = tH.clone(), tR.clone()
    tQ, tZ = node.device_matrix(n), node.device_matrix(n)
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
    assert ca["below_subdiagonal"] == 0
    # T upper triangular: nothing below the diagonal (sub-diagonal included)
    assert cb["below_subdiagonal"] == 0 and float(torch.diagonal(tR[:, :n], offset=1).abs().max()) == 0.0
    M = tH[:, :n]
    sub = torch.diagonal(M, offset=1)
    nz = sub != 0
    assert not bool((nz[:-1] & nz[1:]).any())
    # returned (alpha, beta) are consistent with the diagonal of the 1x1 blocks
    one = np.ones(n, dtype=bool)
    nzh = nz.cpu().numpy()
    one[:-1] &= ~nzh; one[1:] &= ~nzh
    dS = torch.diagonal(M).cpu().numpy(); dT = torch.diagonal(tR[:, :n]).cpu().numpy()
    assert np.allclose(ar[one] * dT[one], be[one] * dS[one], rtol=1e-12, atol=0.0)
    assert np.all(ai[one] == 0.0)


def structured(kind, n, M):
    """Writes the upper Hessenberg test matrix into M (M[c, r] = H(r, c), zero on entry)."""
    import torch
    idx = torch.arange(n, device="cuda")
    if kind == "all_ones":                 # H(r, c) = 1 for r <= c + 1
        M.copy_(torch.tril(torch.ones((n, n), dtype=torch.float64, device="cuda"), 1))
    elif kind == "toeplitz":               # tridiag(-1, 2, -1)
        M[idx, idx] = 2.0; M[idx[:-1], idx[1:]] = -1.0; M[idx[1:], idx[:-1]] = -1.0
    elif kind == "orthogonal":
        # unreduced orthogonal upper Hessenberg matrix: the product G_1 G_2 ... G_{n-1} of Givens
        # rotations (G_k acts on coordinates k, k+1), built column by column on the host in O(n^2)
        rng = np.random.RandomState(2019)
        th = rng.uniform(0.3, 2.8, n - 1)
        cs, sn = np.cos(th), np.sin(th)
        H = np.zeros((n, n))
        # apply the rotations to the identity from the right, one after the other: column k is
        # final after G_k, the other half of the rotation is carried into column k+1
        carry = np.zeros(n); carry[0] = 1.0
        for k in range(n - 1):
            e = np.zeros(n); e[k + 1] = 1.0
            H[:, k] = cs[k] * carry + sn[k] * e
            carry = -sn[k] * carry + cs[k] * e
        H[:, n - 1] = carry
        M.copy_(torch.from_numpy(np.ascontiguousarray(H.T)).cuda())
    else:
        raise ValueError(kind)


@pytest.mark.parametrize("kind", ["all_ones", "toeplitz", "orthogonal"])
def test_structured_inputs_n8000(node, kind):
    """Slowly converging structured Hessenberg matrices at n = 8000 (the failure mode
    shows at large n only): the reference's acceptance limits, residual and orthogonality below
    the warn level of 500 u (hooks.c:52).  Measured in round 2, residual / orthogonality in u:
    180 / 148 (all ones), 178 / 102 (Toeplitz), ~190 / 130 (orthogonal); LAPACK dhseqr (OpenBLAS
    0.3.29, same host, same matrices): 333 / 376 and 214 / 124.  Round 1 was at 1205 u / 1136 u:
    the 3x3 reflectors of the chase kernel scaled their input by a ROUNDED reciprocal
    (schur_common.h small_reflector; now an exact power of two), and the shift multiplicity was
    fixed (now adaptive, schur.hip `replicate`)."""
    import torch
    n = 8000
    tH0 = node.device_matrix(n)
    structured(kind, n, tH0[:, :n])
    tH = tH0.clone(); tQ = node.device_matrix(n)
    node.set_matrix_device(tQ, n, n, 0.0, 1.0)
    rc, real, imag, st = node.schur_device(tH, tQ, n=n)
    torch.cuda.synchronize()
    assert rc == 0
    rc, chk = node.check_device(tQ, tH, tH0, n=n)
    assert rc == 0
    assert chk["below_subdiagonal"] == 0 and schur_form_ok_device(tH, n)
    assert chk["residual_u"] < WARN_U and chk["orthogonality_u"] < WARN_U, (kind, chk, st)
    if kind == "toeplitz":      # known spectrum: 2 - 2 cos(k pi / (n + 1))
        ev = np.sort(real)
        ref = 2.0 - 2.0 * np.cos(np.arange(1, n + 1) * np.pi / (n + 1))
        assert np.all(imag == 0.0) and np.abs(ev - ref).max() < 1e4 * 2.0 ** -52 * 4.0
    if kind == "orthogonal":    # eigenvalues on the unit circle
        assert np.abs(np.hypot(real, imag) - 1.0).max() < 1e4 * 2.0 ** -52


def test_small_limit_above_window_with_lookahead(node):
    """an explicit small_limit far above the bulge window combined with a sweep head in
    flight -- the small-block branch must let the sweep through first."""
    import torch
    n = 3600
    tA0 = node.device_matrix(n)
    assert node.lcg_fill_device(tA0, n, n, seed=2019, mode=0) == 0
    tH = tA0.clone(); tQ = node.device_matrix(n)
    node.set_matrix_device(tQ, n, n, 0.0, 1.0)
    assert node.hessenberg_device(tH, tQ, n=n) == 0
    conf = node.schur_init_conf()
    conf.small_limit = 1024
    rc, real, imag, st = node.schur_device(tH, tQ, n=n, conf=conf)
    torch.cuda.synchronize()
    assert rc == 0
    rc, chk = node.check_device(tQ, tH, tA0, n=n)
    assert rc == 0 and chk["residual_u"] < WARN_U and chk["orthogonality_u"] < WARN_U
    assert chk["below_subdiagonal"] == 0
    assert O.check_schur_form(to_host(tH)) == 0


def test_schur_conf_range_checks(node):
    """schur/core.c:2360-2384 (thresholds -> INVALID_CONFIGURATION) and
    schur/process_args.c:271-437 (-> INVALID_ARGUMENTS), same accept/reject table."""
    n = 64
    H0 = O.random_hessenberg(n)
    ld = H0.shape[0]

    def run(**kw):
        conf = node.schur_init_conf()
        for k, v in kw.items():
            setattr(conf, k, v)
        H = H0.copy(order="F"); Q = O.identity(n, ld=ld)
        return node.SEP_SM_Schur_expert(conf, n, H, ld, Q, ld, None, None)

    IC, IA = node.INVALID_CONFIGURATION, node.INVALID_ARGUMENTS
    assert run() == 0
    assert run(left_threshold=0.0) == IC and run(left_threshold=-4.0) == IC
    assert run(right_threshold=-5.0) == IC and run(inf_threshold=-3.0) == IC
    assert run(left_threshold=-3.0) == 0 and run(left_threshold=1e-14) == 0
    assert run(iteration_limit=0) == IA and run(iteration_limit=10) == 0
    assert run(small_limit=2) == IA and run(small_limit=3) == 0
    assert run(shift_count=1) == IA and run(shift_count=2) == 0
    assert run(aed_window_size=4) == IA and run(aed_window_size=5) == 0
    assert run(aed_window_size=20, shift_count=21) == IA and run(aed_window_size=20, shift_count=20) == 0
    assert run(aed_nibble=0) == IA and run(aed_nibble=100) == IA and run(aed_nibble=99) == 0
    assert run(aed_parallel_soft_limit=0) == IA and run(aed_parallel_hard_limit=0) == IA
    assert run(window_size=4) == IA and run(window_size=-2) == 0 and run(window_size=64) == 0
    assert run(shifts_per_window=1) == IA and run(shifts_per_window=4) == 0
    assert run(update_width=0) == 0 and run(update_height=-7) == 0     # reference: warning + default
    assert run(tile_size=40) == 0
