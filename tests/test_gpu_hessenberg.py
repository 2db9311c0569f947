"""GPU parity tests proper: the HIP Hessenberg path, called through the C-ABI,
against the CPU oracle on the same seeded inputs, against the committed LAPACK
golden vectors, and -- at the BASELINE sizes -- through the reference's
size-independent acceptance checks (exact Hessenberg structure, residual and
orthogonality in units of u)."""
import os

import numpy as np
import pytest

import oracle as O
from helpers import U, WARN_U, elementwise_tolerance, to_device, to_host, torch_check
from test_oracle import partial_input

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

# reference's own measured values at n=4000 (docs/_7_test_driver.md:233-247): 15 u / 11 u; the
# limit is 1.5 x those (round 3 measures 9.5 u / 8.9 u at n = 4000, 9.7 u / 9.2 u at n = 8000: the
# GEMM updates sum every product from zero in chunks of 256 terms)
REF_RESIDUAL_U = 15.0 * 1.5
REF_ORTH_U = 11.0 * 1.5


def run_host_api(S, A0, begin=0, end=None, panel_width=None):
    n = A0.shape[1]
    A = A0.copy(order="F"); Q = O.identity(n, ld=A0.shape[0])
    if begin == 0 and end is None and panel_width is None:
        rc = S.SEP_SM_Hessenberg(n, A, A.shape[0], Q, Q.shape[0])
    else:
        conf = S.hessenberg_init_conf()
        if panel_width: conf.panel_width = panel_width
        rc = S.SEP_SM_Hessenberg_expert(conf, n, begin, n if end is None else end,
                                        A, A.shape[0], Q, Q.shape[0])
    assert rc == 0
    return A, Q


def compare_with_oracle(A_gpu, Q_gpu, A0, **kw):
    n = A0.shape[1]
    Ao = A0.copy(order="F"); Qo = O.identity(n, ld=A0.shape[0])
    O.hessenberg(Ao, Qo, **kw)
    scale = np.linalg.norm(A0[:n])
    tol = elementwise_tolerance(n)
    assert np.abs(A_gpu[:n] - Ao[:n]).max() / scale <= tol
    assert np.abs(Q_gpu[:n] - Qo[:n]).max() <= tol * np.sqrt(n)
    sub_g, sub_o = np.diag(A_gpu[:n], -1), np.diag(Ao[:n], -1)
    assert np.array_equal(np.sign(sub_g), np.sign(sub_o))
    # structure: bit-exact zeros exactly where the oracle has them
    assert np.array_equal(A_gpu[:n] == 0.0, Ao[:n] == 0.0) or O.count_below_subdiagonal(A_gpu) == 0


@pytest.mark.parametrize("n", [1, 2, 3, 4, 17, 64, 129, 300, 777])
def test_matches_oracle_default_conf(node, n):
    A0 = O.random_fullpos(n)
    A, Q = run_host_api(node, A0)
    assert O.count_below_subdiagonal(A) == 0
    compare_with_oracle(A, Q, A0)
    if n > 1:
        assert O.residual_u(Q, A, A0) < WARN_U
        assert O.orthogonality_u(Q) < WARN_U


@pytest.mark.parametrize("n", [64, 200, 512, 2000])
def test_matches_lapack_golden(node, n):
    g = np.load(os.path.join(GOLDEN, f"hessenberg_lcg2019_n{n}.npz"))
    A0 = O.random_fullpos(n, seed=int(g["seed"]))
    A, Q = run_host_api(node, A0)
    H = A[:n]
    tol = elementwise_tolerance(n) * float(g["a_fro"])
    assert np.abs(np.diag(H, -1) - g["h_subdiag"]).max() <= tol
    assert np.array_equal(np.sign(np.diag(H, -1)), np.sign(g["h_subdiag"]))
    assert np.abs(np.diag(H) - g["h_diag"]).max() <= tol
    assert np.abs(H[0, :] - g["h_first_row"]).max() <= tol
    assert np.abs(H[:, -1] - g["h_last_col"]).max() <= tol
    assert O.count_below_subdiagonal(A) == 0


@pytest.mark.parametrize("pw", [8, 35, 45, 64, 170, 303])
def test_panel_widths(node, pw):
    """reference CTest sweep: panel widths 45 314 400 410 170 35 303 (test/CMakeLists.txt:367)."""
    n = 400
    A0 = O.random_fullpos(n)
    A, Q = run_host_api(node, A0, panel_width=pw)
    assert O.count_below_subdiagonal(A) == 0
    compare_with_oracle(A, Q, A0, panel_width=pw)


_ORACLE_4000 = {}


@pytest.mark.parametrize("pw", [45, 314, 400, 410, 170, 35, 303])
def test_ctest_hessenberg_panel_widths_n4000(node, pw):
    """hessenberg-panel-{45,314,400,410,170,35,303}: `--experiment hessenberg --n 4000 --panel-width pw`
    (test/CMakeLists.txt:367-388), at the reference's own size: exact Hessenberg structure, the reference's
    residual hooks, and H elementwise against the oracle (ONE oracle reduction at n = 4000 serves the seven
    widths: the panel width only re-associates the sums, and the tolerance 8 sqrt(n) u ||A||_F is the one the
    same-width comparisons at n = 400 use)."""
    n = 4000
    A0 = O.random_fullpos(n)
    A, Q = run_host_api(node, A0, panel_width=pw)
    assert O.count_below_subdiagonal(A) == 0
    if "H" not in _ORACLE_4000:
        Ao = A0.copy(order="F"); Qo = O.identity(n, ld=A0.shape[0])
        O.hessenberg(Ao, Qo)
        _ORACLE_4000["H"], _ORACLE_4000["Q"] = Ao, Qo
    Ao, Qo = _ORACLE_4000["H"], _ORACLE_4000["Q"]
    tol = elementwise_tolerance(n)
    err = np.abs(A[:n] - Ao[:n]).max() / np.linalg.norm(A0[:n])
    errq = np.abs(Q[:n] - Qo[:n]).max()
    print(f"panel width {pw}: max|H - H_oracle| / ||A|| = {err / tol:.3f} tol, max|Q - Q_oracle| = {errq / (tol * np.sqrt(n)):.3f} tol")
    assert err <= tol and errq <= tol * np.sqrt(n)
    assert np.array_equal(np.sign(np.diag(A[:n], -1)), np.sign(np.diag(Ao[:n], -1)))
    res, orth = torch_check(to_device(Q), to_device(A), to_device(A0), n)
    assert res < 1.5 * 15 and orth < 1.5 * 11, (res, orth)      # 1.5 x the reference's published n = 4000 values


def test_ctest_hessenberg_partial_3569(node):
    """hessenberg-partial-3569: `--experiment partial-hessenberg --n 3569 --begin 892 --end 2676`
    (test/CMakeLists.txt:390-405), elementwise against the oracle on the same range."""
    n = 3569
    begin, end = n // 4, 3 * n // 4
    A0 = partial_input(n, begin, end)
    A, Q = run_host_api(node, A0, begin=begin, end=end)
    H = A[:n]
    for c in range(n - 1):
        k = 2 if begin <= c < end - 1 else 1
        assert np.all(H[c + k:, c] == 0.0)
    compare_with_oracle(A, Q, A0, begin=begin, end=end)
    res, orth = torch_check(to_device(Q), to_device(A), to_device(A0), n)
    assert res < WARN_U and orth < WARN_U


@pytest.mark.parametrize("n,begin,end", [(47, 3, 40), (88, 0, 50), (88, 20, 88), (333, 100, 250), (554, 1, 553)])
def test_partial_ranges(node, n, begin, end):
    """reference partial-hessenberg experiment, n in {47,88,333,554,...} (test/CMakeLists.txt:390-405)."""
    A0 = partial_input(n, begin, end)
    A, Q = run_host_api(node, A0, begin=begin, end=end, panel_width=32)
    H = A[:n]
    for c in range(n - 1):
        k = 2 if begin <= c < end - 1 else 1
        assert np.all(H[c + k:, c] == 0.0)
    compare_with_oracle(A, Q, A0, begin=begin, end=end, panel_width=32)
    assert O.residual_u(Q, A, A0) < WARN_U


def test_odd_leading_dimension_and_nonidentity_q(node):
    """ld > n and odd (8-byte fallback of the gemv), Q_in a random orthogonal matrix."""
    n, ld = 150, 157
    A0 = O.random_fullpos(n, ld=ld)
    rng = np.random.RandomState(1)
    Q0 = np.zeros((ld, n), order="F"); Q0[:n] = np.linalg.qr(rng.randn(n, n))[0]
    A = A0.copy(order="F"); Q = Q0.copy(order="F")
    assert node.SEP_SM_Hessenberg(n, A, ld, Q, ld) == 0
    assert O.count_below_subdiagonal(A) == 0
    # Q_out H Q_out^T = Q_in A Q_in^T
    X = Q0[:n] @ A0[:n] @ Q0[:n].T
    R = Q[:n] @ A[:n] @ Q[:n].T - X
    assert np.linalg.norm(R) / np.linalg.norm(X) < WARN_U * U
    assert O.orthogonality_u(Q) < WARN_U
    # device API with an odd ld takes the unaligned gemv
    tA, tQ = to_device(A0), to_device(O.identity(n, ld=ld))
    assert node.hessenberg_device(tA, tQ, n=n) == 0
    A2 = to_host(tA)
    assert np.abs(A2[:n] - A[:n]).max() <= elementwise_tolerance(n) * np.linalg.norm(A0[:n])


def test_lcg_on_device_is_bit_exact(node):
    """The in-HBM generator reproduces the reference LCG stream bit for bit."""
    n = 333
    t = node.device_matrix(n)
    assert node.lcg_fill_device(t, n, n, seed=2019, mode=0) == 0
    A0 = O.random_fullpos(n)
    assert np.array_equal(to_host(t)[:n], A0[:n])
    assert node.lcg_fill_device(t, n, n, seed=7, mode=1) == 0
    O.lib().oracle_init_prand(7)
    ref = np.zeros((n, n), order="F")
    O.lib().oracle_fill_random_full(n, n, O.oracle._p(ref), n)
    assert np.array_equal(to_host(t)[:n], ref)


def test_device_checks_agree_with_oracle_checks(node):
    n = 300
    A0 = O.random_fullpos(n)
    tA, tQ, tA0 = to_device(A0), to_device(O.identity(n)), to_device(A0)
    assert node.hessenberg_device(tA, tQ, n=n) == 0
    rc, chk = node.check_device(tQ, tA, tA0, n=n)
    assert rc == 0 and chk["below_subdiagonal"] == 0
    A, Q = to_host(tA), to_host(tQ)
    assert chk["residual_u"] == pytest.approx(O.residual_u(Q, A, A0), rel=0.5)
    assert chk["orthogonality_u"] == pytest.approx(O.orthogonality_u(Q), rel=0.5)


@pytest.mark.parametrize("n", [2000, 4000, 8000])
def test_baseline_sizes_properties(node, n):
    """BASELINE configs 1-2 (n=2000, n=8000): size-independent acceptance checks of the
    reference (test/common/hooks.c:434-456, checks.c:180-208) on the LCG input, all on
    the GPU; limits = the reference's own measured residuals x 1.5 (docs/_7_test_driver.md)."""
    tA0 = node.device_matrix(n)
    assert node.lcg_fill_device(tA0, n, n, seed=2019, mode=0) == 0
    tA = tA0.clone()
    tQ = node.device_matrix(n)
    node.set_matrix_device(tQ, n, n, 0.0, 1.0)
    rc, st = node.hessenberg_device(tA, tQ, n=n, stats=True)
    assert rc == 0
    rc, chk = node.check_device(tQ, tA, tA0, n=n)
    assert rc == 0
    assert chk["below_subdiagonal"] == 0
    assert chk["residual_u"] < REF_RESIDUAL_U
    assert chk["orthogonality_u"] < REF_ORTH_U
    # similarity invariants: trace and Frobenius norm are preserved
    import torch
    tr0 = torch.diagonal(tA0[:, :n]).sum().item(); tr1 = torch.diagonal(tA[:, :n]).sum().item()
    assert abs(tr0 - tr1) <= 1e-11 * abs(tr0)
    f0 = torch.linalg.norm(tA0[:, :n]).item(); f1 = torch.linalg.norm(tA[:, :n]).item()
    assert abs(f0 - f1) <= 1e-12 * f0


def test_problem_larger_than_the_device_is_an_error_code_not_an_abort(node):
    """two 200000 x 200000 matrices are 640 GB: the host-array entry points see that before they
    allocate anything and return STARNEIG_GENERIC_ERROR (the arrays are never touched)"""
    n = 200000
    stub = np.zeros(16)
    L = node.lib.load()
    p = stub.ctypes.data
    assert L.starneig_SEP_SM_Hessenberg(n, p, n, p, n) == node.GENERIC_ERROR
    assert L.starneig_SEP_SM_Schur(n, p, n, p, n, None, None) == node.GENERIC_ERROR
    # the library is still usable
    m = 64
    A0 = O.random_fullpos(m)
    A = A0.copy(order="F"); Q = O.identity(m)
    assert node.SEP_SM_Hessenberg(m, A, A.shape[0], Q, Q.shape[0]) == 0
    assert O.residual_u(Q, A, A0) < 50


@pytest.mark.parametrize("n,ld,partial", [(2000, None, False), (777, 781, False), (3000, None, True)])
def test_two_reductions_of_one_matrix_give_the_same_bits(node, n, ld, partial):
    """the Hessenberg reduction is bit-reproducible.  The cross-workgroup sums of the
    column chain (w, w_v, the norm) are added up in workgroup order by the last workgroup of a slot
    (csrc/hessenberg.hip slot_fold), the slices of the split-K products in slice order
    (csrc/dgemm_mfma.hip dgemm_splitk_sum_kernel); rounds 1-5 used fp64 atomics for both -- the reference's
    STARPU_COMMUTE accumulations, hessenberg/tasks.c:374,515,622 -- and two runs differed in the last bits.
    Aligned and odd leading dimension (the 8-byte gemv), a partial range, three runs each; the matrices of other
    sizes in between make the cached workspaces hold stale data of another shape."""
    A0 = O.random_fullpos(n, ld=ld)
    out = []
    for rep in range(3):
        if partial:
            A, Q = run_host_api(node, partial_input(n, 100, n - 50), 100, n - 50)
        else:
            A, Q = run_host_api(node, A0)
        out.append((A, Q))
        run_host_api(node, O.random_fullpos(300 + 211 * rep, seed=7 + rep))
    for A, Q in out[1:]:
        assert np.array_equal(A, out[0][0]) and np.array_equal(Q, out[0][1])
