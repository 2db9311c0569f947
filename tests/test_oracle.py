"""CPU suite, part 1: the oracle against the committed golden vectors (LAPACK
dgehrd, the reference test driver's own comparator) and against the reference's
invariant checks."""
import os

import numpy as np
import pytest

import oracle as O
from helpers import U, WARN_U, elementwise_tolerance

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("n", [64, 200, 512, 2000])
def test_oracle_matches_lapack_golden(n):
    g = np.load(os.path.join(GOLDEN, f"hessenberg_lcg2019_n{n}.npz"))
    A0 = O.random_fullpos(n, seed=int(g["seed"]))
    # the LCG restatement reproduces the fixture's input bit for bit
    assert np.array_equal(A0[:n, 0], g["a_first_col"])
    assert np.array_equal(A0[:n, -1], g["a_last_col"])
    A = A0.copy(order="F")
    Q = O.identity(n)
    O.hessenberg(A, Q)
    H = A[:n]
    tol = elementwise_tolerance(n) * float(g["a_fro"])
    assert np.abs(np.diag(H, -1) - g["h_subdiag"]).max() <= tol
    assert np.array_equal(np.sign(np.diag(H, -1)), np.sign(g["h_subdiag"]))
    assert np.abs(np.diag(H) - g["h_diag"]).max() <= tol
    assert np.abs(H[0, :] - g["h_first_row"]).max() <= tol
    assert np.abs(H[:, -1] - g["h_last_col"]).max() <= tol
    assert abs(np.linalg.norm(H) - float(g["h_fro"])) <= tol
    # reference invariants: exact zeros, residuals below the warn threshold
    assert O.count_below_subdiagonal(A) == 0
    assert O.residual_u(Q, A, A0) < WARN_U
    assert O.orthogonality_u(Q) < WARN_U


def partial_input(n, begin, end, seed=2019):
    """test/misc/partial_hessenberg.c:144-156: random upper triangular matrix whose
    diagonal block [begin,end) is dense below the diagonal."""
    rng = np.random.RandomState(seed)
    A = O.identity(n) * 0.0
    A[:n] = np.triu(rng.uniform(-1, 1, (n, n)))
    blk = rng.uniform(-1, 1, (end - begin, end - begin))
    A[begin:end, begin:end] += np.tril(blk, -1)
    return np.asfortranarray(A)


@pytest.mark.parametrize("n,begin,end", [(47, 3, 40), (88, 0, 50), (88, 20, 88)])
def test_oracle_partial_range(n, begin, end):
    """_expert(begin,end): similarity is preserved, columns [begin,end) are reduced,
    the rest of the structure is untouched (reference test/misc/partial_hessenberg.c)."""
    A0 = partial_input(n, begin, end)
    A = A0.copy(order="F")
    Q = O.identity(n)
    O.hessenberg(A, Q, begin=begin, end=end, panel_width=16)
    H = A[:n]
    # test/misc/partial_hessenberg.c:172-176: Hessenberg inside [begin,end-1), triangular outside
    for c in range(n - 1):
        k = 2 if begin <= c < end - 1 else 1
        assert np.all(H[c + k:, c] == 0.0)
    assert O.residual_u(Q, A, A0) < WARN_U
    assert O.orthogonality_u(Q) < WARN_U
    # rows/columns outside [begin,end) of Q stay identity
    Qn = Q[:n]
    assert np.array_equal(Qn[:begin + 1, :begin + 1], np.eye(begin + 1))


@pytest.mark.parametrize("pw", [8, 35, 64])
def test_oracle_panel_width_independent(pw):
    n = 150
    A0 = O.random_fullpos(n)
    ref = A0.copy(order="F"); Qr = O.identity(n); O.hessenberg(ref, Qr, panel_width=280)
    A = A0.copy(order="F"); Q = O.identity(n); O.hessenberg(A, Q, panel_width=pw)
    assert np.abs(A[:n] - ref[:n]).max() <= elementwise_tolerance(n) * np.linalg.norm(A0[:n])


def test_dlarfg_known_answers():
    """LAPACK dlarfg semantics: beta = -sign(alpha)*norm, H*x = beta*e1, tau in [1,2]."""
    import ctypes as C
    L = O.lib()
    for alpha0, x0 in [(3.0, [4.0]), (-1.0, [2.0, 2.0]), (0.0, [1.0, 0.0, 0.0]), (5.0, [0.0, 0.0])]:
        alpha = C.c_double(alpha0)
        x = np.array(x0, dtype=np.float64)
        tau = L.oracle_dlarfg(len(x0) + 1, C.byref(alpha), x.ctypes.data_as(C.POINTER(C.c_double)))
        full = np.array([alpha0] + x0)
        nrm = np.linalg.norm(full)
        if np.linalg.norm(x0) == 0.0:
            assert tau == 0.0 and alpha.value == alpha0
            continue
        assert alpha.value == pytest.approx(-np.copysign(nrm, alpha0) if alpha0 != 0 else -nrm)
        v = np.concatenate([[1.0], x])
        Hx = full - tau * v * (v @ full)
        assert np.abs(Hx[1:]).max() < 1e-15 * nrm
        assert Hx[0] == pytest.approx(alpha.value)
        assert 1.0 <= tau <= 2.0


def test_default_panel_width():
    # hessenberg/interface.c:74-78: 280 @ 2000, 288 @ 4000/8000, 312 @ 20000
    assert [O.default_panel_width(n) for n in (2000, 4000, 8000, 20000)] == [280, 288, 288, 312]


@pytest.mark.parametrize("n", [64, 200, 512, 2000])
def test_schur_oracle_matches_lapack_golden_eigenvalues(n):
    """Schur leg of the oracle (double-shift QR restatement) against numpy/LAPACK eigenvalues of
    the same LCG matrix (fixture), plus the reference's Schur-form / residual checks."""
    g = np.load(os.path.join(GOLDEN, f"hessenberg_lcg2019_n{n}.npz"))
    A0 = O.random_fullpos(n, seed=int(g["seed"]))
    H = A0.copy(order="F"); Q = O.identity(n)
    O.hessenberg(H, Q)
    wr, wi = O.schur(H, Q)
    assert O.check_schur_form(H) == 0
    assert O.residual_u(Q, H, A0) < WARN_U
    assert O.orthogonality_u(Q) < WARN_U
    ev = g["eig_real"] + 1j * g["eig_imag"]
    assert O.match_eigenvalues(wr + 1j * wi, ev) < 1e4          # reference warn level
    er, ei = O.extract_eigenvalues(H)
    assert np.array_equal(er, wr) and np.array_equal(ei, wi)


def test_dlanv2_known_answers():
    """LAPACK dlanv2 semantics on hand-checked blocks."""
    import ctypes as C
    L = O.lib()
    L.oracle_dlanv2.argtypes = [C.POINTER(C.c_double)] * 10
    def run(a, b, c, d):
        v = [C.c_double(x) for x in (a, b, c, d)] + [C.c_double() for _ in range(6)]
        L.oracle_dlanv2(*[C.byref(x) for x in v])
        return [x.value for x in v]
    # already upper triangular
    a, b, c, d, r1, i1, r2, i2, cs, sn = run(1.0, 2.0, 0.0, 3.0)
    assert (a, b, c, d, r1, i1, r2, i2, cs, sn) == (1.0, 2.0, 0.0, 3.0, 1.0, 0.0, 3.0, 0.0, 1.0, 0.0)
    # complex pair: rotation matrix scaled -> standard form keeps equal diagonal, b*c < 0
    a, b, c, d, r1, i1, r2, i2, cs, sn = run(1.0, -2.0, 2.0, 1.0)
    assert a == d == 1.0 and b * c < 0 and (r1, r2) == (1.0, 1.0) and abs(i1 - 2.0) < 1e-15 and i2 == -i1
    # real eigenvalues 1 and 4 of [[2,1],[2,3]]: c is annihilated
    a, b, c, d, r1, i1, r2, i2, cs, sn = run(2.0, 1.0, 2.0, 3.0)
    assert c == 0.0 and i1 == 0.0 and sorted([round(r1, 12), round(r2, 12)]) == [1.0, 4.0]
    assert abs(cs * cs + sn * sn - 1.0) < 1e-15


def test_multishift_port_of_the_schur_leg():
    """oracle/msqr_port.c (bench.py's cpu_baseline): multishift QR with AED on the host, driving the
    host-only window kernels of the product's test library -- a valid Schur decomposition with the
    eigenvalues LAPACK finds"""
    import starneig_amd as S
    L = S.lib.load_test_hooks()
    n = 500
    A0 = O.random_fullpos(n)
    H = A0.copy(order="F"); Q = O.identity(n)
    O.hessenberg(H, Q)
    rc, wr, wi, st = O.msqr_port(H, Q, L.sn_internal_aed_window, L.sn_internal_small_schur, nw=96, ns=60, W=64,
                                 small_limit=64)
    assert rc == 0 and st["sweeps"] > 0 and st["aeds"] > 0
    assert O.check_schur_form(H) == 0
    assert O.residual_u(Q, H, A0) < 500 and O.orthogonality_u(Q) < 500
    assert O.match_eigenvalues(wr + 1j * wi, np.linalg.eigvals(A0[:n])) < 1e4
