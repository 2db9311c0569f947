"""ctypes front end of liboracle.so (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-B", "-C", _HERE, "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()        # rebuilds only when a source is newer than the library
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        L.oracle_init_prand.argtypes = [C.c_uint]
        L.oracle_prand.restype = C.c_int
        L.oracle_fill_random_fullpos.argtypes = [C.c_int, C.c_int, dp, C.c_int]
        L.oracle_fill_random_full.argtypes = [C.c_int, C.c_int, dp, C.c_int]
        L.oracle_fill_random_hessenberg.argtypes = [C.c_int, dp, C.c_int]
        L.oracle_hessenberg.argtypes = [C.c_int] * 4 + [dp, C.c_int, dp, C.c_int]
        L.oracle_hessenberg.restype = C.c_int
        L.oracle_default_panel_width.argtypes = [C.c_int]
        L.oracle_default_panel_width.restype = C.c_int
        L.oracle_residual_u.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int]
        L.oracle_residual_u.restype = C.c_double
        L.oracle_orthogonality_u.argtypes = [C.c_int, dp, C.c_int]
        L.oracle_orthogonality_u.restype = C.c_double
        L.oracle_count_below_subdiagonal.argtypes = [C.c_int, dp, C.c_int]
        L.oracle_count_below_subdiagonal.restype = C.c_long
        L.oracle_dlarfg.argtypes = [C.c_int, dp, dp]
        L.oracle_dlarfg.restype = C.c_double
        L.oracle_schur.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, dp, dp]
        L.oracle_schur.restype = C.c_int
        L.oracle_extract_eigenvalues.argtypes = [C.c_int, dp, C.c_int, dp, dp]
        L.oracle_check_schur_form.argtypes = [C.c_int, dp, C.c_int]
        L.oracle_check_schur_form.restype = C.c_int
        L.oracle_fill_random_uptriag.argtypes = [C.c_int, dp, C.c_int]
        L.oracle_gep_schur.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int, dp, dp, dp]
        L.oracle_gep_schur.restype = C.c_int
        L.oracle_gep_extract_eigenvalues.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, dp, dp, dp]
        L.oracle_check_gep_schur_form.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int]
        L.oracle_check_gep_schur_form.restype = C.c_int
        L.oracle_ht_qr.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int]
        L.oracle_ht_qr.restype = C.c_int
        L.oracle_ht_reduce.argtypes = [C.c_int] + [dp, C.c_int] * 4
        L.oracle_ht_reduce.restype = C.c_int
        L.oracle_hessenberg_triangular.argtypes = [C.c_int] + [dp, C.c_int] * 4
        L.oracle_hessenberg_triangular.restype = C.c_int
        L.oracle_lartg.argtypes = [C.c_double, C.c_double, dp, dp, dp]
        ip = C.POINTER(C.c_int)
        L.oracle_known_spectrum.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, dp, dp, dp, C.c_int]
        L.oracle_uniform_select.argtypes = [C.c_int, dp, C.c_int, C.c_double, ip]
        L.oracle_place_blocks.argtypes = [C.c_int, dp, dp, dp, dp, C.c_int, dp, C.c_int]
        L.oracle_householder_vector.argtypes = [C.c_int, dp]
        L.oracle_decouple.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, C.c_int]
        L.oracle_known_eigenvalues_check.argtypes = [C.c_int] + [dp] * 6 + [C.c_double, C.c_double, dp, ip]
        L.oracle_eigenvalues_check.argtypes = [C.c_int] + [dp] * 6 + [C.c_double, C.c_double, dp, ip]
        L.oracle_msqr_port.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, dp]
        L.oracle_msqr_port.restype = C.c_int
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_set_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    assert a.dtype == np.float64 and a.flags.f_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_double))


def ld_for(n):
    """test/common/common.c:99 -- leading dimension rounded up to 8 doubles."""
    return (n + 7) // 8 * 8


def random_fullpos(n, seed=2019, ld=None):
    """The test driver's default Hessenberg input (test/hessenberg/experiment.c:102)."""
    ld = ld or ld_for(n)
    A = np.zeros((ld, n), order="F")
    lib().oracle_init_prand(seed)
    lib().oracle_fill_random_fullpos(n, n, _p(A), ld)
    return A


def random_hessenberg(n, seed=2019, ld=None):
    ld = ld or ld_for(n)
    A = np.zeros((ld, n), order="F")
    lib().oracle_init_prand(seed)
    lib().oracle_fill_random_hessenberg(n, _p(A), ld)
    return A


def identity(n, ld=None):
    ld = ld or ld_for(n)
    Q = np.zeros((ld, n), order="F")
    Q[np.arange(n), np.arange(n)] = 1.0
    return Q


def default_panel_width(n):
    return lib().oracle_default_panel_width(n)


def hessenberg(A, Q, begin=0, end=None, panel_width=None):
    """In-place oracle reduction; A, Q are (ld, n) Fortran arrays."""
    n = A.shape[1]
    end = n if end is None else end
    pw = panel_width or default_panel_width(n)
    rc = lib().oracle_hessenberg(n, begin, end, pw, _p(A), A.shape[0], _p(Q), Q.shape[0])
    if rc != 0:
        raise MemoryError("oracle_hessenberg")
    return A, Q


def residual_u(Q, H, A):
    n = A.shape[1]
    return lib().oracle_residual_u(n, _p(Q), Q.shape[0], _p(H), H.shape[0], _p(A), A.shape[0])


def orthogonality_u(Q):
    n = Q.shape[1]
    return lib().oracle_orthogonality_u(n, _p(Q), Q.shape[0])


def count_below_subdiagonal(H):
    n = H.shape[1]
    return lib().oracle_count_below_subdiagonal(n, _p(H), H.shape[0])


def schur(H, Z):
    """In-place oracle Schur reduction (double-shift QR) of the upper Hessenberg (ld, n)
    array H; Z <- Z*U.  Returns (real, imag)."""
    n = H.shape[1]
    wr = np.zeros(n); wi = np.zeros(n)
    rc = lib().oracle_schur(n, _p(H), H.shape[0], _p(Z), Z.shape[0],
                            wr.ctypes.data_as(C.POINTER(C.c_double)),
                            wi.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise RuntimeError(f"oracle_schur did not converge (row {rc - 1})")
    return wr, wi


def extract_eigenvalues(S):
    n = S.shape[1]
    wr = np.zeros(n); wi = np.zeros(n)
    lib().oracle_extract_eigenvalues(n, _p(S), S.shape[0],
                                     wr.ctypes.data_as(C.POINTER(C.c_double)),
                                     wi.ctypes.data_as(C.POINTER(C.c_double)))
    return wr, wi


def check_schur_form(S):
    return lib().oracle_check_schur_form(S.shape[1], _p(S), S.shape[0])


def random_pencil(n, seed=2019, ld=None):
    """The test driver's random Hessenberg-triangular pencil (test/schur/experiment.c:203-207):
    generate_random_hessenberg, then generate_random_uptriag on the same LCG stream."""
    ld = ld or ld_for(n)
    H = np.zeros((ld, n), order="F")
    R = np.zeros((ld, n), order="F")
    lib().oracle_init_prand(seed)
    lib().oracle_fill_random_hessenberg(n, _p(H), ld)
    lib().oracle_fill_random_uptriag(n, _p(R), ld)
    return H, R


def random_pencil_wellcond(n, seed=2019, ld=None):
    """Same LCG data with a well-conditioned triangular factor (tests/golden/make_golden_gep.py):
    strict upper part / sqrt(n), diagonal 1 + |r_ii| -- eigenvalues comparable to ~1e4 u."""
    H, R = random_pencil(n, seed, ld)
    Rw = np.zeros_like(R)
    Rw[:n] = np.triu(R[:n], 1) / np.sqrt(n) + np.diag(1.0 + np.abs(np.diag(R[:n])))
    return H, Rw


def gep_schur(H, R, Q, Z):
    """In place: (H, R) -> generalized real Schur form, Q <- Q U1, Z <- Z U2.
    Returns (info, alpha_r, alpha_i, beta)."""
    n = H.shape[1]
    ar, ai, be = np.zeros(n), np.zeros(n), np.zeros(n)
    dp = C.POINTER(C.c_double)
    info = lib().oracle_gep_schur(n, _p(H), H.shape[0], _p(R), R.shape[0], _p(Q), Q.shape[0],
                                  _p(Z), Z.shape[0], ar.ctypes.data_as(dp), ai.ctypes.data_as(dp),
                                  be.ctypes.data_as(dp))
    return info, ar, ai, be


def gep_extract_eigenvalues(S, T):
    n = S.shape[1]
    ar, ai, be = np.zeros(n), np.zeros(n), np.zeros(n)
    dp = C.POINTER(C.c_double)
    lib().oracle_gep_extract_eigenvalues(n, _p(S), S.shape[0], _p(T), T.shape[0],
                                         ar.ctypes.data_as(dp), ai.ctypes.data_as(dp), be.ctypes.data_as(dp))
    return ar, ai, be


def dlag2(A, B):
    """LAPACK DLAG2 restated (oracle/gep_oracle.c:oracle_dlag2): (scale1, scale2, wr1, wr2, wi) of the
    2 x 2 pencil (A, B), B upper triangular."""
    a = np.asfortranarray(A, dtype=np.float64); b = np.asfortranarray(B, dtype=np.float64)
    out = np.zeros(5)
    dp = C.POINTER(C.c_double)
    L = lib()
    L.oracle_dlag2.argtypes = [dp, C.c_int, dp, C.c_int, C.c_double, dp, dp, dp, dp, dp]
    L.oracle_dlag2.restype = None
    o = out.ctypes.data_as(dp)
    L.oracle_dlag2(a.ctypes.data_as(dp), 2, b.ctypes.data_as(dp), 2, float(np.finfo(np.float64).tiny),
                   C.cast(C.addressof(o.contents), dp), C.cast(C.addressof(o.contents) + 8, dp),
                   C.cast(C.addressof(o.contents) + 16, dp), C.cast(C.addressof(o.contents) + 24, dp),
                   C.cast(C.addressof(o.contents) + 32, dp))
    return out


def check_gep_schur_form(S, T):
    return lib().oracle_check_gep_schur_form(S.shape[1], _p(S), S.shape[0], _p(T), T.shape[0])


def pencil_residual_u(Q, S, Z, A):
    """2^52 ||Q S Z^T - A||_F / ||A||_F (test/common/checks.c two-sided residual)."""
    n = A.shape[1]
    return float(np.linalg.norm(Q[:n] @ S[:n] @ Z[:n].T - A[:n]) / np.linalg.norm(A[:n]) * 2.0 ** 52)


def match_eigenvalues(ev_a, ev_b):
    """Greedy nearest-neighbour matching of two eigenvalue multisets, like the reference's
    known-eigenvalues hook (test/common/hooks.c:1178-1250).  Returns the largest relative
    difference |a-b| / max(|b|, tiny) over the matching, in units of u = 2^-52."""
    a = np.asarray(ev_a, dtype=complex).copy()
    b = list(np.asarray(ev_b, dtype=complex))
    worst = 0.0
    scale = max(np.abs(a).max(), 1e-300)
    order = np.argsort(-np.abs(a))
    for idx in order:
        x = a[idx]
        d = np.abs(np.array(b) - x)
        k = int(np.argmin(d))
        worst = max(worst, d[k] / max(abs(x), 1e-3 * scale))
        b.pop(k)
    return worst / 2.0 ** -52


def random_fullpos_pair(n, seed=2019, ld=None):
    """The test driver's generalized Hessenberg input: A, then B, both generate_random_fullpos
    on one LCG stream (test/hessenberg/experiment.c:102-106)."""
    ld = ld or ld_for(n)
    A = np.zeros((ld, n), order="F")
    B = np.zeros((ld, n), order="F")
    lib().oracle_init_prand(seed)
    lib().oracle_fill_random_fullpos(n, n, _p(A), ld)
    lib().oracle_fill_random_fullpos(n, n, _p(B), ld)
    return A, B


def ht_qr(A, B, Q):
    """B = Q0 R; A <- Q0^T A; Q <- Q Q0; B <- R (wrappers/lapack.c:143-160). In place."""
    n = A.shape[1]
    rc = lib().oracle_ht_qr(n, _p(A), A.shape[0], _p(B), B.shape[0], _p(Q), Q.shape[0])
    assert rc == 0
    return A, B, Q


def ht_reduce(A, B, Q, Z):
    """dgghrd-ordered rotations on (A, triangular B). In place."""
    n = A.shape[1]
    rc = lib().oracle_ht_reduce(n, _p(A), A.shape[0], _p(B), B.shape[0], _p(Q), Q.shape[0], _p(Z), Z.shape[0])
    assert rc == 0
    return A, B, Q, Z


def hessenberg_triangular(A, B, Q, Z):
    """starneig_GEP_SM_HessenbergTriangular restated (wrappers/lapack.c:45-176). In place."""
    n = A.shape[1]
    rc = lib().oracle_hessenberg_triangular(n, _p(A), A.shape[0], _p(B), B.shape[0], _p(Q), Q.shape[0],
                                            _p(Z), Z.shape[0])
    assert rc == 0
    return A, B, Q, Z


def lartg(f, g):
    c, s, r = C.c_double(), C.c_double(), C.c_double()
    lib().oracle_lartg(f, g, C.byref(c), C.byref(s), C.byref(r))
    return c.value, s.value, r.value


def count_below_diagonal(T):
    n = T.shape[1]
    return int(np.count_nonzero(np.tril(T[:n, :n], -1)))


# ---- the reference TEST DRIVER's Schur experiments (oracle/testdriver_oracle.c) -------------

def _householder_apply(v, S, Zv=None):
    """mul_QAZT(Q, S, Z) = Q S Z^T for Q = I - 2 v v^T, Z = I - 2 z z^T (test/common/init.c:543-550);
    the reference forms Q and multiplies, this is the same product without the n^3."""
    z = v if Zv is None else Zv
    X = S - 2.0 * np.outer(v, v @ S)
    return np.asfortranarray(X - 2.0 * np.outer(X @ z, z))


def householder_matrix(n, ld=None):
    """generate_random_householder (test/common/init.c:523-541): I - 2 v v^T, v from the LCG."""
    ld = ld or ld_for(n)
    v = np.zeros(n)
    lib().oracle_householder_vector(n, _p(v))
    Q = np.zeros((ld, n), order="F")
    Q[:n] = np.eye(n) - 2.0 * np.outer(v, v)
    return Q, v


def known_pencil(n, generalized=False, seed=2019, complex_ratio=0.5, zero_ratio=0.01, inf_ratio=0.01,
                 ld=None):
    """`starneig-test --experiment schur --init known [--generalized]` before its Hessenberg
    step (test/schur/experiment.c:295-353): returns the dense A (and B), and the prescribed
    eigenvalues (real, imag, beta) as the reference's extract_eigenvalues reads them off the
    generating Schur form."""
    ld = ld or ld_for(n)
    L = lib()
    L.oracle_init_prand(seed)
    S = np.zeros((ld, n), order="F")
    L.oracle_fill_random_uptriag(n, _p(S), ld)
    T = None
    if generalized:
        T = identity(n, ld)
    real, imag, beta = np.zeros(n), np.zeros(n), np.zeros(n)
    L.oracle_known_spectrum(n, int(generalized), complex_ratio, zero_ratio, inf_ratio,
                            _p(real), _p(imag), _p(beta), 0)
    L.oracle_place_blocks(n, _p(real), _p(imag), _p(beta), _p(S), ld,
                          _p(T) if generalized else None, ld)
    if generalized:
        kr, ki, kb = gep_extract_eigenvalues(S, T)
    else:
        kr, ki = extract_eigenvalues(S)
        kb = np.ones(n)
    q = np.zeros(n)
    L.oracle_householder_vector(n, _p(q))
    A = np.zeros((ld, n), order="F")
    if generalized:
        z = np.zeros(n)
        L.oracle_householder_vector(n, _p(z))
        A[:n] = _householder_apply(q, S[:n], z)
        B = np.zeros((ld, n), order="F")
        B[:n] = _householder_apply(q, T[:n], z)
        return A, B, kr, ki, kb
    A[:n] = _householder_apply(q, S[:n])
    return A, None, kr, ki, kb


def schur_random_input(n, generalized=False, decouple=0, set_to_inf=0, seed=2019, ld=None):
    """`starneig-test --experiment schur [--generalized] [--decouple k] [--set-to-inf k]`, the
    default `random` initializer (test/schur/experiment.c:181-214): random upper Hessenberg H,
    Q a random Householder matrix, (B random upper triangular, Z a second Householder matrix),
    then the cuts.  Returns H, Q, B, Z (B, Z None in the standard case)."""
    ld = ld or ld_for(n)
    L = lib()
    L.oracle_init_prand(seed)
    H = np.zeros((ld, n), order="F")
    L.oracle_fill_random_hessenberg(n, _p(H), ld)
    Q, _ = householder_matrix(n, ld)
    B = Z = None
    if generalized:
        B = np.zeros((ld, n), order="F")
        L.oracle_fill_random_uptriag(n, _p(B), ld)
        Z, _ = householder_matrix(n, ld)
    if decouple > 0 or set_to_inf > 0:
        L.oracle_decouple(n, decouple, set_to_inf, _p(H), ld, _p(B) if generalized else None, ld)
    return H, Q, B, Z


def _hook(fn, a, b, warn, fail):
    n = len(a[0])
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (*a, *b)]
    out = np.zeros(3)
    counts = (C.c_int * 2)()
    fn(n, *[_p(x) for x in arrs], float(warn), float(fail), _p(out), counts)
    return {"mean_u": out[0], "min_u": out[1], "max_u": out[2], "warnings": counts[0], "failures": counts[1]}


def known_eigenvalues_check(computed, known, warn=1e4, fail=1e6):
    """The reference's `known-eigenvalues` hook with its default thresholds
    (test/common/hooks.c:1071-1072,1178-1296). computed / known = (real, imag, beta)."""
    return _hook(lib().oracle_known_eigenvalues_check, computed, known, warn, fail)


def eigenvalues_check(extracted, returned, warn=1e3, fail=1e4):
    """The reference's `eigenvalues` hook (test/common/hooks.c:787-788,891-991): eigenvalues read
    off the diagonal blocks of the result against those the solver returned, by position."""
    return _hook(lib().oracle_eigenvalues_check, extracted, returned, warn, fail)


# ---- CPU port of the multishift QR / AED Schur leg (oracle/msqr_port.c; bench.py's cpu_baseline) ----

def set_threads(n):
    """OpenMP threads of the oracle's parallel regions from here on (n <= 0: leave as is); returns the count in force."""
    return int(lib().oracle_set_threads(int(n)))


def msqr_port(H, Q, aed_fn, small_fn, nw=None, ns=None, W=128, small_limit=128):
    """In place: the upper Hessenberg (ld, n) array H -> real Schur form, Q <- Q U, by the multi-threaded
    host restatement of the reference's Schur leg.  aed_fn / small_fn: C function pointers of the AED window
    kernel and the small-block solver (the host-only kernels of the product's test library).  Returns
    (rc, real, imag, stats)."""
    n = H.shape[1]
    # the reference's defaults (schur/process_args.c:116-162 through LAPACK's iparmq table): here the
    # window the host kernel is fastest at
    nw = nw or min(n, max(64, min(288, 136 + int(0.006 * n))))
    ns = ns or max(2, (nw * 5 // 8) // 2 * 2)
    st = np.zeros(4)
    rc = lib().oracle_msqr_port(n, _p(H), H.shape[0], _p(Q), Q.shape[0], nw, ns, W, small_limit,
                                C.cast(aed_fn, C.c_void_p), C.cast(small_fn, C.c_void_p), _p(st))
    wr, wi = extract_eigenvalues(H)
    return rc, wr, wi, {"sweeps": int(st[0]), "aeds": int(st[1]), "update_flops": st[2], "aed_s": st[3]}


def reorder_input(n, seed=2019, fortify=True, complex_ratio=0.5, select_ratio=0.35, ld=None):
    """`starneig-test --experiment reorder [--fortify]` (test/common/init_schur.c:122-181): a random upper
    triangular S with the blocks of `complex_distr uniform` on its diagonal (with --fortify: eigenvalues two
    apart), Q a random Householder matrix, and the selection of `select_distr uniform` (35 % of the rows, whole
    blocks).  Returns S, Q, selected (int32), and the prescribed eigenvalues (real, imag) in diagonal order."""
    ld = ld or ld_for(n)
    L = lib()
    L.oracle_init_prand(seed)
    S = np.zeros((ld, n), order="F")
    L.oracle_fill_random_uptriag(n, _p(S), ld)
    real, imag, beta = np.zeros(n), np.zeros(n), np.zeros(n)
    L.oracle_known_spectrum(n, 0, complex_ratio, 0.01, 0.01, _p(real), _p(imag), _p(beta), int(fortify))
    L.oracle_place_blocks(n, _p(real), _p(imag), _p(beta), _p(S), ld, None, ld)
    Q, _ = householder_matrix(n, ld)
    sel = np.zeros(n, dtype=np.int32)
    L.oracle_uniform_select(n, _p(S), ld, select_ratio, sel.ctypes.data_as(C.POINTER(C.c_int)))
    kr, ki = extract_eigenvalues(S)
    return S, Q, sel, kr, ki
