"""ctypes binding of libstarneig_amd.so -- the C-ABI drop-in boundary.

The functions mirror the reference's C interface one to one (same names without
the ``starneig_`` prefix, same argument order, same return codes):
``SEP_SM_Hessenberg`` <-> reference src/include/starneig/sep_sm.h:89-92, etc.
There is no fallback of any kind: a missing library raises at import of the
symbol table, a missing GPU aborts in ``node_init``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstarneig_amd.so")

# error codes, include/starneig/error.h
SUCCESS = 0
GENERIC_ERROR = 1
NOT_INITIALIZED = 2
INVALID_CONFIGURATION = 3
INVALID_ARGUMENTS = 4
DID_NOT_CONVERGE = 6
PARTIAL_REORDERING = 7

USE_ALL = -1
DEFAULT = 0x0
HINT_SM = 0x0
NO_VERBOSE = 0x10
NO_MESSAGES = 0x30


class HessenbergConf(C.Structure):
    _fields_ = [("tile_size", C.c_int), ("panel_width", C.c_int)]


class SchurConf(C.Structure):
    _fields_ = [(k, C.c_int) for k in (
        "iteration_limit", "tile_size", "small_limit", "aed_window_size", "aed_nibble",
        "aed_parallel_soft_limit", "aed_parallel_hard_limit", "shift_origin", "shift_count",
        "window_size", "shifts_per_window", "update_width", "update_height")] + [
        ("left_threshold", C.c_double), ("right_threshold", C.c_double),
        ("inf_threshold", C.c_double)]


class ReorderConf(C.Structure):
    _fields_ = [(k, C.c_int) for k in (
        "plan", "blueprint", "tile_size", "values_per_chain", "window_size", "small_window_size",
        "small_window_threshold", "update_width", "update_height")]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# every symbol include/*.h declares, with its signature
SIGNATURES = {
    "starneig_node_init": (None, [C.c_int, C.c_int, C.c_uint]),
    "starneig_node_initialized": (C.c_int, []),
    "starneig_node_get_cores": (C.c_int, []),
    "starneig_node_set_cores": (None, [C.c_int]),
    "starneig_node_get_gpus": (C.c_int, []),
    "starneig_node_set_gpus": (None, [C.c_int]),
    "starneig_node_finalize": (None, []),
    "starneig_node_enable_pinning": (None, []),
    "starneig_node_disable_pinning": (None, []),
    "starneig_hessenberg_init_conf": (None, [C.POINTER(HessenbergConf)]),
    "starneig_schur_init_conf": (None, [C.POINTER(SchurConf)]),
    "starneig_SEP_SM_Hessenberg": (C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int]),
    "starneig_SEP_SM_Hessenberg_expert": (
        C.c_int, [C.POINTER(HessenbergConf), C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int]),
    "starneig_SEP_SM_Schur": (C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp]),
    "starneig_SEP_SM_Schur_expert": (
        C.c_int, [C.POINTER(SchurConf), C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp]),
    "starneig_SEP_SM_Reduce": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "starneig_SEP_SM_Select": (C.c_int, [C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "starneig_reorder_init_conf": (None, [C.POINTER(ReorderConf)]),
    "starneig_SEP_SM_ReorderSchur": (C.c_int, [C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp]),
    "starneig_SEP_SM_ReorderSchur_expert": (
        C.c_int, [C.POINTER(ReorderConf), C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp]),
    "starneig_amd_reorder_schur_device": (
        C.c_int, [C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, C.POINTER(ReorderConf), _vp, _dp]),
    "starneig_GEP_SM_Schur": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp]),
    "starneig_GEP_SM_Schur_expert": (
        C.c_int, [C.POINTER(SchurConf), C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp,
                  C.c_int, _vp, _vp, _vp]),
    "starneig_GEP_SM_HessenbergTriangular": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int]),
    "starneig_GEP_SM_Reduce": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp,
                  _vp, _vp, _vp, _vp]),
    "starneig_amd_hessenberg_triangular_device": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _dp]),
    "starneig_amd_gep_schur_device": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp,
                  C.POINTER(SchurConf), _vp, _dp]),
    "starneig_amd_lcg_pencil_device": (
        C.c_int, [C.c_int, C.c_uint, _vp, C.c_int, _vp, C.c_int, _vp]),
    "starneig_amd_check_pencil_device": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _dp, _vp]),
    "starneig_amd_schur_device": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, C.POINTER(SchurConf), _vp, _dp]),
    "starneig_amd_schur_rows_device": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, C.POINTER(SchurConf), _vp, _dp]),
    "starneig_amd_schur_sharded_device": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp,
                  C.POINTER(SchurConf), _vp, _dp]),
    "starneig_amd_hessenberg_device": (
        C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _dp]),
    "starneig_amd_hessenberg_panel_ld": (C.c_int, [C.c_int, C.c_int]),
    "starneig_amd_hessenberg_sharded_device": (
        C.c_int, [C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, C.c_long,
                  C.c_int, C.c_int, _vp, _vp, _vp, _vp, _dp]),
    "starneig_amd_rccl_unique_id": (C.c_int, [_vp]),
    "starneig_amd_rccl_init": (C.c_int, [C.c_int, C.c_int, _vp]),
    "starneig_amd_rccl_finalize": (None, []),
    "starneig_amd_rccl_allreduce_sum": (C.c_int, [_vp, C.c_long, _vp]),
    "starneig_amd_rccl_broadcast": (C.c_int, [_vp, C.c_long, C.c_int, _vp]),
    "starneig_amd_dgemm_device": (
        C.c_int, [C.c_char, C.c_char, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int,
                  _vp, C.c_int, C.c_double, _vp, C.c_int, _vp]),
    "starneig_amd_lcg_fill_device": (
        C.c_int, [C.c_int, C.c_int, C.c_uint, C.c_int, _vp, C.c_int, _vp]),
    "starneig_amd_set_matrix_device": (
        C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, _vp, C.c_int, _vp]),
    "starneig_amd_check_device": (
        C.c_int, [C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _dp, _vp]),
    "starneig_amd_default_panel_width": (C.c_int, [C.c_int]),
    "starneig_amd_release_workspace": (None, []),
}

_lib = None
_test_lib = None
TEST_LIB_PATH = os.path.join(_HERE, "libstarneig_amd_test.so")


def load_test_hooks():
    """The test-support build (product objects + the sn_internal_* hooks that tests/ and scratch/
    call; csrc/Makefile).  The product library exports none of them."""
    global _test_lib
    if _test_lib is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise RuntimeError(f"{TEST_LIB_PATH} is missing: build it with `python -m starneig_amd.build`")
        _test_lib = C.CDLL(TEST_LIB_PATH, mode=C.RTLD_LOCAL)
    return _test_lib


def load():
    """Loads the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m starneig_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)        # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


# ---- life cycle (reference node.h:178-220) --------------------------------------

_atexit_registered = [False]


def node_init(cores=USE_ALL, gpus=1, flags=DEFAULT):
    """starneig_node_init.  `gpus` defaults to ONE device here (the C interface takes what the caller
    passes, STARNEIG_USE_ALL included): in the one-process-per-GPU mode of distributed.py every rank owns
    one device, and a default of "all" would make every rank start a team on every device of the node.
    Several devices from one process are an explicit request: node_init(cores, gpus=N)."""
    load().starneig_node_init(cores, gpus, flags)
    if not _atexit_registered[0]:
        import atexit
        atexit.register(_finalize_at_exit)
        _atexit_registered[0] = True


def _finalize_at_exit():
    try:
        if _lib is not None and _lib.starneig_node_initialized():
            _lib.starneig_node_finalize()
    except Exception:
        pass


def node_initialized():
    return bool(load().starneig_node_initialized())


def node_finalize():
    load().starneig_node_finalize()


def hessenberg_init_conf():
    conf = HessenbergConf()
    load().starneig_hessenberg_init_conf(C.byref(conf))
    return conf


# ---- host-array interface ---------------------------------------------------------

def _arr_ptr(a):
    return None if a is None else a.ctypes.data


def _host_ptr(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.f_contiguous
    return a.ctypes.data


def SEP_SM_Hessenberg(n, A, ldA, Q, ldQ):
    """reference sep_sm.h:89-92; A, Q are Fortran-ordered (ld, n) float64 arrays or None."""
    return load().starneig_SEP_SM_Hessenberg(n, _host_ptr(A), ldA, _host_ptr(Q), ldQ)


def SEP_SM_Hessenberg_expert(conf, n, begin, end, A, ldA, Q, ldQ):
    """reference sep_sm.h:380-384."""
    cp = C.byref(conf) if conf is not None else None
    return load().starneig_SEP_SM_Hessenberg_expert(
        cp, n, begin, end, _host_ptr(A), ldA, _host_ptr(Q), ldQ)


def schur_init_conf():
    conf = SchurConf()
    load().starneig_schur_init_conf(C.byref(conf))
    return conf


def SEP_SM_Schur(n, H, ldH, Q, ldQ, real, imag):
    """reference sep_sm.h:126-130; real/imag: float64 arrays of length n or None."""
    return load().starneig_SEP_SM_Schur(
        n, _host_ptr(H), ldH, _host_ptr(Q), ldQ,
        None if real is None else real.ctypes.data, None if imag is None else imag.ctypes.data)


def SEP_SM_Schur_expert(conf, n, H, ldH, Q, ldQ, real, imag):
    """reference sep_sm.h:424-429."""
    cp = C.byref(conf) if conf is not None else None
    return load().starneig_SEP_SM_Schur_expert(
        cp, n, _host_ptr(H), ldH, _host_ptr(Q), ldQ,
        None if real is None else real.ctypes.data, None if imag is None else imag.ctypes.data)


def GEP_SM_Schur(n, H, ldH, R, ldR, Q, ldQ, Z, ldZ, real, imag, beta):
    """reference gep_sm.h:164-170; eigenvalues are (real + i imag) / beta."""
    return load().starneig_GEP_SM_Schur(
        n, _host_ptr(H), ldH, _host_ptr(R), ldR, _host_ptr(Q), ldQ, _host_ptr(Z), ldZ,
        _arr_ptr(real), _arr_ptr(imag), _arr_ptr(beta))


def GEP_SM_HessenbergTriangular(n, A, ldA, B, ldB, Q, ldQ, Z, ldZ):
    """reference gep_sm.h:106-111 (wrappers/lapack.c:45-176)."""
    return load().starneig_GEP_SM_HessenbergTriangular(
        n, _host_ptr(A), ldA, _host_ptr(B), ldB, _host_ptr(Q), ldQ, _host_ptr(Z), ldZ)


def GEP_SM_Reduce(n, A, ldA, B, ldB, Q, ldQ, Z, ldZ, real, imag, beta):
    """reference gep_sm.h:316-326 without a predicate: HessenbergTriangular + Schur."""
    return load().starneig_GEP_SM_Reduce(
        n, _host_ptr(A), ldA, _host_ptr(B), ldB, _host_ptr(Q), ldQ, _host_ptr(Z), ldZ,
        _arr_ptr(real), _arr_ptr(imag), _arr_ptr(beta), None, None, None, None)


def GEP_SM_Schur_expert(conf, n, H, ldH, R, ldR, Q, ldQ, Z, ldZ, real, imag, beta):
    """reference gep_sm.h:503-510."""
    cp = C.byref(conf) if conf is not None else None
    return load().starneig_GEP_SM_Schur_expert(
        cp, n, _host_ptr(H), ldH, _host_ptr(R), ldR, _host_ptr(Q), ldQ, _host_ptr(Z), ldZ,
        _arr_ptr(real), _arr_ptr(imag), _arr_ptr(beta))


PREDICATE_FN = C.CFUNCTYPE(C.c_int, C.c_double, C.c_double, C.c_void_p)


def SEP_SM_Reduce(n, A, ldA, Q, ldQ, real, imag, predicate=None):
    """reference sep_sm.h:230-240 (common/combined.c:46-98): Hessenberg + Schur (+ Select and
    ReorderSchur when a predicate(real, imag) -> bool is given).  Returns rc without a predicate,
    (rc, selected, count) with one."""
    if predicate is None:
        return load().starneig_SEP_SM_Reduce(
            n, _host_ptr(A), ldA, _host_ptr(Q), ldQ, real.ctypes.data, imag.ctypes.data,
            None, None, None, None)
    cb = PREDICATE_FN(lambda re, im, arg: 1 if predicate(re, im) else 0)
    sel = np.zeros(n, dtype=np.int32)
    cnt = C.c_int(0)
    rc = load().starneig_SEP_SM_Reduce(
        n, _host_ptr(A), ldA, _host_ptr(Q), ldQ, real.ctypes.data, imag.ctypes.data,
        C.cast(cb, C.c_void_p), None, sel.ctypes.data, C.addressof(cnt))
    return rc, sel, cnt.value


def reorder_init_conf():
    conf = ReorderConf()
    load().starneig_reorder_init_conf(C.byref(conf))
    return conf


def SEP_SM_ReorderSchur(n, selected, S, ldS, Q, ldQ, real, imag, conf=None):
    """reference sep_sm.h:174-179 / :474-480; selected: int32 array of length n (in/out)."""
    assert selected.dtype == np.int32
    if conf is None:
        return load().starneig_SEP_SM_ReorderSchur(
            n, selected.ctypes.data, _host_ptr(S), ldS, _host_ptr(Q), ldQ, _arr_ptr(real), _arr_ptr(imag))
    return load().starneig_SEP_SM_ReorderSchur_expert(
        C.byref(conf), n, selected.ctypes.data, _host_ptr(S), ldS, _host_ptr(Q), ldQ,
        _arr_ptr(real), _arr_ptr(imag))


def SEP_SM_Select(n, S, ldS, predicate):
    """reference sep_sm.h:327-334; predicate(real, imag) -> bool.  Returns (rc, selected, count)."""
    cb = PREDICATE_FN(lambda re, im, arg: 1 if predicate(re, im) else 0) if predicate else None
    sel = np.zeros(n, dtype=np.int32)
    cnt = C.c_int(0)
    rc = load().starneig_SEP_SM_Select(
        n, _host_ptr(S), ldS, C.cast(cb, C.c_void_p) if cb else None, None,
        sel.ctypes.data, C.addressof(cnt))
    return rc, sel, cnt.value


RAW_HEADER = "STARNEIG RAW REAL DOUBLE M %d N %d\n"


def write_raw(path, A):
    """The reference test driver's raw matrix format (test/common/io.c:217-268): one header
    line, then the columns as native doubles -- fixtures written here can be fed to
    `starneig-test --init read-raw` and vice versa."""
    A = np.asarray(A, dtype=np.float64)
    with open(path, "wb") as f:
        f.write((RAW_HEADER % A.shape).encode())
        f.write(np.asfortranarray(A).tobytes(order="F"))


def read_raw(path):
    import re
    with open(path, "rb") as f:
        header = f.readline().decode()
        m = re.match(r"STARNEIG RAW REAL DOUBLE M (\d+) N (\d+)", header)
        if not m:
            raise ValueError("not a STARNEIG RAW file")
        rows, cols = int(m.group(1)), int(m.group(2))
        data = np.frombuffer(f.read(rows * cols * 8), dtype=np.float64)
    return np.asfortranarray(data.reshape((rows, cols), order="F"))


# ---- device-pointer extension (torch tensors only carry the memory) ---------------

def _dev_ptr(t):
    return None if t is None else t.data_ptr()


def _stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def device_matrix(n, ld=None, m=None):
    """Column-major (ld x n) fp64 matrix in HBM: a torch tensor of shape (n, ld)."""
    import torch
    m = n if m is None else m
    ld = ld or (m + 15) // 16 * 16
    return torch.zeros((n, ld), dtype=torch.float64, device="cuda")


def hessenberg_device(tA, tQ, n=None, begin=0, end=None, panel_width=-1, stats=False,
                      sample_every=0):
    n = tA.shape[0] if n is None else n
    end = n if end is None else end
    st = (C.c_double * 16)() if stats else None
    if stats:
        st[7] = float(sample_every)
    rc = load().starneig_amd_hessenberg_device(
        n, begin, end, panel_width, _dev_ptr(tA), tA.shape[1],
        _dev_ptr(tQ), tQ.shape[1] if tQ is not None else 0, _stream_ptr(), st)
    if stats:
        return rc, {"total_ms": st[0], "gemv_bytes": st[1], "gemm_flops": st[2],
                    "gemv_sampled_ms": st[3], "gemv_sampled_bytes": st[4],
                    "gemv_launches": int(st[5]), "gemv_sampled_launches": int(st[6]),
                    "gemm_main_ms": st[8], "gemm_main_flops": st[9], "gemm_side_ms": st[10],
                    "gemm_fused_ms": st[11], "gemm_fused_flops": st[12]}
    return rc


def schur_device(tH, tQ, n=None, conf=None, eigenvalues=True):
    """Returns (rc, real, imag, stats)."""
    n = tH.shape[0] if n is None else n
    real = np.zeros(n) if eigenvalues else None
    imag = np.zeros(n) if eigenvalues else None
    st = (C.c_double * 8)()
    rc = load().starneig_amd_schur_device(
        n, _dev_ptr(tH), tH.shape[1], _dev_ptr(tQ), tQ.shape[1] if tQ is not None else 0,
        None if real is None else real.ctypes.data, None if imag is None else imag.ctypes.data,
        C.byref(conf) if conf is not None else None, _stream_ptr(), st)
    stats = {"total_ms": st[0], "sweeps": int(st[1]), "aeds": int(st[2]),
             "small_solves": int(st[3]), "chase_launches": int(st[4]), "gemm_flops": st[5],
             "aed_host_s": st[6], "gpu_wait_s": st[7]}
    return rc, real, imag, stats


def schur_sharded_device(tH, tQrows, q_rows, rank, world, n=None, conf=None):
    """One rank's share of the sharded Schur leg (starneig_amd_schur_sharded_device): a replica of H,
    q_rows rows of Q starting at tQrows (a device pointer or a tensor), the deflated column tiles T of
    H with T % world == rank.  Returns (rc, real, imag, stats)."""
    n = tH.shape[0] if n is None else n
    real, imag = np.zeros(n), np.zeros(n)
    st = (C.c_double * 8)()
    qptr, ldq = (tQrows.data_ptr(), tQrows.shape[1]) if hasattr(tQrows, "data_ptr") else tQrows
    rc = load().starneig_amd_schur_sharded_device(
        n, _dev_ptr(tH), tH.shape[1], qptr, ldq, q_rows, rank, world, real.ctypes.data, imag.ctypes.data,
        C.byref(conf) if conf is not None else None, _stream_ptr(), st)
    stats = {"total_ms": st[0], "sweeps": int(st[1]), "aeds": int(st[2]),
             "small_solves": int(st[3]), "chase_launches": int(st[4]), "gemm_flops": st[5],
             "aed_host_s": st[6], "gpu_wait_s": st[7]}
    return rc, real, imag, stats


def reorder_schur_device(tS, tQ, selected, n=None, conf=None, eigenvalues=True):
    """Returns (rc, real, imag, stats); `selected` (int32, length n) is updated in place."""
    n = tS.shape[0] if n is None else n
    assert selected.dtype == np.int32
    real = np.zeros(n) if eigenvalues else None
    imag = np.zeros(n) if eigenvalues else None
    st = (C.c_double * 4)()
    rc = load().starneig_amd_reorder_schur_device(
        n, selected.ctypes.data, _dev_ptr(tS), tS.shape[1], _dev_ptr(tQ),
        tQ.shape[1] if tQ is not None else 0, _arr_ptr(real), _arr_ptr(imag),
        C.byref(conf) if conf is not None else None, _stream_ptr(), st)
    return rc, real, imag, {"windows": int(st[0]), "gemm_flops": st[1], "rounds": int(st[2])}


def hessenberg_triangular_device(tA, tB, tQ, tZ, n=None):
    """Device-resident Hessenberg-triangular reduction. Returns (rc, stats)."""
    n = tA.shape[0] if n is None else n
    st = (C.c_double * 8)()
    rc = load().starneig_amd_hessenberg_triangular_device(
        n, _dev_ptr(tA), tA.shape[1], _dev_ptr(tB), tB.shape[1],
        _dev_ptr(tQ), tQ.shape[1] if tQ is not None else 0,
        _dev_ptr(tZ), tZ.shape[1] if tZ is not None else 0, _stream_ptr(), st)
    return rc, {"total_ms": st[0], "qr_ms": st[1], "rotation_ms": st[2], "gemm_flops": st[3],
                "rotations": st[4], "two_stage": bool(st[5]), "stage1_ms": st[6]}


def gep_schur_device(tH, tR, tQ, tZ, n=None, conf=None, eigenvalues=True):
    """Returns (rc, real, imag, beta, stats)."""
    n = tH.shape[0] if n is None else n
    real = np.zeros(n) if eigenvalues else None
    imag = np.zeros(n) if eigenvalues else None
    beta = np.zeros(n) if eigenvalues else None
    st = (C.c_double * 8)()
    rc = load().starneig_amd_gep_schur_device(
        n, _dev_ptr(tH), tH.shape[1], _dev_ptr(tR), tR.shape[1],
        _dev_ptr(tQ), tQ.shape[1] if tQ is not None else 0,
        _dev_ptr(tZ), tZ.shape[1] if tZ is not None else 0,
        _arr_ptr(real), _arr_ptr(imag), _arr_ptr(beta),
        C.byref(conf) if conf is not None else None, _stream_ptr(), st)
    stats = {"total_ms": st[0], "sweeps": int(st[1]), "aeds": int(st[2]),
             "small_solves": int(st[3]), "chase_launches": int(st[4]), "gemm_flops": st[5],
             "aed_host_s": st[6], "gpu_wait_s": st[7]}
    return rc, real, imag, beta, stats


def lcg_pencil_device(tH, tR, n, seed=2019):
    return load().starneig_amd_lcg_pencil_device(
        n, seed, _dev_ptr(tH), tH.shape[1], _dev_ptr(tR), tR.shape[1], _stream_ptr())


def check_pencil_device(tQ, tS, tZ, tA0, n=None):
    """2^52 ||Q S Z^T - A0|| / ||A0||, orthogonality of Q and Z (in u), non-zeros of S below
    the first sub-diagonal -- on the GPU."""
    import torch
    n = tA0.shape[0] if n is None else n
    w1 = torch.empty((n, n), dtype=torch.float64, device="cuda")
    w2 = torch.empty((n, n), dtype=torch.float64, device="cuda")
    out = (C.c_double * 4)()
    rc = load().starneig_amd_check_pencil_device(
        n, _dev_ptr(tQ), tQ.shape[1], _dev_ptr(tS), tS.shape[1], _dev_ptr(tZ), tZ.shape[1],
        _dev_ptr(tA0), tA0.shape[1], _dev_ptr(w1), _dev_ptr(w2), out, _stream_ptr())
    return rc, {"residual_u": out[0], "orthogonality_q_u": out[1], "orthogonality_z_u": out[2],
                "below_subdiagonal": int(out[3])}


def dgemm_device(transA, transB, m, n, k, alpha, tA, ldA, tB, ldB, beta, tC, ldC):
    return load().starneig_amd_dgemm_device(
        transA.encode(), transB.encode(), m, n, k, alpha, _dev_ptr(tA), ldA,
        _dev_ptr(tB), ldB, beta, _dev_ptr(tC), ldC, _stream_ptr())


def lcg_fill_device(t, m, n, seed=2019, mode=0):
    return load().starneig_amd_lcg_fill_device(m, n, seed, mode, _dev_ptr(t), t.shape[1], _stream_ptr())


def set_matrix_device(t, m, n, value=0.0, diag=0.0):
    return load().starneig_amd_set_matrix_device(m, n, value, diag, _dev_ptr(t), t.shape[1], _stream_ptr())


def check_device(tQ, tH, tA0, n=None):
    """Residual / orthogonality in units of u and the count of non-zeros below the
    sub-diagonal, computed on the GPU (reference test/common/checks.c:180-208)."""
    import torch
    n = tA0.shape[0] if n is None else n
    w1 = torch.empty((n, n), dtype=torch.float64, device="cuda")
    w2 = torch.empty((n, n), dtype=torch.float64, device="cuda")
    out = (C.c_double * 3)()
    rc = load().starneig_amd_check_device(
        n, _dev_ptr(tQ), tQ.shape[1], _dev_ptr(tH), tH.shape[1], _dev_ptr(tA0), tA0.shape[1],
        _dev_ptr(w1), _dev_ptr(w2), out, _stream_ptr())
    return rc, {"residual_u": out[0], "orthogonality_u": out[1], "below_subdiagonal": int(out[2])}


def default_panel_width(n):
    return load().starneig_amd_default_panel_width(n)
