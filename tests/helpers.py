"""Shared test helpers: tolerances and conversions (tests only)."""
import numpy as np

U = 2.0 ** -52

# reference thresholds, test/common/hooks.c:52,57 (units of u)
WARN_U = 500.0
FAIL_U = 10000.0


def to_device(a):
    """(ld, n) Fortran numpy array -> torch (n, ld) tensor on the GPU."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(a.T)).cuda()


def to_host(t):
    return np.asfortranarray(t.cpu().numpy().T)


def elementwise_tolerance(n):
    """max|H_gpu - H_oracle| / ||A||_F <= c*sqrt(n)*u, c = 8.
    (LAPACK vs the oracle itself measures 4.5 u at n=64 ... 15 u at n=300.)"""
    return 8.0 * np.sqrt(n) * U


def eig_backward_error_u(A0, B0, alpha, beta, sample=48):
    """Backward error of computed generalized eigenvalues, independent of their conditioning:
    max over (a sample of) the pairs of sigma_min(beta A0 - alpha B0) / (|beta| ||A0||_F +
    |alpha| ||B0||_F), in units of u.  A pair is an exact eigenvalue of a pencil that close."""
    import scipy.linalg as sl
    n = A0.shape[1]
    A, B = A0[:n], B0[:n]
    na, nb = np.linalg.norm(A), np.linalg.norm(B)
    idx = np.unique(np.concatenate([np.linspace(0, n - 1, min(n, sample)).astype(int),
                                    [int(np.argmax(np.abs(alpha))), int(np.argmin(np.abs(alpha)))]]))
    worst = 0.0
    for i in idx:
        M = beta[i] * A - alpha[i] * B
        smin = sl.svdvals(M)[-1]
        worst = max(worst, smin / (abs(beta[i]) * na + abs(alpha[i]) * nb))
    return worst / U


def _chunked_mm(a, b, chunk=128):
    """a @ b in fp64 through torch (rocBLAS), the k dimension in chunks whose products are added
    one after the other: two-level summation, so that the check's own rounding stays ~1 u
    (one k = n chain on a diagonal of ones adds 5-20 u by itself)."""
    import torch
    out = torch.zeros((a.shape[0], b.shape[1]), dtype=torch.float64, device=a.device)
    for k0 in range(0, a.shape[1], chunk):
        out.addmm_(a[:, k0:k0 + chunk], b[k0:k0 + chunk, :])
    return out


def torch_check(tQ, tH, tA0, n):
    """The reference's residual and orthogonality measures (test/common/checks.c:180-208:
    2^52 ||Q H Q^T - A||_F / ||A||_F and 2^52 ||Q Q^T - I||_F / sqrt(n)) computed with
    torch.matmul in fp64 -- independent of the library's own kernels.  Tensors are the
    (n, ld) transposed images of the column-major matrices."""
    import torch
    q, h, a0 = tQ[:n, :n], tH[:n, :n], tA0[:n, :n]          # = Q^T, H^T, A0^T
    r = _chunked_mm(_chunked_mm(q.T, h), q) - a0            # (Q H Q^T - A0)^T
    res = float(torch.linalg.norm(r) / torch.linalg.norm(a0)) / U
    del r
    o = _chunked_mm(q.T, q)                                 # Q Q^T
    o.diagonal().sub_(1.0)
    orth = float(torch.linalg.norm(o)) / np.sqrt(n) / U
    return res, orth


def torch_check_pencil(tQ, tS, tZ, tA0, n):
    """2^52 ||Q S Z^T - A||_F / ||A||_F and the orthogonality of Q and Z, by torch in fp64."""
    import torch
    q, s, z, a0 = tQ[:n, :n], tS[:n, :n], tZ[:n, :n], tA0[:n, :n]
    r = _chunked_mm(_chunked_mm(z.T, s), q) - a0            # (Q S Z^T - A0)^T = Z S^T Q^T - A0^T
    res = float(torch.linalg.norm(r) / torch.linalg.norm(a0)) / U
    del r
    out = [res]
    for m in (q, z):
        o = _chunked_mm(m.T, m)
        o.diagonal().sub_(1.0)
        out.append(float(torch.linalg.norm(o)) / np.sqrt(n) / U)
    return tuple(out)
