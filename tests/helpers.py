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
