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
