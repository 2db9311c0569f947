"""Stage 2 of the two-stage Hessenberg-triangular reduction in WAVEFRONT order, as the device code runs it: the steps
(sweep j, position t) with t + LAG j = tau together -- first every step's two reflectors from the state at the start
of the wavefront (kernel ht2_gen), then all left applications, then all right applications.  Which LAG is enough?
python scratch/ht2_lag.py"""
import importlib.util, os, sys
import numpy as np
spec = importlib.util.spec_from_file_location("p", os.path.join(os.path.dirname(__file__), "ht2_proto.py"))
P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)


def stage2_wavefronts(A, B, Q, Z, r, lag):
    n = A.shape[0]
    waves = 0
    tau = 0
    while True:
        steps = []
        for j in range(min(tau // lag, n - 3), -1, -1):
            t = tau - lag * j
            p = j + 1 + r * t
            if p > n - 2:
                break
            steps.append((j, t, p, min(p + r, n), j if t == 0 else p - r))
        if not steps:
            if tau // lag >= n - 3:
                break
            tau += 1
            continue
        waves += 1
        refl = []
        for j, t, p, p1, c0 in steps:                       # gen: from the state at the start of the wavefront
            I = slice(p, p1)
            v, th, beta = P.house(A[I, c0])
            A[p, c0] = beta; A[p + 1:p1, c0] = 0.0
            M = B[I, I] - th * np.outer(v, v @ B[I, I])
            x = P.opposite(M)
            w, tz, _ = P.house(x)
            refl.append((v, th, w, tz))
        for (j, t, p, p1, c0), (v, th, w, tz) in zip(steps, refl):      # left
            I = slice(p, p1)
            A[I, c0 + 1:] -= th * np.outer(v, v @ A[I, c0 + 1:])
            B[I, p:] -= th * np.outer(v, v @ B[I, p:])
            Q[:, I] -= th * np.outer(Q[:, I] @ v, v)
        for (j, t, p, p1, c0), (v, th, w, tz) in zip(steps, refl):      # right
            I = slice(p, p1)
            rb = p1; ra = min(p1 + r, n)
            B[:rb, I] -= tz * np.outer(B[:rb, I] @ w, w)
            B[p + 1:p1, p] = 0.0
            A[:ra, I] -= tz * np.outer(A[:ra, I] @ w, w)
            Z[:, I] -= tz * np.outer(Z[:, I] @ w, w)
        tau += 1
    return waves


if __name__ == "__main__":
    u = 2.0 ** -52
    for n, r in [(60, 4), (97, 8), (150, 8), (200, 16)]:
        for lag in (3, 2, 1):
            rng = np.random.default_rng(n)
            A0 = rng.standard_normal((n, n)); B0 = np.triu(rng.standard_normal((n, n)))
            A, B = A0.copy(), B0.copy(); Q = np.eye(n); Z = np.eye(n)
            P.stage1(A, B, Q, Z, r)
            waves = stage2_wavefronts(A, B, Q, Z, r, lag)
            print(f"n={n} r={r} lag={lag}: {waves} wavefronts, below subdiagonal {np.abs(np.tril(A, -2)).max():.1e}, B lower {np.abs(np.tril(B, -1)).max():.1e}, "
                  f"residuals {np.linalg.norm(Q @ A @ Z.T - A0) / np.linalg.norm(A0) / u:.1f} / {np.linalg.norm(Q @ B @ Z.T - B0) / np.linalg.norm(B0) / u:.1f} u, "
                  f"orthogonality {np.linalg.norm(Z.T @ Z - np.eye(n)) / u:.1f} u")
