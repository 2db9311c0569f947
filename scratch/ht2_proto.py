"""numpy prototype of the two-stage Hessenberg-triangular reduction (Dackland-Kagstrom; Kagstrom, Kressner,
Quintana-Orti, Quintana-Orti 2008): stage 1 to r-Hessenberg-triangular form by QR of (2r x r) blocks of A and
RQ of the bottom rows of the filled diagonal blocks of B; stage 2 a Householder bulge chase (left reflector of
length r from A's overhanging column, 'opposite' reflector from the right that restores the first column of
the B block)."""
import numpy as np


def house(x):
    """v, tau, beta with (I - tau v v^T) x = beta e1, v[0] = 1 (dlarfg)"""
    x = np.asarray(x, dtype=float)
    alpha = x[0]
    xn = np.linalg.norm(x[1:])
    if xn == 0.0:
        return np.r_[1.0, np.zeros(len(x) - 1)], 0.0, alpha
    beta = -np.copysign(np.hypot(alpha, xn), alpha)
    tau = (beta - alpha) / beta
    v = np.r_[1.0, x[1:] / (alpha - beta)]
    return v, tau, beta


def qr_house(M):
    """Householder QR; returns list of (v, tau) acting on rows k.."""
    M = M.copy()
    m, n = M.shape
    refl = []
    for k in range(min(m - 1, n)):
        v, tau, beta = house(M[k:, k])
        M[k:, k:] -= tau * np.outer(v, v @ M[k:, k:])
        refl.append((k, v, tau))
    return refl, M


def apply_left(refl, X, transpose=True):
    """X <- Q^T X (Q = H_0 H_1 ...)"""
    for k, v, tau in refl:
        X[k:, :] -= tau * np.outer(v, v @ X[k:, :])


def apply_right(refl, X):
    """X <- X Q"""
    for k, v, tau in refl:
        X[:, k:] -= tau * np.outer(X[:, k:] @ v, v)


def stage1(A, B, Q, Z, r):
    n = A.shape[0]
    for jc in range(0, n - r - 1, r):
        nb = min(r, n - jc)
        top = jc + r
        # row blocks of r from `top`
        starts = list(range(top, n, r))
        for k in range(len(starts) - 1, 0, -1):
            i0 = starts[k - 1]; i1 = min(starts[k] + r, n)
            I = slice(i0, i1)
            refl, _ = qr_house(A[I, jc:jc + nb])
            apply_left(refl, A[I, jc:])
            A[i0 + nb:i1, jc:jc + nb] = np.tril(A[i0 + nb:i1, jc:jc + nb], -10**9) * 0  # exact zeros below R
            for c in range(nb):
                A[i0 + c + 1:i1, jc + c] = 0.0
            apply_left(refl, B[I, i0:])
            apply_right(refl, Q[:, I])
            # restore B: bottom rows [i0 + r, i1) of the block B(I, I): RQ
            m = i1 - i0
            mb = i1 - (i0 + r)            # bottom rows
            if mb > 0:
                Mb = B[i0 + r:i1, I]      # mb x m ; want Mb G = [0 R]
                # QR of flipped transpose: Mb^T (m x mb), reflect so that columns end in the last rows
                Pm = np.eye(m)[::-1]; Pb = np.eye(mb)[::-1]
                reflr, _ = qr_house(Pm @ Mb.T @ Pb)
                # G = Pm Qr Pm
                def right_G(X):
                    Xp = X[:, ::-1].copy()
                    apply_right(reflr, Xp)
                    X[:, :] = Xp[:, ::-1]
                right_G(B[:i1, I]); right_G(A[:, I]); right_G(Z[:, I])
                for rr in range(mb):
                    B[i0 + r + rr, i0:i0 + r + rr] = 0.0
        if len(starts) == 1 and n - top > 1:
            I = slice(top, n)
            refl, _ = qr_house(A[I, jc:jc + nb])
            apply_left(refl, A[I, jc:])
            for c in range(nb):
                A[top + c + 1:n, jc + c] = 0.0
            apply_left(refl, B[I, top:])
            apply_right(refl, Q[:, I])
        # top block rows [top, top + r): full r x r diagonal block of B -> RQ
        i0 = top; i1 = min(top + r, n); m = i1 - i0
        if m > 1:
            Mb = B[i0:i1, i0:i1]
            P = np.eye(m)[::-1]
            reflr, _ = qr_house(P @ Mb.T @ P)
            def right_G(X):
                Xp = X[:, ::-1].copy()
                apply_right(reflr, Xp)
                X[:, :] = Xp[:, ::-1]
            I = slice(i0, i1)
            right_G(B[:i1, I]); right_G(A[:, I]); right_G(Z[:, I])
            for rr in range(m):
                B[i0 + rr, i0:i0 + rr] = 0.0


def opposite(M):
    """unit x with (M x)[1:] = 0 : orthogonal to rows 1.. of M"""
    m = M.shape[0]
    if m == 1:
        return np.ones(1)
    refl, _ = qr_house(M[1:, :].T.copy())       # m x (m-1)
    e = np.zeros((m, 1)); e[m - 1, 0] = 1.0
    # last column of Q = H_0 ... H_{m-2} e_m
    for k, v, tau in reversed(refl):
        e[k:, :] -= tau * np.outer(v, v @ e[k:, :])
    return e[:, 0]


def stage2(A, B, Q, Z, r, log=None):
    n = A.shape[0]
    for j in range(n - 2):
        p = j + 1; c0 = j; t = 0
        while True:
            p1 = min(p + r, n)
            if p1 - p < 2:
                break
            I = slice(p, p1)
            v, tau, beta = house(A[I, c0])
            A[I, c0:] -= tau * np.outer(v, v @ A[I, c0:])
            A[p + 1:p1, c0] = 0.0
            B[I, :] -= tau * np.outer(v, v @ B[I, :])
            Q[:, I] -= tau * np.outer(Q[:, I] @ v, v)
            x = opposite(B[I, I])
            w, tz, _ = house(x)                  # G = I - tz w w^T, G e1 = +-x
            B[:, I] -= tz * np.outer(B[:, I] @ w, w)
            B[p + 1:p1, p] = 0.0
            A[:, I] -= tz * np.outer(A[:, I] @ w, w)
            Z[:, I] -= tz * np.outer(Z[:, I] @ w, w)
            if log is not None:
                log.append((j, t, p, p1, c0))
            c0 = p; p = p + r; t += 1
            if p >= n - 1:
                break


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for n, r in [(40, 4), (97, 8), (150, 16)]:
        A0 = rng.standard_normal((n, n)); B0 = np.triu(rng.standard_normal((n, n)))
        if n == 97:
            B0[10, 10] = 0.0; B0[50, 50] = 0.0          # singular B
        A, B = A0.copy(), B0.copy(); Q = np.eye(n); Z = np.eye(n)
        stage1(A, B, Q, Z, r)
        band = max(abs(A[i, c]) for i in range(n) for c in range(n) if i > c + r) if n > r + 1 else 0
        print(n, r, "stage 1: below band", band, "B lower", np.abs(np.tril(B, -1)).max(),
              "res", np.linalg.norm(Q @ A @ Z.T - A0) / np.linalg.norm(A0), np.linalg.norm(Q @ B @ Z.T - B0) / np.linalg.norm(B0))
        stage2(A, B, Q, Z, r)
        print("      stage 2: below subdiag", np.abs(np.tril(A, -2)).max(), "B lower", np.abs(np.tril(B, -1)).max(),
              "res", np.linalg.norm(Q @ A @ Z.T - A0) / np.linalg.norm(A0), np.linalg.norm(Q @ B @ Z.T - B0) / np.linalg.norm(B0),
              "orth", np.linalg.norm(Q.T @ Q - np.eye(n)), np.linalg.norm(Z.T @ Z - np.eye(n)))
