"""Stage 2 of the two-stage Hessenberg-triangular reduction with DEFERRED updates, in the device's wavefront order
(scratch/ht2_lag.py): which parts of the reflector applications to A and B may wait until the group of GS sweeps a
reflector belongs to is through, and which may not.

  right_top  (valid, what csrc/ht_twostage.hip does from round 6 on): the opposite reflector G(j, t) is applied at
             once only to the rows [top, ...) with top = (j // GS) * GS + 1; the rows [0, top) of A and B take the
             reflectors of a whole group later, position by position in DEcreasing t like Q and Z do.  Why it is
             valid: every left reflector from wavefront 2 * (j // GS) * GS on acts on rows >= top (sweep j' at
             position t' = tau - 2 j' has p' = j' + 1 + t' r >= top whenever tau >= 2 (top - 1)), so those rows only
             ever see right reflectors again; the right reflectors of OLDER groups that overlap a deferred one in
             columns were generated (and applied there at once) before it, the later ones are disjoint from it.
  left_far   (NOT valid, the negative control): the left reflector H(j, t) is applied at once only to the columns
             [c0 + 1, p + extent) and to the rest when its group is through.  A right reflector of a later step whose
             column range straddles that boundary then mixes updated and stale columns on the rows of H.  This is the
             "order of availability" problem of DESIGN.md section 4d: left and right reflectors commute as operators,
             but only when each is applied to ALL of the entries the other one mixes.

python scratch/ht2_defer.py"""
import importlib.util
import os

import numpy as np

spec = importlib.util.spec_from_file_location("p", os.path.join(os.path.dirname(__file__), "ht2_proto.py"))
P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)


def stage2_deferred(A, B, Q, Z, r, gs, mode="right_top", lag=2, extent=None, count=None):
    """mode: "right_top" | "left_far" | "none".  count (dict, optional): bytes moved by the immediate applications
    ("now") and by the deferred ones if each group's reflectors of one position share one pass ("later")."""
    n = A.shape[0]
    pending = {}                 # group -> list of (j, t, p, p1, c0, v, th, w, tz, boundary)
    extent = 3 * r if extent is None else extent

    def last_wave(g):
        jl = min(g * gs + gs - 1, n - 3)
        return lag * jl + (n - 3 - jl) // r

    def flush(g):
        items = pending.pop(g, [])
        # decreasing position, increasing sweep: (j, t + 1) precedes (j', t) for j' > j, the only overlapping pairs
        items.sort(key=lambda it: (-it[1], it[0]))
        seen_t = set()
        for (j, t, p, p1, c0, v, th, w, tz, bnd) in items:
            I = slice(p, p1)
            if mode == "right_top":
                top = bnd
                B[:top, I] -= tz * np.outer(B[:top, I] @ w, w)
                A[:top, I] -= tz * np.outer(A[:top, I] @ w, w)
                if count is not None and t not in seen_t:
                    seen_t.add(t); count["later"] += 2 * 16 * top * (gs - 1 + r)
            elif mode == "left_far":
                A[I, bnd:] -= th * np.outer(v, v @ A[I, bnd:])
                B[I, bnd:] -= th * np.outer(v, v @ B[I, bnd:])

    tau, closed, ngroups = 0, 0, (n - 2 + gs - 1) // gs
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
        refl = []
        for j, t, p, p1, c0 in steps:
            I = slice(p, p1)
            v, th, beta = P.house(A[I, c0])
            A[p, c0] = beta; A[p + 1:p1, c0] = 0.0
            M = B[I, I] - th * np.outer(v, v @ B[I, I])
            x = P.opposite(M)
            w, tz, _ = P.house(x)
            refl.append((v, th, w, tz))
        for (j, t, p, p1, c0), (v, th, w, tz) in zip(steps, refl):      # left
            I = slice(p, p1)
            bnd = min(n, p + extent) if mode == "left_far" else n
            A[I, c0 + 1:bnd] -= th * np.outer(v, v @ A[I, c0 + 1:bnd])
            B[I, p:bnd] -= th * np.outer(v, v @ B[I, p:bnd])
            Q[:, I] -= th * np.outer(Q[:, I] @ v, v)
            if mode == "left_far" and bnd < n:
                pending.setdefault(j // gs, []).append((j, t, p, p1, c0, v, th, w, tz, bnd))
            if count is not None:
                count["now"] += 16 * (p1 - p) * ((bnd - c0 - 1) + (bnd - p))
        for (j, t, p, p1, c0), (v, th, w, tz) in zip(steps, refl):      # right
            I = slice(p, p1)
            rb = p1; ra = min(p1 + r, n)
            top = (j // gs) * gs + 1 if mode == "right_top" else 0
            B[top:rb, I] -= tz * np.outer(B[top:rb, I] @ w, w)
            B[p + 1:p1, p] = 0.0
            A[top:ra, I] -= tz * np.outer(A[top:ra, I] @ w, w)
            Z[:, I] -= tz * np.outer(Z[:, I] @ w, w)
            if mode == "right_top" and top > 0:
                pending.setdefault(j // gs, []).append((j, t, p, p1, c0, v, th, w, tz, top))
            if count is not None:
                count["now"] += 16 * (p1 - p) * ((rb - top) + (ra - top))
        while closed < ngroups and last_wave(closed) <= tau:
            flush(closed); closed += 1
        tau += 1
    while closed < ngroups:
        flush(closed); closed += 1


if __name__ == "__main__":
    u = 2.0 ** -52
    for n, r, gs in [(97, 8, 8), (150, 8, 16), (200, 16, 16), (260, 8, 64)]:
        for mode in ("none", "right_top", "left_far"):
            rng = np.random.default_rng(n)
            A0 = rng.standard_normal((n, n)); B0 = np.triu(rng.standard_normal((n, n)))
            A, B = A0.copy(), B0.copy(); Q = np.eye(n); Z = np.eye(n)
            P.stage1(A, B, Q, Z, r)
            cnt = {"now": 0, "later": 0}
            stage2_deferred(A, B, Q, Z, r, gs, mode, count=cnt)
            print(f"n={n} r={r} gs={gs} {mode:9s}: below subdiagonal {np.abs(np.tril(A, -2)).max():.1e}, B lower {np.abs(np.tril(B, -1)).max():.1e}, "
                  f"residuals {np.linalg.norm(Q @ A @ Z.T - A0) / np.linalg.norm(A0) / u:.3g} / "
                  f"{np.linalg.norm(Q @ B @ Z.T - B0) / np.linalg.norm(B0) / u:.3g} u; bytes now {cnt['now'] / 1e6:.2f} MB, later {cnt['later'] / 1e6:.2f} MB")
