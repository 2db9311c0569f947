"""Stage 2 of the two-stage Hessenberg-triangular reduction: in which order may the launches of a wavefront run (round 6)?
The 75 us factorisation behind every opposite reflector hid under the left application only; the question was whether
the generation of wavefront tau + 1 may run BEFORE the bulk of wavefront tau's right applications.

  device order until round 6 (scratch/ht2_lag.py):   gen(tau) | left(tau) | right(tau) | gen(tau + 1) | ...
  1. stage2_overlapped:    ... | near-right(tau) | gen(tau + 1) | far-right(tau) | left(tau + 1) | near-right(tau + 1) | ...
  2. stage2_two_streams, merged=True -- what csrc/ht_twostage.hip runs, on ONE stream:
        gen(tau) | far(tau - 1) | left(tau) without the steps' own blocks | near(tau) = H on the own blocks, near-right(tau)
     on the device gen(tau) is split: its first half shares a launch with far(tau - 1), its second half with left(tau).
  3. stage2_two_streams, merged=False -- a latency chain gen | near | gen | ... on one stream, left and far on a second
     one, as early or as late as three hand-over rules allow.  Valid (and measured slower on the device: every hand-over
     costs 10-14 us), kept here with its negative controls because it states exactly which orderings matter.

near-right(tau) of a step at p: the rows [p - r + 1, ...) of its right application (the r - 1 rows above its block, the
block, and for A the r rows below) -- everything gen(tau + 1) reads: the column A(p + r : p + 2r, p) its own sweep's next
left reflector comes from, and the column B(p - r + 1 : p + 1, p) that is the last column of the YOUNGER neighbour's
next block.  far-right(tau): the rows above, which only later applications touch.

python scratch/ht2_overlap.py"""
import importlib.util
import os

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("p", os.path.join(here, "ht2_proto.py"))
P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)
spec2 = importlib.util.spec_from_file_location("lag", os.path.join(here, "ht2_lag.py"))
LAGM = importlib.util.module_from_spec(spec2); spec2.loader.exec_module(LAGM)


def stage2_overlapped(A, B, Q, Z, r, lag=2, gs=None, near_above=None):
    """gs: group size of the deferred top rows (None: no deferral, top = 0).  near_above: rows above the block that
    belong to the near part (default r - 1; smaller values are the negative control)."""
    n = A.shape[0]
    near_above = r - 1 if near_above is None else near_above
    pending = []                 # far parts of the previous wavefront: (p, p1, top, cut, w, tz)
    deferred = {}                # group -> list of (t, j, p, p1, top, w, tz): rows [0, top)
    tau = 0

    def flush_far():
        for (p, p1, top, cut, w, tz) in pending:
            I = slice(p, p1)
            if cut > top:
                B[top:cut, I] -= tz * np.outer(B[top:cut, I] @ w, w)
                A[top:cut, I] -= tz * np.outer(A[top:cut, I] @ w, w)
        pending.clear()

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
        for j, t, p, p1, c0 in steps:                       # gen(tau): the far parts of tau - 1 are still pending
            I = slice(p, p1)
            v, th, beta = P.house(A[I, c0])
            A[p, c0] = beta; A[p + 1:p1, c0] = 0.0
            M = B[I, I] - th * np.outer(v, v @ B[I, I])
            x = P.opposite(M)
            w, tz, _ = P.house(x)
            refl.append((v, th, w, tz))
        flush_far()                                          # far-right(tau - 1)
        for (j, t, p, p1, c0), (v, th, w, tz) in zip(steps, refl):      # left(tau)
            I = slice(p, p1)
            A[I, c0 + 1:] -= th * np.outer(v, v @ A[I, c0 + 1:])
            B[I, p:] -= th * np.outer(v, v @ B[I, p:])
            Q[:, I] -= th * np.outer(Q[:, I] @ v, v)
        for (j, t, p, p1, c0), (v, th, w, tz) in zip(steps, refl):      # near-right(tau)
            I = slice(p, p1)
            rb = p1; ra = min(p1 + r, n)
            top = 0 if gs is None else (j // gs) * gs + 1
            cut = max(top, p - near_above)
            B[cut:rb, I] -= tz * np.outer(B[cut:rb, I] @ w, w)
            B[p + 1:p1, p] = 0.0
            A[cut:ra, I] -= tz * np.outer(A[cut:ra, I] @ w, w)
            Z[:, I] -= tz * np.outer(Z[:, I] @ w, w)
            pending.append((p, p1, top, cut, w, tz))
            if top > 0:
                deferred.setdefault(j // gs, []).append((t, j, p, p1, top, w, tz))
        tau += 1
    flush_far()
    for g in sorted(deferred):                               # the rows above the groups' tops: decreasing t, increasing j
        for (t, j, p, p1, top, w, tz) in sorted(deferred[g], key=lambda it: (-it[0], it[1])):
            I = slice(p, p1)
            B[:top, I] -= tz * np.outer(B[:top, I] @ w, w)
            A[:top, I] -= tz * np.outer(A[:top, I] @ w, w)


def run(n, r, gs=None, near_above=None, seed=None):
    rng = np.random.default_rng(n if seed is None else seed)
    A0 = rng.standard_normal((n, n)); B0 = np.triu(rng.standard_normal((n, n)))
    A, B = A0.copy(), B0.copy(); Q = np.eye(n); Z = np.eye(n)
    P.stage1(A, B, Q, Z, r)
    A1, B1, Q1, Z1 = A.copy(), B.copy(), Q.copy(), Z.copy()
    LAGM.stage2_wavefronts(A1, B1, Q1, Z1, r, 2)
    stage2_overlapped(A, B, Q, Z, r, 2, gs, near_above)
    u = 2.0 ** -52
    res = max(np.linalg.norm(Q @ A @ Z.T - A0) / np.linalg.norm(A0), np.linalg.norm(Q @ B @ Z.T - B0) / np.linalg.norm(B0)) / u
    same = (np.array_equal(A, A1) and np.array_equal(B, B1))
    diff = max(np.abs(A - A1).max(), np.abs(B - B1).max())
    return res, same, diff, np.abs(np.tril(A, -2)).max(), np.abs(np.tril(B, -1)).max()


if __name__ == "__main__":
    for n, r, gs in [(97, 8, None), (150, 8, None), (200, 16, None), (150, 8, 16), (260, 8, 64)]:
        res, same, diff, la, lb = run(n, r, gs)
        print(f"n={n} r={r} gs={gs}: residual {res:.1f} u, identical to the wavefront order: {same} (max difference {diff:.1e}), below sub-diagonal {la:.1e} / diagonal {lb:.1e}")
    for near_above in (7, 6, 0):
        res, same, diff, la, lb = run(150, 8, None, near_above)
        print(f"negative control, near part only {near_above} rows above the block: residual {res:.3g} u, identical {same}")


# ---------------------------------------------------------------------------------------------------------------------
# Two streams.  H (the latency chain): genh + gen(tau) | near(tau) | genh + gen(tau + 1) | ...      near(tau) of a step =
#   its left reflector on its OWN diagonal blocks A(I, I), B(I, I), then the near part of its right reflector.
# L (the HBM-bound rest): left(tau) without those blocks | far(tau) | left(tau + 1) | ...
# Orderings between the two that the device enforces with events: left(tau) after genh(tau); far(tau) after gen(tau);
# near(tau + 1) after far(tau).  NOTHING else: genh(tau + 1) may run before left(tau) and far(tau) have finished.
# Linearisations checked: `lag_l` = False: L as early as allowed; True: L as late as allowed (left(tau), far(tau) run
# after genh + gen(tau + 1)).
def stage2_two_streams(A, B, Q, Z, r, lag=2, gs=None, lag_l=True, own_block_in_near=True, gap_rule=True, merged=False):
    n = A.shape[0]
    tau = 0
    deferred = {}

    def steps_of(tau):
        steps = []
        for j in range(min(tau // lag, n - 3), -1, -1):
            t = tau - lag * j
            p = j + 1 + r * t
            if p > n - 2:
                break
            steps.append((j, t, p, min(p + r, n), j if t == 0 else p - r))
        return steps

    def gen(steps):
        refl = []
        for j, t, p, p1, c0 in steps:
            I = slice(p, p1)
            v, th, beta = P.house(A[I, c0])
            A[p, c0] = beta; A[p + 1:p1, c0] = 0.0
            M = B[I, I] - th * np.outer(v, v @ B[I, I])
            x = P.opposite(M)
            w, tz, _ = P.house(x)
            refl.append((v, th, w, tz))
        return refl

    def near(steps, refl):
        for (j, t, p, p1, c0), (v, th, w, tz) in zip(steps, refl):
            I = slice(p, p1)
            if own_block_in_near:
                A[I, I] -= th * np.outer(v, v @ A[I, I])
                B[I, I] -= th * np.outer(v, v @ B[I, I])
            rb = p1; ra = min(p1 + r, n)
            top = 0 if gs is None else (j // gs) * gs + 1
            cut = max(top, p - (r - 1))
            B[cut:rb, I] -= tz * np.outer(B[cut:rb, I] @ w, w)
            B[p + 1:p1, p] = 0.0
            A[cut:ra, I] -= tz * np.outer(A[cut:ra, I] @ w, w)
            Z[:, I] -= tz * np.outer(Z[:, I] @ w, w)
            if top > 0:
                deferred.setdefault(j // gs, []).append((t, j, p, p1, top, w, tz))

    def left_far(steps, refl):
        left_only(steps, refl)
        far_only(steps, refl)

    def left_only(steps, refl):
        for (j, t, p, p1, c0), (v, th, w, tz) in zip(steps, refl):          # left(tau), the step's own blocks left out
            I = slice(p, p1)
            if own_block_in_near:
                A[I, c0 + 1:p] -= th * np.outer(v, v @ A[I, c0 + 1:p])
                A[I, p1:] -= th * np.outer(v, v @ A[I, p1:])
                B[I, p1:] -= th * np.outer(v, v @ B[I, p1:])
            else:
                A[I, c0 + 1:] -= th * np.outer(v, v @ A[I, c0 + 1:])
                B[I, p:] -= th * np.outer(v, v @ B[I, p:])
            Q[:, I] -= th * np.outer(Q[:, I] @ v, v)

    def far_only(steps, refl):
        for (j, t, p, p1, c0), (v, th, w, tz) in zip(steps, refl):          # far(tau)
            I = slice(p, p1)
            top = 0 if gs is None else (j // gs) * gs + 1
            cut = max(top, p - (r - 1))
            if cut > top:
                B[top:cut, I] -= tz * np.outer(B[top:cut, I] @ w, w)
                A[top:cut, I] -= tz * np.outer(A[top:cut, I] @ w, w)

    waiting = None               # (tau, steps, refl) of the wavefront whose L part has not run yet
    while True:
        steps = steps_of(tau)
        if not steps:
            if tau // lag >= n - 3:
                break
            tau += 1
            continue
        # H waits for L twice: genh(tau) for everything L holds of wavefronts <= tau - 2 (only the tail of the chase
        # has such a gap: its sweeps stop at t = 0, every other wavefront is empty, and sweep j + 1 starts on the rows
        # sweep j has just left -- `gap_rule` False is the negative control), near(tau) for far(tau - 1)
        if waiting is not None and gap_rule and not merged and waiting[0] <= tau - 2:
            left_far(*waiting[1:]); waiting = None
        refl = gen(steps)                                    # H: genh + gen(tau)
        if merged:
            # the order the library runs (ONE stream, no hand-overs): launch M1 = {first half of gen(tau), far(prev)},
            # launch M2 = {second half of gen(tau), left(tau)}, launch near(tau); prev = the last non-empty wavefront
            if waiting is not None:
                far_only(*waiting[1:]); waiting = None
            left_only(steps, refl)
            near(steps, refl)
            waiting = (tau, steps, refl)
            tau += 1
            continue
        if waiting is not None:
            left_far(*waiting[1:]); waiting = None           # L, late: left(tau - 1), far(tau - 1) -- before near(tau)
        if not lag_l:
            left_far(steps, refl)                            # L, early (far(tau) before near(tau): they are independent)
        near(steps, refl)                                    # H
        if lag_l:
            waiting = (tau, steps, refl)
        tau += 1
    if waiting is not None:
        (far_only if merged else left_far)(*waiting[1:])
    for g in sorted(deferred):
        for (t, j, p, p1, top, w, tz) in sorted(deferred[g], key=lambda it: (-it[0], it[1])):
            I = slice(p, p1)
            B[:top, I] -= tz * np.outer(B[:top, I] @ w, w)
            A[:top, I] -= tz * np.outer(A[:top, I] @ w, w)


def run2(n, r, gs, lag_l, own, gap_rule=True, merged=False):
    rng = np.random.default_rng(n)
    A0 = rng.standard_normal((n, n)); B0 = np.triu(rng.standard_normal((n, n)))
    A, B = A0.copy(), B0.copy(); Q = np.eye(n); Z = np.eye(n)
    P.stage1(A, B, Q, Z, r)
    stage2_two_streams(A, B, Q, Z, r, 2, gs, lag_l, own, gap_rule, merged)
    u = 2.0 ** -52
    res = max(np.linalg.norm(Q @ A @ Z.T - A0) / np.linalg.norm(A0), np.linalg.norm(Q @ B @ Z.T - B0) / np.linalg.norm(B0)) / u
    return res, np.abs(np.tril(A, -2)).max(), np.abs(np.tril(B, -1)).max()


if __name__ == "__main__":
    print("one stream, the generation split over the far and the left launch (the library's order):")
    for n, r, gs in [(97, 8, None), (150, 8, 16), (200, 16, 16), (260, 8, 64)]:
        res, la, lb = run2(n, r, gs, True, True, True, True)
        print(f"n={n} r={r} gs={gs}: residual {res:.3g} u, below sub-diagonal {la:.1e} / diagonal {lb:.1e}")
    print("two streams:")
    for n, r, gs in [(97, 8, None), (150, 8, 16), (200, 16, 16), (260, 8, 64)]:
        for lag_l in (False, True):
            res, la, lb = run2(n, r, gs, lag_l, True)
            print(f"n={n} r={r} gs={gs} L {'late ' if lag_l else 'early'}: residual {res:.3g} u, below sub-diagonal {la:.1e} / diagonal {lb:.1e}")
    res, la, lb = run2(150, 8, 16, True, False)
    print(f"negative control (the step's own blocks stay with left(tau) on L, L late): residual {res:.3g} u")
    res, la, lb = run2(150, 8, 16, True, True, False)
    print(f"negative control (genh does not wait for L across an empty wavefront, L late): residual {res:.3g} u")
