"""CPU: the numpy prototype of the two-stage Hessenberg-triangular reduction (scratch/ht2_proto.py) that
DESIGN.md section 4d's worked estimate rests on -- stage 1 to r-Hessenberg-triangular form (QR of 2r x r blocks
of A, RQ of the bottom rows of the filled diagonal blocks of B), stage 2 a Householder bulge chase with
'opposite' reflectors from the right (Kagstrom, Kressner, Quintana-Orti, Quintana-Orti 2008).  Not a product
path: it documents that the algorithm is backward stable on the inputs the rotation path is kept for
(singular B included) before its cost on the GPU is estimated."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ht2_proto", os.path.join(ROOT, "scratch", "ht2_proto.py"))
P = importlib.util.module_from_spec(spec)
spec.loader.exec_module(P)
U = 2.0 ** -52


@pytest.mark.parametrize("n,r,singular", [(40, 4, False), (97, 8, True), (150, 16, False)])
def test_two_stage_reduction_is_backward_stable(n, r, singular):
    rng = np.random.default_rng(n)
    A0 = rng.standard_normal((n, n)); B0 = np.triu(rng.standard_normal((n, n)))
    if singular:
        B0[10, 10] = 0.0; B0[50, 50] = 0.0
    A, B = A0.copy(), B0.copy(); Q = np.eye(n); Z = np.eye(n)
    P.stage1(A, B, Q, Z, r)
    assert all(A[i, c] == 0.0 for c in range(n) for i in range(c + r + 1, n))      # r-Hessenberg
    assert np.abs(np.tril(B, -1)).max() == 0.0
    log = []
    P.stage2(A, B, Q, Z, r, log)
    assert np.abs(np.tril(A, -2)).max() == 0.0 and np.abs(np.tril(B, -1)).max() == 0.0
    assert np.linalg.norm(Q @ A @ Z.T - A0) <= 100 * U * np.linalg.norm(A0)
    assert np.linalg.norm(Q @ B @ Z.T - B0) <= 100 * U * np.linalg.norm(B0)
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) <= 100 * U * np.sqrt(n)
    assert np.linalg.norm(Z.T @ Z - np.eye(n)) <= 100 * U * np.sqrt(n)
    # the chase of stage 2: ~ n^2 / (2 r) steps of one left and one right reflector of length <= r
    assert abs(len(log) - n * n / (2.0 * r)) <= 0.35 * n * n / (2.0 * r)


# ---- the order the device code runs stage 2 in ------------------------------------------------------------------
spec2 = importlib.util.spec_from_file_location("ht2_lag", os.path.join(ROOT, "scratch", "ht2_lag.py"))
L = importlib.util.module_from_spec(spec2)
spec2.loader.exec_module(L)


@pytest.mark.parametrize("n,r,singular", [(60, 4, False), (97, 8, True), (131, 8, False)])
def test_stage_2_in_wavefronts_of_lag_2(n, r, singular):
    """csrc/ht_twostage.hip runs the steps (sweep j, position t) with t + 2 j = tau together: all reflectors from the
    state at the start of the wavefront, then all left, then all right applications.  That order must give a correct
    reduction (lag 2 does; lag 1 -- neighbouring sweeps one block apart -- must not, or this test checks nothing)."""
    def run(lag):
        rng = np.random.default_rng(n)
        A0 = rng.standard_normal((n, n)); B0 = np.triu(rng.standard_normal((n, n)))
        if singular:
            B0[10, 10] = 0.0; B0[50, 50] = 0.0
        A, B = A0.copy(), B0.copy(); Q = np.eye(n); Z = np.eye(n)
        P.stage1(A, B, Q, Z, r)
        waves = L.stage2_wavefronts(A, B, Q, Z, r, lag)
        return waves, np.abs(np.tril(A, -2)).max(), np.abs(np.tril(B, -1)).max(), \
            np.linalg.norm(Q @ A @ Z.T - A0) / np.linalg.norm(A0), np.linalg.norm(Q @ B @ Z.T - B0) / np.linalg.norm(B0)
    w2, la, lb, ra, rb = run(2)
    assert la == 0.0 and lb == 0.0 and ra <= 100 * U and rb <= 100 * U
    w3, la3, lb3, ra3, rb3 = run(3)
    assert la3 == 0.0 and ra3 <= 100 * U and w2 < 0.75 * w3
    _, _, _, ra1, _ = run(1)
    assert ra1 > 1e6 * U


# ---- which applications of stage 2 may be deferred (round 6) ---------------------------------------------------------
spec3 = importlib.util.spec_from_file_location("ht2_defer", os.path.join(ROOT, "scratch", "ht2_defer.py"))
D = importlib.util.module_from_spec(spec3)
spec3.loader.exec_module(D)


@pytest.mark.parametrize("n,r,gs,singular", [(97, 8, 8, True), (150, 8, 16, False), (200, 16, 64, False)])
def test_stage_2_with_the_top_rows_of_the_right_reflectors_deferred(n, r, gs, singular):
    """csrc/ht_twostage.hip applies the opposite reflector of sweep j at once only to the rows from
    top = (j // GS) GS + 1 on; the rows above take the reflectors of a whole group later, position by position in
    decreasing order (beside Z).  Valid because those rows see no left reflector from the group's first wavefront on
    (scratch/ht2_defer.py has the argument).  The same deferral of the LEFT reflectors' far columns must NOT work --
    a later right reflector that straddles the boundary mixes updated and stale columns -- or this test checks
    nothing: it is the reason the left application stays in the chase."""
    def run(mode):
        rng = np.random.default_rng(n)
        A0 = rng.standard_normal((n, n)); B0 = np.triu(rng.standard_normal((n, n)))
        if singular:
            B0[10, 10] = 0.0; B0[50, 50] = 0.0
        A, B = A0.copy(), B0.copy(); Q = np.eye(n); Z = np.eye(n)
        P.stage1(A, B, Q, Z, r)
        cnt = {"now": 0, "later": 0}
        D.stage2_deferred(A, B, Q, Z, r, gs, mode, count=cnt)
        return (np.abs(np.tril(A, -2)).max(), np.abs(np.tril(B, -1)).max(),
                np.linalg.norm(Q @ A @ Z.T - A0) / np.linalg.norm(A0) / U,
                np.linalg.norm(Q @ B @ Z.T - B0) / np.linalg.norm(B0) / U, cnt)
    la0, lb0, ra0, rb0, c0 = run("none")
    la, lb, ra, rb, c1 = run("right_top")
    assert la == 0.0 and lb == 0.0 and ra < 100 and rb < 100, (la, lb, ra, rb)
    assert abs(ra - ra0) < 5 and abs(rb - rb0) < 5
    # a quarter of the chase's bytes leave it (half of the right application's), what comes back is blocked
    assert c1["now"] < 0.85 * c0["now"] and c1["later"] < 0.15 * c0["now"]
    la, lb, ra, rb, _ = run("left_far")
    assert ra > 1e6 or rb > 1e6


# ---- the generation of wavefront tau + 1 in front of the far part of wavefront tau's right applications (round 6) -----
spec4 = importlib.util.spec_from_file_location("ht2_overlap", os.path.join(ROOT, "scratch", "ht2_overlap.py"))
O = importlib.util.module_from_spec(spec4)
spec4.loader.exec_module(O)


@pytest.mark.parametrize("n,r,gs", [(97, 8, None), (150, 8, 16), (200, 16, 64)])
def test_stage_2_with_the_generation_split_over_the_far_and_the_left_launch(n, r, gs):
    """csrc/ht_twostage.hip runs a wavefront as three launches: M2 = {second halves of its generations | its left
    applications without the steps' own diagonal blocks}, near = {H on those blocks, then the rows from p - (r - 1) on of
    the right applications}, M1 = {first halves of the NEXT wavefront's generations | the rows above: the far part}.  The
    numpy statement of that order (scratch/ht2_overlap.py: generation of tau before far(tau - 1)) must reduce the
    pencil like the plain wavefront order does; with one row less in the near part, or with the own blocks left to the
    left application while it lags behind the next generation, it must NOT -- or this test checks nothing."""
    res, la, lb = O.run2(n, r, gs, True, True, True, True)                       # the library's order
    assert la == 0.0 and lb == 0.0 and res < 100, (res, la, lb)
    res_plain, same, diff, la, lb = O.run(n, r, gs)                               # far(tau) after gen(tau + 1), one stream
    assert la == 0.0 and lb == 0.0 and res_plain < 100 and diff < 1e-8
    res_bad, _, _, _, _ = O.run(n, r, gs, near_above=r - 2)                        # negative control 1
    assert res_bad > 1e6
    if gs is not None:
        # the two-stream variant (measured slower on the device): as early and as late as its
        # hand-overs allow; negative controls: own blocks left on the lagging stream, no wait over an empty wavefront
        for late in (False, True):
            res2, la, lb = O.run2(n, r, gs, late, True)
            assert la == 0.0 and lb == 0.0 and res2 < 100
        assert O.run2(n, r, gs, True, False)[0] > 1e6
        assert O.run2(n, r, gs, True, True, False)[0] > 1e6
