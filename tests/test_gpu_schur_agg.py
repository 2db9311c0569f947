"""Aggregated lazy updates of the multi-shift sweeps (starneig_amd/csrc/schur_agg.h): the window
factors of a batch of steps, grouped into tiles and applied as wide products, must give what the
factors give when they are applied one by one in issue order (the reference applies every window's
lQ separately: schur/core.c:129-460, common/cpu.c:54-162)."""
import ctypes as C

import numpy as np
import pytest

import starneig_amd as S

pytestmark = pytest.mark.gpu
U_ = 2.0 ** -52


def window_steps(ilo, ihi, ws, nbc, chains, t_first, t_last):
    """(t, c, lo, n) of every window step, in issue order -- schur.hip sweep_issue / make_task."""
    adv = ws - 1 - 3 * nbc
    gap = -(-(ws + adv) // adv)
    size = ihi - ilo
    spc = 1 if size <= ws else -(-(size - ws) // adv) + 1
    out = []
    for t in range(t_first, t_last + 1):
        cmin = 0 if t - spc + 1 <= 0 else -(-(t - spc + 1) // gap)
        cmax = min(chains - 1, t // gap)
        for c in range(cmin, cmax + 1):
            p = t - c * gap
            lo = ilo + p * adv
            n = ihi - lo if lo + ws >= ihi else ws
            out.append((t, c, lo, n))
    return out, spc + (chains - 1) * gap


def random_factors(facs, rng):
    Us = np.zeros((len(facs), 96, 96))
    for i, (_, _, _, n) in enumerate(facs):
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        Us[i, :n, :n] = q.T                      # block i is column-major: Us[i, c, r] = U(r, c)
    return Us


@pytest.mark.parametrize("geom", [
    # (ilo, ihi, ws, nbc, chains, t_first, t_last): the standard geometry (96 / 15 -> adv 50, gap 3)
    (0, 3000, 96, 15, 9, 0, 40),            # introduction at the top, partial tiles
    (100, 2500, 96, 15, 5, 20, 75),         # steady state and the chains leaving at the bottom
    (7, 1500, 96, 15, 13, 5, 33),           # many chains, odd offsets
    (0, 900, 90, 14, 4, 0, 30),             # a narrower window (adv 47, gap 3)
])
@pytest.mark.parametrize("mode", [0, 1])
def test_aggregated_tiles_equal_the_factors_one_by_one(node, geom, mode):
    ilo, ihi, ws, nbc, chains, t_first, t_last = geom
    facs, total = window_steps(ilo, ihi, ws, nbc, chains, t_first, min(t_last, 10 ** 9))
    assert facs
    rng = np.random.default_rng(7)
    Us = random_factors(facs, rng)
    ncols = ihi + 20
    L = S.lib.load_test_hooks()
    dp = C.POINTER(C.c_double)
    L.sn_internal_agg_apply.argtypes = [C.c_int] * 8 + [dp, dp] + [C.c_int] * 5
    if mode == 0:
        m, ld = 333, 340                    # rows of X (odd count: partial row blocks)
        X = np.asfortranarray(rng.standard_normal((ld, ncols)))
        ref = X.copy(order="F")
        for i, (_, _, lo, n) in enumerate(facs):
            ref[:m, lo:lo + n] = ref[:m, lo:lo + n] @ Us[i, :n, :n].T
        out = X.copy(order="F")
        nt = L.sn_internal_agg_apply(0, ilo, ihi, ws, nbc, chains, t_first, t_last,
                                     Us.ctypes.data_as(dp), out.ctypes.data_as(dp), ld, ncols, m, 0, 0)
        assert nt > 0
        assert np.array_equal(out[m:], X[m:])                       # rows past m untouched
    else:
        ld = ihi + 9                        # X holds rows 0..ihi; columns c_lo..c_hi are updated
        c_lo, c_hi, ncols = 11, 11 + 203, 230
        X = np.asfortranarray(rng.standard_normal((ld, ncols)))
        ref = X.copy(order="F")
        for i, (_, _, lo, n) in enumerate(facs):
            ref[lo:lo + n, c_lo:c_hi] = Us[i, :n, :n] @ ref[lo:lo + n, c_lo:c_hi]
        out = X.copy(order="F")
        nt = L.sn_internal_agg_apply(1, ilo, ihi, ws, nbc, chains, t_first, t_last,
                                     Us.ctypes.data_as(dp), out.ctypes.data_as(dp), ld, ncols, 0, c_lo, c_hi)
        assert nt > 0
        assert np.array_equal(out[:, :c_lo], X[:, :c_lo]) and np.array_equal(out[:, c_hi:], X[:, c_hi:])
    assert nt < len(facs)                   # factors really were grouped
    err = np.abs(out - ref).max()
    assert err <= 200 * U_ * np.abs(ref).max(), (err, nt, len(facs))
