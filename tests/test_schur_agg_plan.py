"""The tiling behind the aggregated lazy updates (starneig_amd/csrc/schur_agg.h, agg_plan) -- host
logic, no GPU: grouping the window factors of a sweep into tiles and applying tile after tile must
be a legal reordering of the issue order (only factors on overlapping columns fail to commute), and
the tiles of one wavefront must touch disjoint columns."""
import ctypes as C

import numpy as np
import pytest

import starneig_amd as S


def issue_order(ilo, ihi, ws, nbc, chains, t_first, t_last):
    adv = ws - 1 - 3 * nbc
    gap = -(-(ws + adv) // adv)
    size = ihi - ilo
    spc = 1 if size <= ws else -(-(size - ws) // adv) + 1
    out = []
    for t in range(t_first, t_last + 1):
        cmin = 0 if t - spc + 1 <= 0 else -(-(t - spc + 1) // gap)
        cmax = min(chains - 1, t // gap)
        for c in range(cmin, cmax + 1):
            lo = ilo + (t - c * gap) * adv
            out.append((t, c, lo, ihi - lo if lo + ws >= ihi else ws))
    return out


@pytest.mark.parametrize("geom", [
    (0, 4000, 96, 15, 29, 0, 200),          # a sweep of the n = 20000 configuration, from the top
    (3, 2600, 96, 15, 8, 30, 90),
    (0, 1100, 96, 15, 5, 0, 40),            # short block: chains leave while others still enter
    (0, 700, 88, 14, 6, 0, 30),
    (5, 640, 90, 14, 3, 0, 25),             # an active block that is not a multiple of anything
])
@pytest.mark.parametrize("shape", [(0, 0), (8, 1), (3, 2)])       # (0, 0): the 5 x 4 tiles of the aggregated updates; (8, 1): chains
def test_tile_order_is_a_legal_reordering(geom, shape):
    ilo, ihi, ws, nbc, chains, t_first, t_last = geom
    facs = issue_order(*geom)
    L = S.lib.load_test_hooks()
    L.sn_internal_agg_plan.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
    out = (C.c_int * (3 * len(facs)))()
    nf = L.sn_internal_agg_plan(ilo, ihi, ws, nbc, chains, t_first, t_last, out, len(facs), shape[0], shape[1])
    assert nf == len(facs)
    tile = np.array(out[0::3]); wave = np.array(out[1::3]); lo = np.array(out[2::3])
    assert np.array_equal(lo, [f[2] for f in facs])
    assert tile.max() + 1 < len(facs) / 2               # factors are grouped
    # every pair of factors on overlapping columns keeps its issue order under (tile, issue index)
    order = sorted(range(nf), key=lambda i: (tile[i], i))
    pos = np.empty(nf, dtype=int); pos[order] = np.arange(nf)
    ends = lo + np.array([f[3] for f in facs])
    for i in range(nf):
        later = np.arange(i + 1, nf)
        overlap = later[(lo[later] < ends[i]) & (lo[i] < ends[later])]
        assert np.all(pos[overlap] > pos[i]), (i, facs[i])
    # tiles are numbered wavefront by wavefront, and the tiles of a wavefront touch disjoint columns
    assert np.all(np.diff(wave[np.argsort(tile, kind="stable")]) >= 0)
    for w in np.unique(wave):
        spans = []
        for t in np.unique(tile[wave == w]):
            spans.append((lo[tile == t].min(), ends[tile == t].max()))
        spans.sort()
        for a, b in zip(spans, spans[1:]):
            assert a[1] <= b[0], (w, a, b)
    # a tile never holds more than 5 x 4 factors; the tiles of the aggregated updates fit 448 columns
    for t in np.unique(tile):
        assert (tile == t).sum() <= 20
        if shape == (0, 0):
            assert ends[tile == t].max() - lo[tile == t].min() <= 448


def test_wide_geometries_are_not_aggregated():
    L = S.lib.load_test_hooks()
    L.sn_internal_agg_plan.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
    out = (C.c_int * 30)()
    assert L.sn_internal_agg_plan(0, 900, 96, 10, 4, 0, 1, out, 10, 0, 0) == -1      # adv 65: a tile would span 486 columns
