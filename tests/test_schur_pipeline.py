"""CPU: model check of the two-stream Schur sweep pipeline (csrc/schur.hip, Driver::sweep).

Per step t the critical stream runs chase(t) then near(t), the far stream runs far(t); the
code lets chase(t+1) overlap far(t).  Every kernel is modelled as a rectangle of H with an
operation kind (W = window chase, L = left update, R = right update; L and R commute with
each other, nothing commutes with W or with an operation of its own kind).  The overlapped
schedule must give every entry of H the same operation history as the serial schedule."""
import pytest


def tasks(ilo, ihi, ws, nbc, adv, gap, nbulges, t):
    """make_task() of schur.hip for all chains in flight at step t"""
    chains = -(-nbulges // nbc)
    spc = -(-(ihi - ilo - ws) // adv) + 1
    out = []
    for c in range(chains):
        p = t - c * gap
        if p < 0 or p >= spc:
            continue
        lo = ilo + p * adv
        out.append((c, lo, ihi - lo if lo + ws >= ihi else ws))
    return out, spc + (chains - 1) * gap


def histories(N, ilo, ihi, ws, nbc, adv, gap, nbulges, overlapped, with_rule):
    _, total = tasks(ilo, ihi, ws, nbc, adv, gap, nbulges, 0)
    sched, pend, last = [], [], None
    for t in range(total):
        tk, _ = tasks(ilo, ihi, ws, nbc, adv, gap, nbulges, t)
        if not tk:
            continue
        chase = [("W", c, t, lo, lo + n, lo, lo + n) for c, lo, n in tk]
        near = [("L", c, t, lo, lo + n, lo + n, min(N, lo + n + adv)) for c, lo, n in tk if lo + n < N]
        far = [("L", c, t, lo, lo + n, lo + n + adv, N) for c, lo, n in tk if lo + n + adv < N] + \
              [("R", c, t, 0, lo, lo, lo + n) for c, lo, n in tk if lo > 0]
        if not overlapped:
            sched += chase + near + far
            continue
        if with_rule and last is not None and last != t - 1:
            sched += pend; pend = []          # the extra wait of Driver::sweep
        sched += chase + pend + near          # chase(t) ahead of far(t-1); near(t) waits for it
        pend, last = far, t
    sched += pend
    hist = {}
    for kind, c, t, r0, r1, c0, c1 in sched:
        for i in range(r0, r1):
            for j in range(c0, c1):
                hist.setdefault((i, j), []).append((kind, c, t))

    def canon(h):
        out, L, R = [], [], []
        for k in h:
            if k[0] == "W":
                out.append((tuple(L), tuple(R), k)); L, R = [], []
            elif k[0] == "L":
                L.append(k)
            else:
                R.append(k)
        return out + [(tuple(L), tuple(R), None)]
    return {k: canon(v) for k, v in hist.items()}


@pytest.mark.parametrize("size", [130, 146, 150, 200, 260, 420])
@pytest.mark.parametrize("nbulges", [15, 32, 45])
def test_overlapped_schedule_equals_serial(size, nbulges):
    args = (520, 60, 60 + size, 96, 15, 50, 3, nbulges)
    serial = histories(*args, overlapped=False, with_rule=True)
    assert histories(*args, overlapped=True, with_rule=True) == serial


def test_rule_is_needed_on_short_blocks():
    args = (520, 60, 60 + 140, 96, 15, 50, 3, 32)
    assert histories(*args, overlapped=True, with_rule=False) != histories(*args, overlapped=False, with_rule=True)
