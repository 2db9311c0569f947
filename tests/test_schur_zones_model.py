"""CPU: model check of the update zones and the look-ahead of the Schur sweep schedule
(csrc/schur.hip: Driver::sweep_issue, flush_lazy, apply_transform, the look-ahead branch of
schur_device).

The schedule is restated as a list of kernel launches -- (stream, kind, rectangle of H) -- with
the event waits the driver inserts.  Kinds: W = window chase / AED block (reads and writes a
diagonal block), L = left (row) update, R = right (column) update.  Streams execute in order;
an event wait orders everything issued on the waited stream before the record point ahead of
everything issued on the waiting stream after the wait.  Checked:

 1. race freedom: two launches that touch a common entry of H are always ordered (in-place
    kernels read their whole panel before writing it: even commuting L and R updates must not
    run concurrently);
 2. program order: two launches that touch a common entry and do NOT commute (same kind, or
    anything with W) execute in the order of the logical operations they belong to;
 3. completeness: the timely and lazy pieces of a logical update tile exactly the region the
    un-zoned update covers.

Q is not modelled: every update of Q runs on one stream in issue order and nothing else
touches it."""
import itertools

import pytest

WS, NBC, ADV, GAP = 96, 15, 50, 3          # window, bulges per chain, advance, steps between chains


class Schedule:
    def __init__(self, n):
        self.n = n
        self.ops = []               # (stream, kind, r0, r1, c0, c1, logical id)
        self.preds = []             # indices of launches that happen before (direct edges)
        self.last = {}              # stream -> index of its last launch
        self.pending_waits = {}     # stream -> set of launch indices its NEXT launch must follow
        self.logical = 0

    def new_logical(self):
        self.logical += 1
        return self.logical

    def launch(self, stream, kind, r0, r1, c0, c1, lid):
        if r1 <= r0 or c1 <= c0:
            return None
        idx = len(self.ops)
        self.ops.append((stream, kind, r0, r1, c0, c1, lid))
        p = set(self.pending_waits.pop(stream, set()))
        if stream in self.last:
            p.add(self.last[stream])
        self.preds.append(p)
        self.last[stream] = idx
        return idx

    def record(self, stream):
        """event recorded on `stream` now: a marker launch (empty rectangle) that follows the
        stream's last launch AND the waits enqueued since"""
        idx = len(self.ops)
        self.ops.append((stream, "E", 0, 0, 0, 0, 0))
        p = set(self.pending_waits.pop(stream, set()))
        if stream in self.last:
            p.add(self.last[stream])
        self.preds.append(p)
        self.last[stream] = idx
        return idx

    def wait(self, stream, event):
        if event is not None:
            self.pending_waits.setdefault(stream, set()).add(event)
        # waits issued while the stream has its own pending work are transitive through `last`

    def ancestors(self):
        anc = []
        for i, p in enumerate(self.preds):
            m = 0
            for q in p:
                m |= anc[q] | (1 << q)
            anc.append(m)
        return anc


def overlap(a, b):
    return a[2] < b[3] and b[2] < a[3] and a[4] < b[5] and b[4] < a[5]


def commute(a, b):
    return {a[1], b[1]} == {"L", "R"}


def check(sch):
    anc = sch.ancestors()
    ops = sch.ops
    for i, j in itertools.combinations(range(len(ops)), 2):
        a, b = ops[i], ops[j]
        if not overlap(a, b) or a[6] == b[6]:
            continue
        ordered_ij = (anc[j] >> i) & 1
        ordered_ji = (anc[i] >> j) & 1
        assert ordered_ij or ordered_ji, f"race: {a} vs {b}"
        if not commute(a, b):
            first = i if a[6] < b[6] else j
            second = j if first == i else i
            assert (anc[second] >> first) & 1, f"program order violated: {ops[first]} must precede {ops[second]}"


class Driver:
    """the host schedule of schur.hip on the model"""

    def __init__(self, n, nw, guard_windows=3):
        self.n, self.nw, self.gw = n, nw, guard_windows
        self.sch = Schedule(n)
        self.guard = 0
        self.lazy = []              # (logical id of the step's L, of its R, tasks, row_split)
        self.far_done = None
        self.near_done = None
        self.coverage = []          # (logical id, kind, expected rect, [pieces])

    # ---- sweeps -----------------------------------------------------------------------------------
    def tasks(self, ilo, ihi, chains, t):
        spc = -(-(ihi - ilo - WS) // ADV) + 1
        out = []
        for c in range(chains):
            p = t - c * GAP
            if 0 <= p < spc:
                lo = ilo + p * ADV
                out.append((c, lo, ihi - lo if lo + WS >= ihi else WS))
        return out, spc + (chains - 1) * GAP, spc

    def wait_lazy_h(self):
        self.sch.wait("s", self.sch.record("hs"))

    def sweep_begin(self, ilo, ihi, chains):
        self.sw = dict(ilo=ilo, ihi=ihi, chains=chains, t=0, col_split=ihi, issued=0, last_t=-2)
        self.wait_lazy_h()

    def sweep_issue(self, limit):
        sw, s = self.sw, self.sch
        ilo, ihi, chains, col_split = sw["ilo"], sw["ihi"], sw["chains"], sw["col_split"]
        _, total, spc = self.tasks(ilo, ihi, chains, 0)
        while sw["t"] < total:
            t = sw["t"]
            tk, _, _ = self.tasks(ilo, ihi, chains, t)
            if not tk:
                sw["t"] += 1
                continue
            cmin = min(c for c, _, _ in tk)
            cmax = max(c for c, _, _ in tk)
            if limit < ihi and ilo + (t - cmin * GAP) * ADV + WS + ADV > limit:
                return False
            if sw["issued"] > 0 and sw["last_t"] != t - 1:
                s.wait("s", self.far_done)
            lids = {}
            for c, lo, w in tk:
                lids[c] = (s.new_logical(), s.new_logical(), s.new_logical())    # W, L, R of this window step
                s.launch("s", "W", lo, lo + w, lo, lo + w, lids[c][0])
            min_lo = min(lo for _, lo, _ in tk)
            rear = min_lo if cmax == chains - 1 else ilo
            row_split = max(0, min(rear, self.guard))
            if sw["issued"] > 0:
                s.wait("s", self.far_done)
            pieces = {c: ([], []) for c, _, _ in tk}
            for c, lo, w in tk:         # near: columns [lo+w, lo+w+adv) left of ihi
                r = (lo, lo + w, lo + w, min(ihi, lo + w + ADV))
                if s.launch("s", "L", *r, lids[c][1]) is not None:
                    pieces[c][0].append(r)
            self.near_done = s.record("s")
            s.wait("f", self.near_done)
            for c, lo, w in tk:         # timely left: [lo+w+adv, col_split)
                r = (lo, lo + w, lo + w + ADV, min(self.n, col_split))
                if s.launch("f", "L", *r, lids[c][1]) is not None:
                    pieces[c][0].append(r)
            for c, lo, w in tk:         # timely right: rows [T0, lo)
                r = (row_split, lo, lo, lo + w)
                if s.launch("f", "R", *r, lids[c][2]) is not None:
                    pieces[c][1].append(r)
            self.far_done = s.record("f")
            self.lazy.append((tk, lids, row_split, self.far_done, pieces, ihi))
            sw["issued"] += 1
            sw["last_t"] = t
            sw["t"] += 1
        return True

    def flush_lazy(self, col_split):
        s = self.sch
        if not self.lazy:
            return
        s.wait("hs", self.lazy[-1][3])
        for tk, lids, row_split, _, pieces, ihi in self.lazy:
            for c, lo, w in tk:
                # lazy left: right of the near strip (phase A, col_split < ihi) or of the window
                c0 = lo + w + (ADV if col_split < ihi else 0)
                r = (lo, lo + w, max(c0, col_split), self.n)
                if s.launch("hs", "L", *r, lids[c][1]) is not None:
                    pieces[c][0].append(r)
                r = (0, min(lo, row_split), lo, lo + w)
                if s.launch("hs", "R", *r, lids[c][2]) is not None:
                    pieces[c][1].append(r)
                self.coverage.append(("L", (lo, lo + w, lo + w, self.n), pieces[c][0]))
                self.coverage.append(("R", (0, lo, lo, lo + w), pieces[c][1]))
        self.lazy = []

    def sweep_finish(self):
        self.sweep_issue(self.sw["ihi"])
        self.sch.wait("s", self.far_done)
        self.flush_lazy(self.sw["col_split"])

    # ---- AED ----------------------------------------------------------------------------------------
    def aed(self, ts, kw, ihi):
        """window [kw, ihi): block on ts, timely rows [guard, kw) on ts, the rest lazy"""
        s = self.sch
        w_id, r_id, l_id = s.new_logical(), s.new_logical(), s.new_logical()
        s.launch(ts, "W", kw, ihi, kw, ihi, w_id)
        split = min(kw, self.guard)
        pr = [(split, kw, kw, ihi)] if s.launch(ts, "R", split, kw, kw, ihi, r_id) is not None else []
        ready = s.record(ts)
        s.wait("hs", ready)
        pl = [(kw, ihi, ihi, self.n)] if s.launch("hs", "L", kw, ihi, ihi, self.n, l_id) is not None else []
        if s.launch("hs", "R", 0, split, kw, ihi, r_id) is not None:
            pr.append((0, split, kw, ihi))
        self.coverage.append(("R", (0, kw, kw, ihi), pr))
        self.coverage.append(("L", (kw, ihi, ihi, self.n), pl))

    def set_guard(self, r1):
        r1 = max(0, r1)
        if r1 < self.guard:
            self.wait_lazy_h()
        self.guard = r1

    # ---- the reduction: sweeps and AED chains, with or without look-ahead --------------------------------
    def run(self, lookahead, deflate=40, chain=3, chains=3):
        n, nw, s = self.n, self.nw, self.sch
        ilo, ihi = 0, n
        first = True
        while ihi - ilo > 3 * WS:
            la = False
            if lookahead and not first:
                r1 = ihi - self.gw * nw
                if r1 - ilo >= 2 * (WS + ADV):
                    self.set_guard(r1)
                    mark = s.record("s")
                    s.wait("a", mark)
                    self.sweep_begin(ilo, ihi, chains)
                    self.sw["col_split"] = r1
                    self.sweep_issue(r1)
                    self.flush_lazy(r1)
                    la = True
            ts = "a" if la else "s"
            for _ in range(chain):                    # the chain of AEDs
                kw = ihi - nw
                if la and kw <= self.guard + 1:
                    break
                if not la and kw < self.guard:
                    self.set_guard(kw - nw)
                self.aed(ts, kw, ihi)
                ihi -= deflate
            if la:
                s.wait("s", s.record("a"))
                self.guard = max(0, min(self.guard, ihi - self.gw * nw))
                self.wait_lazy_h()
                self.sw["ihi"] = ihi
                self.sw["col_split"] = ihi
                self.sweep_finish()
            else:
                self.set_guard(ihi - self.gw * nw)
                self.sweep_begin(ilo, ihi, chains)
                self.sweep_finish()
            first = False
        return self


def tiles_exactly(region, pieces):
    r0, r1, c0, c1 = region
    area = sum((p[1] - p[0]) * (p[3] - p[2]) for p in pieces)
    if area != max(0, r1 - r0) * max(0, c1 - c0):
        return False
    for p in pieces:
        if p[0] < r0 or p[1] > r1 or p[2] < c0 or p[3] > c1:
            return False
    return all(not (a[0] < b[1] and b[0] < a[1] and a[2] < b[3] and b[2] < a[3])
               for a, b in itertools.combinations(pieces, 2))


@pytest.mark.parametrize("lookahead", [False, True])
@pytest.mark.parametrize("n,nw", [(900, 64), (1100, 96)])
def test_zone_schedule_is_race_free_and_ordered(n, nw, lookahead):
    d = Driver(n, nw).run(lookahead)
    assert len(d.sch.ops) > 200
    streams = {o[0] for o in d.sch.ops}
    assert "hs" in streams and (("a" in streams) == lookahead)
    check(d.sch)
    for kind, region, pieces in d.coverage:
        assert tiles_exactly(region, pieces), (kind, region, pieces)


@pytest.mark.parametrize("deflate,chain,chains,gw", [(10, 6, 2, 3), (60, 1, 4, 2), (25, 4, 3, 4), (40, 8, 2, 3)])
def test_zone_schedule_variants(deflate, chain, chains, gw):
    """long and short AED chains (a long one runs into the guard row), strong and weak deflation,
    different guard distances and chain counts"""
    for lookahead in (False, True):
        d = Driver(1000, 64, guard_windows=gw).run(lookahead, deflate=deflate, chain=chain, chains=chains)
        check(d.sch)
        for kind, region, pieces in d.coverage:
            assert tiles_exactly(region, pieces), (kind, region, pieces)


def test_the_checker_sees_a_missing_wait():
    """without the wait for the lazy stream at the start of a sweep, rows that were above the band
    in the previous sweep race with the new sweep's timely updates"""
    d = Driver(900, 64)
    d.wait_lazy_h = lambda: None
    d.run(False)
    with pytest.raises(AssertionError):
        check(d.sch)


def test_the_checker_sees_the_introduction_phase_bug():
    """rows above the rearmost chain are NOT out of reach while chains are still being
    introduced at the top (the bug the row zones had at first): a lazy right update of rows
    [0, lo) of the first chain's window columns meets the near update of the chain that is
    introduced at the top afterwards"""
    d = Driver(900, 64)
    s = d.sch
    d.sweep_begin(0, 900, 2)
    lid = (s.new_logical(), s.new_logical(), s.new_logical())
    s.launch("s", "W", 50, 146, 50, 146, lid[0])            # chain 0, second window
    d.far_done = s.record("s")
    s.wait("hs", d.far_done)
    s.launch("hs", "R", 0, 50, 50, 146, lid[2])             # rows [0, 50) of its columns, lazily
    lid2 = (s.new_logical(), s.new_logical(), s.new_logical())
    s.launch("s", "W", 0, 96, 0, 96, lid2[0])               # chain 1 is introduced at the top ...
    s.launch("s", "L", 0, 96, 96, 146, lid2[1])             # ... and its near update meets those entries
    with pytest.raises(AssertionError):
        check(s)
