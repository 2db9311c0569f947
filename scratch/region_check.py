# Brute-force dependency check of the Schur sweep pipeline: does chase(t+1) touch anything far(t) touches?
def tasks(ilo, ihi, ws, nbc, adv, gap, nbulges, t):
    chains = -(-nbulges // nbc)
    size = ihi - ilo
    spc = -(-(size - ws) // adv) + 1
    out = []
    for c in range(chains):
        p = t - c * gap
        if p < 0 or p >= spc: continue
        lo = ilo + p * adv
        n = ihi - lo if lo + ws >= ihi else ws
        out.append((c, lo, n))
    return out, spc + (chains - 1) * gap
def rects_far(tk, adv, N):
    r = []
    for c, lo, n in tk:
        if lo + n + adv < N: r.append(("farL%d" % c, lo, lo + n, lo + n + adv, N))   # rows lo..lo+n, cols >= lo+n+adv
        if lo > 0: r.append(("farR%d" % c, 0, lo, lo, lo + n))
    return r
def rects_near(tk, adv, N):
    return [("near%d" % c, lo, lo + n, lo + n, min(N, lo + n + adv)) for c, lo, n in tk]
def rects_chase(tk):
    return [("chase%d" % c, lo, lo + n, lo, lo + n) for c, lo, n in tk]
def overlap(a, b):
    return a[1] < b[2] and b[1] < a[2] and a[3] < b[4] and b[3] < a[4]
import sys
N, ilo, ihi, ws, nbc, adv, gap, nb = 1000, 0, 1000, 96, 15, 50, 3, 32
for ihi in (1000, 909, 873, 700, 431):
    _, total = tasks(ilo, ihi, ws, nbc, adv, gap, nb, 0)
    bad = 0
    for t in range(total - 1):
        a, _ = tasks(ilo, ihi, ws, nbc, adv, gap, nb, t)
        b, _ = tasks(ilo, ihi, ws, nbc, adv, gap, nb, t + 1)
        for x in rects_far(a, adv, N):
            for y in rects_chase(b) + []:
                if overlap(x, y):
                    bad += 1
                    if bad < 6: print("ihi", ihi, "step", t, x, "conflicts with", y)
    print("ihi", ihi, "conflicts:", bad)
