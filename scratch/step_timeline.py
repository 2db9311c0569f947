"""Anatomy of a window step of the Schur sweeps from a rocprofv3 kernel trace: python step_timeline.py <dir>"""
import csv, glob, os, sys
import numpy as np
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def pick(sub, excl=None):
    r = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x.get("Queue_Id", "?")) for x in rows
         if sub in x["Kernel_Name"] and (excl is None or excl not in x["Kernel_Name"])]
    return sorted(r)
chase = pick("schur_chase_")
names = sorted({x["Kernel_Name"].split("(")[0] for x in rows if "schur_update" in x["Kernel_Name"]})
print("update kernel names:", names)
near = pick("schur_near_kernel") or pick("schur_update_kernel<2")
far0 = pick("schur_update_kernel<0")
far1 = pick("schur_update_kernel<1")
pair = pick("schur_update_pair")
qk = pick("schur_update_kernel<3")
print(f"chase {len(chase)} near {len(near)} far-left {len(far0)} far-right {len(far1)} lazy pair {len(pair)} q {len(qk)}")
print("queues: chase", {c[2] for c in chase}, "near", {c[2] for c in near}, "far", {c[2] for c in far0} | {c[2] for c in far1},
      "pair", {c[2] for c in pair}, "q", {c[2] for c in qk})
# second reduction only (the last half of the chase launches)
half = len(chase) // 2
chase = chase[half:]
t_lo = chase[0][0]
near = [x for x in near if x[0] >= t_lo]; far0 = [x for x in far0 if x[0] >= t_lo]; far1 = [x for x in far1 if x[0] >= t_lo]
cs = np.array([c[0] for c in chase], float) / 1e3; ce = np.array([c[1] for c in chase], float) / 1e3
ns = np.array([c[0] for c in near], float) / 1e3; ne = np.array([c[1] for c in near], float) / 1e3
m = min(len(cs), len(ns))
print(f"steps analysed {m}")
per = np.diff(cs[:m])
ok = per <= 600
cd = (ce - cs)[:m - 1][ok]
g1 = (ns[:m - 1] - ce[:m - 1])[ok]            # chase end -> near start
nd = (ne - ns)[:m - 1][ok]
g2 = (cs[1:m] - ne[:m - 1])[ok]               # near end -> next chase start
def q(x): return f"median {np.median(x):6.1f}  mean {x.mean():6.1f}  p90 {np.percentile(x, 90):6.1f}"
print(f"step period (<= 600 us: {ok.sum()} steps, sum {per[ok].sum() / 1e6:.3f} s): {q(per[ok])}")
print(f"  chase duration           : {q(cd)}")
print(f"  chase end -> near start  : {q(g1)}")
print(f"  near duration            : {q(nd)}")
print(f"  near end -> next chase   : {q(g2)}")
# far updates: start lag behind the near update of the same step, duration
def lag(fr, label):
    fs = np.array([c[0] for c in fr], float) / 1e3; fe = np.array([c[1] for c in fr], float) / 1e3
    idx = np.searchsorted(ne, fs, side="right") - 1
    good = idx >= 0
    l = fs[good] - ne[idx[good]]
    l = l[l < 600]
    print(f"  {label}: launches {len(fr)}, start lag behind the last finished near update {q(l)}; duration {q(fe - fs)}")
lag(far0, "timely far-left ")
lag(far1, "timely far-right")
# how long after far(t-1) ended did near(t) start (the wait of the critical stream for the far stream)
fe_all = np.sort(np.array([c[1] for c in far0 + far1], float) / 1e3)
idx = np.searchsorted(fe_all, ns[:m], side="right") - 1
good = idx >= 0
w = ns[:m][good] - fe_all[idx[good]]
print(f"  near start - latest far end before it: {q(w[w < 600])}")
