#!/bin/bash
cd /tmp && export TMPDIR=/tmp
N=${1:-8000}
rm -rf /tmp/pq; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pq -- python3 $GRAFT_REPO_ROOT/scratch/r5_ht2.py $N > /tmp/pq.log 2>&1
t=$(find /tmp/pq -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys, statistics as st
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    for k in ("ht2_factor", "ht2_wy_left", "ht2_wy_right2", "ht2_wy_right", "ht2_m2"):
        if k in nm: return k
    return None
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows if short(r["Kernel_Name"])]
first_m2 = min(e[0] for e in ev if e[2] == "ht2_m2")
ev = [e for e in ev if e[0] < first_m2]
fac = [e for e in ev if e[2] == "ht2_factor"]
print("factor launches", len(fac), "stage 1 span %.3f s" % ((fac[-1][1] - fac[0][0]) / 1e9))
mid = len(fac) // 2
t0 = fac[mid][0]
print("timeline around the middle (us):")
for e in ev:
    if e[0] >= t0 - 1000 and e[0] < fac[mid + 3][0]:
        print("  %-14s q%-3s start %8.1f end %8.1f dur %7.1f" % (e[2], e[3], (e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3))
for k in ("ht2_factor", "ht2_wy_left", "ht2_wy_right", "ht2_wy_right2"):
    d = [(e[1] - e[0]) / 1e3 for e in ev if e[2] == k]
    print("  %-14s n %6d median %7.1f mean %7.1f" % (k, len(d), st.median(d), st.mean(d)))
per = [(b[0] - a[0]) / 1e3 for a, b in zip(fac, fac[1:])]
print("period between factor launches: median %.1f mean %.1f" % (st.median(per), st.mean(per)))
PY
