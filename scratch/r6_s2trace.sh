#!/bin/bash
cd /tmp && export TMPDIR=/tmp
N=${1:-8000}
rm -rf /tmp/pq; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pq -- python3 $GRAFT_REPO_ROOT/scratch/r5_ht2.py $N > /tmp/pq.log 2>&1; tail -5 /tmp/pq.log
t=$(find /tmp/pq -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys, statistics as st
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    for k in ("ht2_m1", "ht2_m2", "ht2_near", "ht2_wy_right", "ht2_group_wy", "ht2_wy_left"):
        if k in nm: return k
    return None
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?"))) for r in rows if short(r["Kernel_Name"])]
import collections
print(collections.Counter(r["Kernel_Name"][:60] for r in rows if "ht2" in r["Kernel_Name"]).most_common(12))
m2 = [e for e in ev if e[2] == "ht2_m2"]
print("wavefronts", len(m2), "stage 2 span %.3f s" % ((m2[-1][1] - m2[0][0]) / 1e9))
for frac in (0.1, 0.5):
    mid = int(len(m2) * frac)
    t0 = m2[mid][0]
    print("timeline of two wavefronts at %.0f %% (us):" % (100 * frac))
    for e in ev:
        if e[0] >= t0 and e[0] < m2[mid + 2][0] and e[2] in ("ht2_m1", "ht2_m2", "ht2_near"):
            print("  %-10s q%-3s grid %-9s start %8.1f end %8.1f dur %7.1f" % (e[2], e[3], e[4], (e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3))
per = [(b[0] - a[0]) / 1e3 for a, b in zip(m2, m2[1:])]
print("period median %.1f mean %.1f us" % (st.median(per), st.mean(per)))
for k in ("ht2_m1", "ht2_m2", "ht2_near"):
    d = [(e[1] - e[0]) / 1e3 for e in ev if e[2] == k]
    q = sorted(d)
    print("  %-10s n %6d median %7.1f mean %7.1f p10 %7.1f p90 %7.1f" % (k, len(d), st.median(d), st.mean(d), q[len(q) // 10], q[9 * len(q) // 10]))
PY
