#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pq -- python3 $GRAFT_REPO_ROOT/scratch/r5_ht2.py 8000 > /tmp/pq.log 2>&1
t=$(find /tmp/pq -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
col = [r for r in rows if "ht_qr_col" in r["Kernel_Name"]]
t0 = int(col[0]["Start_Timestamp"]); t1 = int(col[-1]["End_Timestamp"])
other = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "ht_qr_col" not in r["Kernel_Name"] and int(r["Start_Timestamp"]) < t1 and int(r["End_Timestamp"]) > t0]
import bisect
other.sort()
starts = [o[0] for o in other]
def overlapped(s, e):
    k = bisect.bisect_right(starts, e)
    for j in range(max(0, k - 40), k):
        if other[j][1] > s and other[j][0] < e: return True
    return False
import statistics as st
# panel index by order: 65 launches a panel
alone = {}; busy = {}
for idx, r in enumerate(col):
    p, jj = divmod(idx, 65)
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    key = (p // 25, jj // 16)
    (busy if overlapped(s, e) else alone).setdefault(key, []).append(d)
print("rows: panel group (25 panels each); columns: jj 0-15, 16-31, 32-47, 48-64; median us alone / beside another kernel (count)")
for pg in range(5):
    line = []
    for jg in range(5):
        a = alone.get((pg, jg), []); b = busy.get((pg, jg), [])
        line.append(f"{st.median(a) if a else 0:5.1f}({len(a):4d}) / {st.median(b) if b else 0:5.1f}({len(b):4d})")
    print(pg, "  ".join(line))
PY
