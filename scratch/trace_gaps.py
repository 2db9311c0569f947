# Gaps between consecutive kernels of the Hessenberg panel chain (gemv -> colA -> colC -> gemv) from a
# rocprofv3 --kernel-trace csv: python scratch/trace_gaps.py <dir>
import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("hess_gemv", "hess_colA", "hess_colC"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
kind = lambda r: "gemv" if "hess_gemv" in r["Kernel_Name"] else ("colA" if "colA" in r["Kernel_Name"] else "colC")
gaps = {}
durs = {}
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g < 50: gaps.setdefault(kind(a) + "->" + kind(b), []).append(g)
for r in rows: durs.setdefault(kind(r), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(gaps.items()):
    print(f"{k:12s} n={len(v):5d} median gap {statistics.median(v):6.2f} us  mean {sum(v)/len(v):6.2f}  p90 {sorted(v)[int(0.9*len(v))]:6.2f}")
for k, v in durs.items():
    print(f"{k:5s} n={len(v):5d} mean duration {sum(v)/len(v):7.2f} us")
first = int(rows[0]["Start_Timestamp"]); last = int(rows[-1]["End_Timestamp"])
print(f"span {(last-first)/1e6:.3f} s, kernels {sum(sum(v) for v in durs.values())/1e6:.3f} s")
