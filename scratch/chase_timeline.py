# Cadence of the chase launches of the Schur leg from a rocprofv3 kernel trace (csv):
#   python scratch/chase_timeline.py <dir with *_kernel_trace.csv>
import csv, glob, os, sys
import numpy as np
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ch = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "schur_chase_kernel" in r["Kernel_Name"])
st = np.array([c[0] for c in ch], dtype=np.float64) / 1e3     # us
en = np.array([c[1] for c in ch], dtype=np.float64) / 1e3
per = np.diff(st)
dur = en - st
print(f"chase launches {len(ch)}, span {(en[-1] - st[0]) / 1e6:.3f} s, kernel duration median {np.median(dur):.0f} us mean {dur.mean():.0f} us")
short = per[per <= 600]
print(f"step periods <= 600 us: {len(short)} sum {short.sum() / 1e6:.3f} s median {np.median(short):.0f} us mean {short.mean():.0f} us  p90 {np.percentile(short, 90):.0f} us")
gaps = per[per > 600]
print(f"gaps > 600 us: {len(gaps)} sum {gaps.sum() / 1e6:.3f} s; > 5 ms: {np.sum(gaps > 5000)} sum {gaps[gaps > 5000].sum() / 1e6:.3f} s")
# idle between the end of a chase and the start of the next (near update + waits), within steps
idle = st[1:] - en[:-1]
print(f"within-step idle (next start - this end), periods <= 600 us: median {np.median(idle[per <= 600]):.0f} us mean {idle[per <= 600].mean():.0f} us")
# per-kernel busy time by name over the Schur span
t0, t1 = st[0] * 1e3, en[-1] * 1e3
tot = {}
for r in rows:
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e_ < t0 or s_ > t1: continue
    k = r["Kernel_Name"].split("(")[0][-48:]
    tot[k] = tot.get(k, 0) + (e_ - s_)
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:12]:
    print(f"  {k:50s} {v / 1e9:7.3f} s")
# the ten longest gaps with what followed
order = np.argsort(-per)[:8]
print("longest gaps (ms):", " ".join(f"{per[i] / 1e3:.1f}" for i in order))
# GPU occupancy over the leg: union of all kernel intervals, tail after the last chase launch, idle gaps
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows
            if int(r["End_Timestamp"]) >= t0 and int(r["Start_Timestamp"]) <= t1 + 2e9 and "hess_" not in r["Kernel_Name"])
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps_idle = []
for s_, e_ in iv[1:]:
    if s_ > cur_e:
        busy += cur_e - cur_s; gaps_idle.append(s_ - cur_e); cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
end_all = max(e for _, e in iv)
print(f"leg span (first chase start .. last kernel end) {(end_all - t0) / 1e9:.3f} s; some kernel running {busy / 1e9:.3f} s; "
      f"tail after the last chase kernel {(end_all - t1) / 1e9:.3f} s; idle gaps > 50 us: {sum(1 for g in gaps_idle if g > 5e4)} "
      f"sum {sum(g for g in gaps_idle if g > 5e4) / 1e9:.3f} s")
# concurrency: sum of kernel durations / busy time
tot_dur = sum(e - s for s, e in iv)
print(f"sum of kernel durations {tot_dur / 1e9:.3f} s -> average concurrency {tot_dur / busy:.2f}")
# what runs inside one of the long gaps between chase launches (the 5th longest: a mid-run cycle)
gi = int(np.argsort(-per)[4])
g0, g1 = st[gi] * 1e3, st[gi + 1] * 1e3
print(f"gap {per[gi] / 1e3:.1f} ms: kernels by name inside it (count, total ms, first start ms, last end ms after the gap's first chase)")
agg = {}
for r in rows:
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e_ < g0 or s_ > g1: continue
    k = r["Kernel_Name"].split("(")[0][-40:]
    a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
    a[0] += 1; a[1] += (e_ - s_) / 1e6; a[2] = min(a[2], (s_ - g0) / 1e6); a[3] = max(a[3], (e_ - g0) / 1e6)
for k, a in sorted(agg.items(), key=lambda x: x[1][2]):
    print(f"  {k:42s} n={a[0]:5d} total {a[1]:8.2f} ms  first {a[2]:8.2f}  last end {a[3]:8.2f}")
# the scan kernels mark the host's AED iterations: their start times inside the gap
sc = sorted((int(r["Start_Timestamp"]) - g0) / 1e6 for r in rows if "scan_subdiag" in r["Kernel_Name"] and g0 <= int(r["Start_Timestamp"]) <= g1)
print("  scan kernel starts (ms):", " ".join(f"{x:.1f}" for x in sc))
# bursts of chase launches (period <= 600 us) and the scan kernels (host AED iterations) between them
scan_t = np.array(sorted(int(r["Start_Timestamp"]) for r in rows if "scan_subdiag" in r["Kernel_Name"]), dtype=np.float64) / 1e3
bursts = []
b0 = 0
for i in range(len(per) + 1):
    if i == len(per) or per[i] > 600:
        bursts.append((st[b0], en[i], i - b0 + 1)); b0 = i + 1
print("bursts of chase launches: start ms, length ms, launches | scans during the burst | idle ms and scans before the next burst")
for k, (a, b, cnt) in enumerate(bursts[:60]):
    nxt = bursts[k + 1][0] if k + 1 < len(bursts) else b
    print(f"  {(a - st[0]) / 1e3:9.1f} {(b - a) / 1e3:7.1f} {cnt:5d} | {int(np.sum((scan_t >= a) & (scan_t <= b))):3d} | {(nxt - b) / 1e3:7.1f} {int(np.sum((scan_t > b) & (scan_t < nxt))):3d}")
