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
