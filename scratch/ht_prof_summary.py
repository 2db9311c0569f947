# Summarises a rocprofv3 --kernel-trace database of scratch/ht_time.py: per-kernel totals and the
# timeline of the first sweep.   python scratch/ht_prof_summary.py gpurun_out/prof_ht/ht_results.db
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'info_kernel_symbol' in t][0]
q = f"select s.kernel_name, d.grid_size_y, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_y order by 4 desc limit 8"
for r in c.execute(q):
    print(f"{r[0][20:60]:40s} gy={r[1]} calls={r[2]:8d} total_ms={r[3]:9.1f} avg_us={r[4]:8.1f}")
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.stream_id from {kd} d join {sym} s on d.kernel_id=s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if 'ht_scan' in r[0]]
if len(idx) > 1 and len(sys.argv) > 2:
    i0, i1 = idx[0], idx[1]; t0 = rows[i0][1]
    for r in rows[i0:i1 + 1]:
        nm = r[0].split('N_1')[1][2:22] if 'N_1' in r[0] else r[0][:20]
        print(f"{nm:22s} start={(r[1]-t0)/1e3:9.1f}us dur={(r[2]-r[1])/1e3:8.1f}us grid=({r[3]},{r[4]}) stream={r[5]}")
    print("sweep length us", (rows[i1][1] - t0) / 1e3)
d = np.array([(r[2] - r[1]) / 1e3 for r in rows if 'ht_chain' in r[0]])
if len(d): print("chain kernel duration percentiles (us) 0/10/50/90/100:", np.percentile(d, [0, 10, 50, 90, 100]))
