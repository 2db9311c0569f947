import csv, sys, os, glob
d = sys.argv[1]
f = glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>6s} total={int(r['TotalDurationNs'])/1e9:7.3f}s avg={float(r['AverageNs'])/1e3:9.1f}us  {float(r['Percentage']):5.2f}%")
