import csv, sys, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "dgemm" not in r["Kernel_Name"]: continue
            key = r["Kernel_Name"][:70]
            tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in tot.items():
    print(k)
    for c, x in sorted(v.items()): print(f"   {c:32s} {x:.4e}")
