import csv,glob,sys,statistics
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=lambda r:(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
g=[r for r in rows if "hess_gemv" in r["Kernel_Name"]]
a=[r for r in rows if "hess_colA" in r["Kernel_Name"]]
c=[r for r in rows if "hess_colC" in r["Kernel_Name"]]
print("launches gemv %d colA %d colC %d"%(len(g),len(a),len(c)))
for j in (1,10,50,100,150,200,250,300,310):
    if j<len(a): print("j=%3d colA %.1f us colC %.1f us gemv %.1f us"%(j,d(a[j-1]),d(c[j]),d(g[j])))
gaps=[(int(g[k+1]["Start_Timestamp"])-int(g[k]["End_Timestamp"]))/1e3 for k in range(min(len(g)-1,311))]
print("gemv->gemv gap: mean %.1f us median %.1f"%(sum(gaps)/len(gaps),statistics.median(gaps)))
# boundary costs inside the gap: gemv end -> colA start, colA end -> colC start, colC end -> gemv start
k=200
print("boundaries at j=200: gemv->colA %.2f, colA->colC %.2f, colC->gemv %.2f us"%(
 (int(a[k]["Start_Timestamp"])-int(g[k]["End_Timestamp"]))/1e3,
 (int(c[k+1]["Start_Timestamp"])-int(a[k]["End_Timestamp"]))/1e3,
 (int(g[k+1]["Start_Timestamp"])-int(c[k+1]["End_Timestamp"]))/1e3))
