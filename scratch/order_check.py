import numpy as np, sys
sys.path.insert(0,'scratch')
from region_check import tasks
N, ilo, ihi, ws, nbc, adv, gap, nb = 1000, 0, 1000, 96, 15, 50, 3, 32
_, total = tasks(ilo, ihi, ws, nbc, adv, gap, nb, 0)
def ops_step(t):
    tk,_ = tasks(ilo, ihi, ws, nbc, adv, gap, nb, t)
    chase=[("W",c,t,lo,lo+n,lo,lo+n) for c,lo,n in tk]
    near=[("L",c,t,lo,lo+n,lo+n,min(N,lo+n+adv)) for c,lo,n in tk if lo+n<N]
    far=[("L",c,t,lo,lo+n,lo+n+adv,N) for c,lo,n in tk if lo+n+adv<N]+[("R",c,t,0,lo,lo,lo+n) for c,lo,n in tk if lo>0]
    return chase,near,far
def run(schedule):
    # per-entry history as separate L-list, R-list with W acting as barrier on both
    hist=[[ [] for _ in range(N)] for _ in range(N)]
    for op in schedule:
        kind,c,t,r0,r1,c0,c1=op
        for i in range(r0,r1):
            row=hist[i]
            for j in range(c0,c1):
                row[j].append((kind,c,t))
    return hist
def canon(h):
    # canonical form: sequence split at W barriers; within a segment L-order and R-order kept separately
    out=[]; L=[]; R=[]
    for k in h:
        if k[0]=="W":
            out.append((tuple(L),tuple(R),k)); L=[]; R=[]
        elif k[0]=="L": L.append(k)
        else: R.append(k)
    out.append((tuple(L),tuple(R),None))
    return out
S1=[]; 
for t in range(total):
    c,nr,f=ops_step(t); S1+=c+nr+f
S2=[]
pend=[]
for t in range(total):
    c,nr,f=ops_step(t)
    S2+=c          # chase(t) before far(t-1)
    S2+=pend       # far(t-1)
    S2+=nr
    pend=f
S2+=pend
h1=run(S1); h2=run(S2)
bad=0
for i in range(N):
    for j in range(N):
        if canon(h1[i][j])!=canon(h2[i][j]):
            bad+=1
            if bad<5: print("entry",i,j,"\n S1",h1[i][j][-6:],"\n S2",h2[i][j][-6:])
print("differing entries:",bad)
