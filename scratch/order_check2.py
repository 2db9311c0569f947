import sys
sys.path.insert(0,'scratch')
from region_check import tasks
def check(N, ilo, ihi, ws, nbc, adv, gap, nb, fixed):
    _, total = tasks(ilo, ihi, ws, nbc, adv, gap, nb, 0)
    def ops_step(t):
        tk,_ = tasks(ilo, ihi, ws, nbc, adv, gap, nb, t)
        chase=[("W",c,t,lo,lo+n,lo,lo+n) for c,lo,n in tk]
        near=[("L",c,t,lo,lo+n,lo+n,min(N,lo+n+adv)) for c,lo,n in tk if lo+n<N]
        far=[("L",c,t,lo,lo+n,lo+n+adv,N) for c,lo,n in tk if lo+n+adv<N]+[("R",c,t,0,lo,lo,lo+n) for c,lo,n in tk if lo>0]
        return chase,near,far
    def run(schedule):
        hist={}
        for kind,c,t,r0,r1,c0,c1 in schedule:
            for i in range(r0,r1):
                for j in range(c0,c1):
                    hist.setdefault((i,j),[]).append((kind,c,t))
        return hist
    def canon(h):
        out=[]; L=[]; R=[]
        for k in h:
            if k[0]=="W": out.append((tuple(L),tuple(R),k)); L=[]; R=[]
            elif k[0]=="L": L.append(k)
            else: R.append(k)
        out.append((tuple(L),tuple(R),None)); return out
    S1=[]; S2=[]; pend=[]; last=None
    for t in range(total):
        c,nr,f=ops_step(t)
        if not c: continue
        S1+=c+nr+f
        if fixed and last is not None and last != t-1:
            S2+=pend; pend=[]            # extra wait: far(last) before chase(t)
        S2+=c; S2+=pend; S2+=nr; pend=f; last=t
    S2+=pend
    h1=run(S1); h2=run(S2)
    return sum(1 for k in h1 if canon(h1[k])!=canon(h2.get(k,[])))
for size in (130, 140, 146, 150, 196, 200, 246, 250, 300, 400):
    for nb in (15, 30, 32, 45):
        a=check(600, 100, 100+size, 96, 15, 50, 3, nb, False); b=check(600, 100, 100+size, 96, 15, 50, 3, nb, True)
        print("size",size,"bulges",nb,"unfixed bad entries",a,"fixed",b)
