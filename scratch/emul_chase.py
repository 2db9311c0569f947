import numpy as np, struct
f=open("gpurun_out/bad_window.bin","rb").read()
n,nb,flags,right=struct.unpack("4i",f[:16]); off=16
W0=np.frombuffer(f[off:off+96*96*8]).reshape(96,96).T.copy(); off+=96*96*8   # column-major ld 96
sr=np.frombuffer(f[off:off+16*nb]); off+=16*nb
si=np.frombuffer(f[off:off+16*nb]); off+=16*nb
Ug=np.frombuffer(f[off:off+96*96*8]).reshape(96,96).T.copy()
print(n,nb,flags,right)
print("shifts", np.c_[sr,si][:8], "...")
print("pairs ok", all(si[2*i]==-si[2*i+1] for i in range(nb)))
print("window hessenberg?", np.abs(np.tril(W0,-2)).max(), "norm", np.linalg.norm(W0))
def refl(x):
    x=np.array(x,float); xn2=(x[1:]**2).sum()
    if xn2==0: return x[0], np.zeros(len(x)-1), 0.0
    a=x[0]; beta=-np.copysign(np.sqrt(a*a+xn2),a); tau=(beta-a)/beta; v=x[1:]/(a-beta); return beta,v,tau
def emulate(W0):
    W=W0.copy(); U=np.eye(n)
    intro=flags&1; fin=flags&2
    left=2-3*nb if intro else 0; rgt = n-2 if fin else right
    worst=0
    for begin in range(left,rgt):
        R=[]
        for i in range(nb):
            j=begin+3*i
            if j>=-1 and j<n-2:
                if j==-1:
                    h=W[:3,:3]
                    s=abs(h[0,0]-sr[2*i+1])+abs(si[2*i+1])+abs(h[1,0])+abs(h[2,0])
                    if s==0: x=[0,0,0]
                    else:
                        h21s=h[1,0]/s; h31s=h[2,0]/s
                        x=[(h[0,0]-sr[2*i])*((h[0,0]-sr[2*i+1])/s)-si[2*i]*(si[2*i+1]/s)+h[0,1]*h21s+h[0,2]*h31s,
                           h21s*(h[0,0]+h[1,1]-sr[2*i]-sr[2*i+1])+h[1,2]*h31s,
                           h31s*(h[0,0]+h[2,2]-sr[2*i]-sr[2*i+1])+h21s*h[2,1]]
                    beta,v,tau=refl(x); ln=3
                else:
                    ln=2 if j==n-3 else 3
                    beta,v,tau=refl(W[j+1:j+1+ln,j]); W[j+1,j]=beta; W[j+2:j+1+ln,j]=0
                if tau!=0: R.append((j+1,ln,np.r_[1,v],tau))
        for (r0,ln,v,tau) in R:
            blk=W[r0:r0+ln,max(r0,0):]; blk-=tau*np.outer(v,v@blk)
        for (r0,ln,v,tau) in R:
            rm=min(n-1,r0+3)
            blk=W[:rm+1,r0:r0+ln]; blk-=tau*np.outer(blk@v,v)
            blk=U[:,r0:r0+ln]; blk-=tau*np.outer(blk@v,v)
        worst=max(worst,np.abs(np.tril(W,-4)).max())
    return W,U
W,U=emulate(W0)
print("emul orth", np.abs(U.T@U-np.eye(n)).max(), "gpu orth", np.abs(Ug.T@Ug-np.eye(n)).max())
print("U diff", np.abs(U-Ug).max())
print("similarity emul", np.linalg.norm(U.T@W0@U-W)/np.linalg.norm(W0))
d=np.abs(Ug.T@Ug-np.eye(n)); idx=np.unravel_index(d.argmax(),d.shape); print("worst at",idx, "col norms bad:", np.where(np.abs((Ug**2).sum(0)-1)>1e-10)[0])
dd=np.abs(U-Ug); print("cols differing:", np.where(dd.max(0)>1e-9)[0])

# which reflector is bad?
W=W0.copy()
bad=[]
intro=flags&1
left=2-3*nb
for begin in range(left,right):
    for i in range(nb):
        j=begin+3*i
        if j>=-1 and j<n-2:
            if j==-1:
                h=W[:3,:3]
                s=abs(h[0,0]-sr[2*i+1])+abs(si[2*i+1])+abs(h[1,0])+abs(h[2,0])
                h21s=h[1,0]/s; h31s=h[2,0]/s
                x=[(h[0,0]-sr[2*i])*((h[0,0]-sr[2*i+1])/s)-si[2*i]*(si[2*i+1]/s)+h[0,1]*h21s+h[0,2]*h31s,
                   h21s*(h[0,0]+h[1,1]-sr[2*i]-sr[2*i+1])+h[1,2]*h31s,
                   h31s*(h[0,0]+h[2,2]-sr[2*i]-sr[2*i+1])+h21s*h[2,1]]
                beta,v,tau=refl(x)
                vv=np.r_[1,v]
                err=abs(tau*(vv@vv)-2)
                print("introduce i",i,"x",x,"beta",beta,"tau",tau,"v",v,"err",err)
    break

print("---- scan all reflectors")
W=W0.copy(); U=np.eye(n)
for begin in range(left,right):
    R=[]
    for i in range(nb):
        j=begin+3*i
        if j>=-1 and j<n-2:
            if j==-1:
                h=W[:3,:3]
                s=abs(h[0,0]-sr[2*i+1])+abs(si[2*i+1])+abs(h[1,0])+abs(h[2,0])
                h21s=h[1,0]/s; h31s=h[2,0]/s
                x=[(h[0,0]-sr[2*i])*((h[0,0]-sr[2*i+1])/s)-si[2*i]*(si[2*i+1]/s)+h[0,1]*h21s+h[0,2]*h31s,
                   h21s*(h[0,0]+h[1,1]-sr[2*i]-sr[2*i+1])+h[1,2]*h31s,
                   h31s*(h[0,0]+h[2,2]-sr[2*i]-sr[2*i+1])+h21s*h[2,1]]
                ln=3
            else:
                ln=3; x=W[j+1:j+4,j].copy()
            beta,v,tau=refl(x)
            if j>=0: W[j+1,j]=beta; W[j+2:j+4,j]=0
            vv=np.r_[1,v]; err=abs(tau*(vv@vv)-2)
            if err>1e-13 and tau!=0: print("begin",begin,"i",i,"j",j,"x",x,"tau",tau,"v",v,"err",err)
            if tau!=0: R.append((j+1,ln,vv,tau))
    for (r0,ln,v,tau) in R:
        blk=W[r0:r0+ln,max(r0,0):]; blk-=tau*np.outer(v,v@blk)
    for (r0,ln,v,tau) in R:
        rm=min(n-1,r0+3)
        blk=W[:rm+1,r0:r0+ln]; blk-=tau*np.outer(blk@v,v)
        blk=U[:,r0:r0+ln]; blk-=tau*np.outer(blk@v,v)
    o=np.abs(U.T@U-np.eye(n)).max()
    if o>1e-12: print("orth lost after begin",begin,o); break
