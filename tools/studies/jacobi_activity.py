import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np
from oracle import kron as Kr
m=128
def run(kind, dth):
    f = Kr.Factor("points", kind, np.linspace(0,1,m), np.linspace(0,1,1024))
    d0 = Kr.dim_prepare(f, 0.2, 1.0); G0 = d0.B@d0.B.T
    lam,Q = np.linalg.eigh(G0)
    d1 = Kr.dim_prepare(f, 0.2*(1+dth), 1.0); G = Q.T@(d1.B@d1.B.T)@Q
    G=(G+G.T)/2
    thr = 1e-13*np.linalg.norm(G)/m
    n1=m-1
    print(kind, dth, "eig range", lam.max(), np.sort(np.abs(lam))[:3], "thr", thr)
    for sweep in range(8):
        nrot=0; idxs=set(); act_rounds=0; hist=[]
        for r in range(n1):
            na=0
            for k in range(m//2):
                if k==0: p,q=r,n1
                else: p=(r+k)%n1; q=(r-k)%n1
                g=G[p,q]
                if abs(g)>thr:
                    na+=1; idxs.add(p); idxs.add(q)
                    dd=G[q,q]-G[p,p]; o=2*g
                    t=abs(o)/(abs(dd)+np.hypot(dd,o))
                    if (dd>=0)!=(o>=0): t=-t
                    c=1/np.sqrt(1+t*t); s=t*c
                    # rotate rows/cols p,q
                    Gp=G[:,p].copy(); Gq=G[:,q].copy()
                    G[:,p]=c*Gp-s*Gq; G[:,q]=s*Gp+c*Gq
                    Gp=G[p,:].copy(); Gq=G[q,:].copy()
                    G[p,:]=c*Gp-s*Gq; G[q,:]=s*Gp+c*Gq
            nrot+=na; act_rounds+= na>0; hist.append(na)
        off=np.abs(G-np.diag(np.diag(G))).max()
        h=np.array(hist)
        print(f" sweep {sweep}: rotations {nrot} active rounds {act_rounds} distinct idx {len(idxs)} max offdiag {off:.2e} rounds with na<=4: {(h[h>0]<=4).sum()} na<=16: {(h[h>0]<=16).sum()}")
        if nrot==0: break
run("rbf",0.01); run("rbf",0.002); run("matern32",0.01)
