import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np
from oracle import kron as Kr
m=128
def jacobi(G, thr, maxsweep=12):
    G=G.copy(); n1=m-1; out=[]
    for sweep in range(maxsweep):
        nrot=0
        for r in range(n1):
            for k in range(m//2):
                if k==0: p,q=r,n1
                else: p=(r+k)%n1; q=(r-k)%n1
                g=G[p,q]
                if abs(g)>thr:
                    nrot+=1
                    dd=G[q,q]-G[p,p]; o=2*g
                    t=abs(o)/(abs(dd)+np.hypot(dd,o))
                    if (dd>=0)!=(o>=0): t=-t
                    c=1/np.sqrt(1+t*t); s=t*c
                    Gp=G[:,p].copy(); Gq=G[:,q].copy()
                    G[:,p]=c*Gp-s*Gq; G[:,q]=s*Gp+c*Gq
                    Gp=G[p,:].copy(); Gq=G[q,:].copy()
                    G[p,:]=c*Gp-s*Gq; G[q,:]=s*Gp+c*Gq
        out.append(nrot)
        if nrot==0: break
    return out
for kind in ("rbf","matern32"):
    f = Kr.Factor("points", kind, np.linspace(0,1,m), np.linspace(0,1,1024))
    d0 = Kr.dim_prepare(f, 0.2, 1.0); G0 = d0.B@d0.B.T
    lam,Q = np.linalg.eigh(G0)
    d1 = Kr.dim_prepare(f, 0.2*1.01, 1.0); G1=d1.B@d1.B.T
    rng=np.random.default_rng(0)
    for name,perm in (("ascending",np.arange(m)),("descending",np.arange(m)[::-1]),("random",rng.permutation(m))):
        Qp=Q[:,perm]
        G=Qp.T@G1@Qp; G=(G+G.T)/2
        thr=1e-13*np.linalg.norm(G)/m
        print(kind,name,jacobi(G,thr))
