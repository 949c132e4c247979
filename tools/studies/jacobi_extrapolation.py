import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np
import os; exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'jacobi_ordering.py')).read().split("for kind in")[0])
for kind in ("rbf","matern32"):
    f = Kr.Factor("points", kind, np.linspace(0,1,m), np.linspace(0,1,1024))
    def basis(ell):
        d = Kr.dim_prepare(f, ell, 1.0); G=d.B@d.B.T
        lam,Q=np.linalg.eigh(G); return G, Q[:,::-1]
    for step in (0.01, 0.003):
        G0,Q0=basis(0.2); G1,Q1=basis(0.2*(1+step)); G2,_=basis(0.2*(1+2*step))
        # fix signs of Q1 relative to Q0 (eigh sign ambiguity) so the rotation is small
        sg=np.sign(np.sum(Q0*Q1,axis=0)); sg[sg==0]=1; Q1=Q1*sg
        thr=1e-13*np.linalg.norm(G2)/m
        Ga=Q1.T@G2@Q1; Ga=(Ga+Ga.T)/2
        R=Q0.T@Q1
        Qp=Q1@R
        Gb=Qp.T@G2@Qp; Gb=(Gb+Gb.T)/2
        offa=np.abs(Ga-np.diag(np.diag(Ga))).max(); offb=np.abs(Gb-np.diag(np.diag(Gb))).max()
        print(kind, step, "plain warm: off", f"{offa:.2e}", jacobi(Ga,thr), " extrapolated: off", f"{offb:.2e}", jacobi(Gb,thr))
