"""RBF: GEMM-shaped first-order refinement restricted to pairs with at least one numerically non-null index
(diag > dtop ||G||): the null x null block is never rotated -- it is the Schur complement and decays by itself as the
range/null coupling X is eliminated.  Off-diagonal sizes (top block, coupling, null block) and remaining Jacobi rotations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'jacobi_ordering.py')).read().split("for kind in")[0])
kind = "rbf"
f = Kr.Factor("points", kind, np.linspace(0, 1, m), np.linspace(0, 1, 1024))
def basis(ell):
    d = Kr.dim_prepare(f, ell, 1.0); G = d.B @ d.B.T
    lam, Q = np.linalg.eigh(G); return G, Q[:, ::-1].T.copy()
lam_prev = None
for dtop in (1e-10, 1e-13, 1e-15):
  for step in (0.01,):
    G0, Q0 = basis(0.2); G1, Q1 = basis(0.2 * (1 + step)); G2, _ = basis(0.2 * (1 + 2 * step))
    sg = np.sign(np.sum(Q0 * Q1, axis=1)); sg[sg == 0] = 1; Q1 = Q1 * sg[:, None]
    nG = np.linalg.norm(G2); thr = 1e-13 * nG / m
    Qc = (Q1 @ Q0.T) @ Q1; Qc = 1.5 * Qc - 0.5 * (Qc @ Qc.T) @ Qc
    print("dtop", dtop)
    for it in range(5):
        Gp = Qc @ G2 @ Qc.T; Gp = (Gp + Gp.T) / 2
        d = np.diag(Gp); off = Gp - np.diag(d)
        lam1 = np.linalg.eigvalsh(G1)[::-1]
        top = np.arange(m) < int((lam1 > dtop * np.linalg.norm(G1)).sum())      # index-based: the basis is sorted by the previous eigenvalues
        TT = top[:, None] & top[None, :]; NN = (~top)[:, None] & (~top)[None, :]; TN = ~(TT | NN)
        mx = lambda M: np.abs(np.where(M, off, 0)).max() / nG
        rots = jacobi(Gp, thr, 8) if it in (0, 2, 4) else None
        with np.errstate(divide='ignore', invalid='ignore'):
            E = np.where((np.abs(off) > thr) & ~NN, off / (d[:, None] - d[None, :]), 0.0)
        E[~np.isfinite(E)] = 0.0
        big = np.abs(E) > 0.05; E[big | big.T] = 0.0
        print(f"  it{it}: ntop {int(top.sum())} off top-top {mx(TT):.1e} top-null {mx(TN):.1e} null-null {mx(NN):.1e} above thr: TT {int(((np.abs(off)>thr)&TT).sum()//2)} TN {int(((np.abs(off)>thr)&TN).sum()//2)} NN {int(((np.abs(off)>thr)&NN).sum()//2)} Emax {np.abs(E).max():.1e} dropped {int(big.sum()//2)} rot {rots}")
        R = np.eye(m) + E + 0.5 * E @ E
        Qc = R @ Qc
        Qc = 1.5 * Qc - 0.5 * (Qc @ Qc.T) @ Qc
