"""RBF: polish with the numerically null block excluded (pairs with both diagonals <= delta ||G||_F are neither rotated
nor polished).  After s dense sweeps: how many pairs are left, does the a-priori bound hold, what is the actual residual?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'polish_study.py')).read().split("for kind in")[0])
kind = "rbf"
f = Kr.Factor("points", kind, np.linspace(0, 1, m), np.linspace(0, 1, 1024))
def basis(ell):
    d = Kr.dim_prepare(f, ell, 1.0); G = d.B @ d.B.T
    lam, Q = np.linalg.eigh(G); return G, Q[:, ::-1].T.copy()
for step in (0.01, 0.003):
    G0, Q0 = basis(0.2); G1, Q1 = basis(0.2 * (1 + step)); G2, _ = basis(0.2 * (1 + 2 * step))
    lam2 = np.linalg.eigvalsh(G2)[::-1]
    sg = np.sign(np.sum(Q0 * Q1, axis=1)); sg[sg == 0] = 1; Q1 = Q1 * sg[:, None]
    nG = np.linalg.norm(G2); thr = 1e-13 * nG / m
    print("eigenvalues/||G||:", " ".join(f"{v/nG:.0e}" for v in lam2[::8]))
    Qx = (Q1 @ Q0.T) @ Q1; Qx = 1.5 * Qx - 0.5 * (Qx @ Qx.T) @ Qx
    Gp = Qx @ G2 @ Qx.T; Gp = (Gp + Gp.T) / 2
    for ns in range(0, 5):
        if ns: Gp, nrot = sweep(Gp, thr)
        d = np.diag(Gp); off = Gp - np.diag(d)
        for delta in (2e-15, 1e-14):
            null = d <= delta * nG
            NN = null[:, None] & null[None, :]
            sup = (np.abs(off) > thr) & ~NN
            with np.errstate(divide='ignore', invalid='ignore'):
                E = np.where(sup, off / (d[:, None] - d[None, :]), 0.0)
            E[~np.isfinite(E)] = 1e9
            offx = np.where(NN, 0.0, off)
            bound = np.linalg.norm(E) * np.linalg.norm(offx)
            R = np.eye(m) + E + 0.5 * E @ E
            Gn = R @ Gp @ R.T; offn = np.where(NN, 0.0, Gn - np.diag(np.diag(Gn)))
            print(f"step {step} sweeps {ns} delta {delta:.0e}: null {int(null.sum()):3d} NNnorm/(m thr) {np.linalg.norm(np.where(NN, off, 0))/(m*thr):.2f} sup(nonNN) {int(sup.sum()//2):5d} "
                  f"(all {int((np.abs(off)>thr).sum()//2)}) Emax {np.abs(E).max():.1e} bound/(m thr) {bound/(m*thr):.2e} actual {np.linalg.norm(offn)/(m*thr):.2e}")
