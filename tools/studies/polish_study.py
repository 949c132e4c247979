"""After ONE dense Jacobi sweep of an extrapolated warm start: can the second sweep be replaced by a first-order polish
Q <- (I + E + E^2/2) Q,  E_ij = g_ij / (g_ii - g_jj)?  Prints the a-priori bound ||E||_F ||Goff||_F against m*thr and the
actual residual."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import kron as Kr
m = 128
def sweep(G, thr):
    G = G.copy(); n1 = m - 1; nrot = 0
    for r in range(n1):
        for k in range(m // 2):
            if k == 0: p, q = r, n1
            else: p = (r + k) % n1; q = (r - k) % n1
            g = G[p, q]
            if abs(g) > thr:
                nrot += 1
                dd = G[q, q] - G[p, p]; o = 2 * g
                t = abs(o) / (abs(dd) + np.hypot(dd, o))
                if (dd >= 0) != (o >= 0): t = -t
                c = 1 / np.sqrt(1 + t * t); s = t * c
                Gp = G[:, p].copy(); Gq = G[:, q].copy()
                G[:, p] = c * Gp - s * Gq; G[:, q] = s * Gp + c * Gq
                Gp = G[p, :].copy(); Gq = G[q, :].copy()
                G[p, :] = c * Gp - s * Gq; G[q, :] = s * Gp + c * Gq
    return G, nrot
for kind in ("matern32", "matern12", "matern52", "rbf"):
    f = Kr.Factor("points", kind, np.linspace(0, 1, m), np.linspace(0, 1, 1024))
    def basis(ell):
        d = Kr.dim_prepare(f, ell, 1.0); G = d.B @ d.B.T
        lam, Q = np.linalg.eigh(G); return G, Q[:, ::-1].T.copy()
    for step in (0.03, 0.01, 0.003):
        G0, Q0 = basis(0.2); G1, Q1 = basis(0.2 * (1 + step)); G2, _ = basis(0.2 * (1 + 2 * step))
        sg = np.sign(np.sum(Q0 * Q1, axis=1)); sg[sg == 0] = 1; Q1 = Q1 * sg[:, None]
        nG = np.linalg.norm(G2); thr = 1e-13 * nG / m
        Qx = (Q1 @ Q0.T) @ Q1; Qx = 1.5 * Qx - 0.5 * (Qx @ Qx.T) @ Qx
        Gp = Qx @ G2 @ Qx.T; Gp = (Gp + Gp.T) / 2
        for ns in (0, 1):
            if ns: Gp, nrot = sweep(Gp, thr)
            d = np.diag(Gp); off = Gp - np.diag(d)
            sup = np.abs(off) > thr
            with np.errstate(divide='ignore', invalid='ignore'):
                E = np.where(sup, off / (d[:, None] - d[None, :]), 0.0)
            E[~np.isfinite(E)] = 1.0
            R = np.eye(m) + E + 0.5 * E @ E
            Gn = R @ Gp @ R.T; offn = Gn - np.diag(np.diag(Gn))
            bound = np.linalg.norm(E) * np.linalg.norm(off)
            print(f"{kind:9s} step {step:5.3f} sweeps {ns}: sup {int(sup.sum()//2):5d} off_F {np.linalg.norm(off)/nG:.1e} Emax {np.abs(E).max():.1e} E_F {np.linalg.norm(E):.1e}"
                  f" bound/(m thr) {bound/(m*thr):.2e}  actual resid_F/(m thr) {np.linalg.norm(offn)/(m*thr):.2e} max/thr {np.abs(offn).max()/thr:.2e} orth {np.abs(R@R.T-np.eye(m)).max():.1e}")
