"""RBF (numerical rank ~20 of 128): does freezing the numerically null block (no rotation between two directions whose
diagonals are both <= delta ||G||_F, no re-ordering among them) keep the null-space basis continuous from step to step,
so that the extrapolated warm start predicts as well as it does for Matern kernels?
Prints, per step of a 1 %-per-step trajectory: off-diagonal size of the start, rotations per sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import kron as Kr
m = 128
def jacobi(G, Qt, thr, dF, maxsweep=12):
    G = G.copy(); Qt = Qt.copy(); n1 = m - 1; out = []
    for s in range(maxsweep):
        nrot = 0
        for r in range(n1):
            for k in range(m // 2):
                if k == 0: p, q = r, n1
                else: p = (r + k) % n1; q = (r - k) % n1
                g = G[p, q]
                if abs(g) > thr and not (abs(G[p, p]) <= dF and abs(G[q, q]) <= dF):
                    nrot += 1
                    dd = G[q, q] - G[p, p]; o = 2 * g
                    t = abs(o) / (abs(dd) + np.hypot(dd, o))
                    if (dd >= 0) != (o >= 0): t = -t
                    c = 1 / np.sqrt(1 + t * t); sn = t * c
                    Gp = G[:, p].copy(); Gq = G[:, q].copy()
                    G[:, p] = c * Gp - sn * Gq; G[:, q] = sn * Gp + c * Gq
                    Gp = G[p, :].copy(); Gq = G[q, :].copy()
                    G[p, :] = c * Gp - sn * Gq; G[q, :] = sn * Gp + c * Gq
                    a = Qt[p].copy(); b = Qt[q].copy()
                    Qt[p] = c * a - sn * b; Qt[q] = sn * a + c * b
        out.append(nrot)
        if nrot == 0: break
    return G, Qt, out
f = Kr.Factor("points", "rbf", np.linspace(0, 1, m), np.linspace(0, 1, 1024))
def gram(ell):
    d = Kr.dim_prepare(f, ell, 1.0); return d.B @ d.B.T
for delta in (0.0, 2e-15):
    Qprev = Qprev2 = None
    print("delta", delta)
    for t in range(7):
        G = gram(0.2 * 1.01 ** t)
        nG = np.linalg.norm(G); thr = 1e-13 * nG / m; dF = delta * nG
        if Qprev is None: Qs = np.eye(m)
        elif Qprev2 is None: Qs = Qprev
        else:
            Qs = (Qprev @ Qprev2.T) @ Qprev
            Qs = 1.5 * Qs - 0.5 * (Qs @ Qs.T) @ Qs
        Gp = Qs @ G @ Qs.T; Gp = (Gp + Gp.T) / 2
        off0 = np.abs(Gp - np.diag(np.diag(Gp))).max() / nG
        Gd, Qt, rots = jacobi(Gp, Qs, thr, dF, 14)
        lam = np.diag(Gd).copy()
        key = np.maximum(lam, dF) if delta > 0 else lam
        order = np.argsort(-key, kind="stable")
        Qt = Qt[order]; lam = lam[order]
        R = Qt @ G @ Qt.T
        offR = R - np.diag(np.diag(R)); nullm = np.abs(np.diag(R)) <= max(dF, 0)
        NN = nullm[:, None] & nullm[None, :]
        print(f"  step {t}: start off {off0:.1e} rotations {rots} null {int(nullm.sum())} final off_F/(m thr): all {np.linalg.norm(offR)/(m*thr):.2f} outside NN {np.linalg.norm(np.where(NN,0,offR))/(m*thr):.2f}")
        Qprev2, Qprev = Qprev, Qt
