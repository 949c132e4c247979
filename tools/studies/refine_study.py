"""First-order eigenbasis refinement (GEMM-shaped) as a substitute for dense Jacobi sweeps: how many rotations are left
after k refinement iterations of an extrapolated warm start?  (R = I + E + E^2/2, E_ij = g_ij / (g_ii - g_jj))."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'jacobi_ordering.py')).read().split("for kind in")[0])

def refine(Qs, G, thr, emax=0.05):
    Gp = Qs @ G @ Qs.T; Gp = (Gp + Gp.T) / 2
    d = np.diag(Gp)
    den = d[:, None] - d[None, :]
    off = Gp - np.diag(d)
    with np.errstate(divide='ignore', invalid='ignore'):
        E = np.where(np.abs(off) > thr, off / den, 0.0)
    E[~np.isfinite(E)] = 0.0
    bad = np.abs(E) > emax
    E[bad | bad.T] = 0.0
    R = np.eye(m) + E + 0.5 * E @ E
    return R @ Qs, int(bad.sum() // 2), np.abs(E).max()

for kind in ("rbf", "matern32", "matern12"):
    f = Kr.Factor("points", kind, np.linspace(0, 1, m), np.linspace(0, 1, 1024))
    def basis(ell):
        d = Kr.dim_prepare(f, ell, 1.0); G = d.B @ d.B.T
        lam, Q = np.linalg.eigh(G); return G, Q[:, ::-1].T.copy()       # rows = eigenvectors, descending
    for step in (0.01, 0.003):
        G0, Q0 = basis(0.2); G1, Q1 = basis(0.2 * (1 + step)); G2, _ = basis(0.2 * (1 + 2 * step))
        sg = np.sign(np.sum(Q0 * Q1, axis=1)); sg[sg == 0] = 1; Q1 = Q1 * sg[:, None]
        thr = 1e-13 * np.linalg.norm(G2) / m
        U = Q1 @ Q0.T
        Qx = U @ Q1
        Qx = 1.5 * Qx - 0.5 * (Qx @ Qx.T) @ Qx
        for name, Qs in (("plain", Q1), ("extrap", Qx)):
            line = []
            Qc = Qs
            for it in range(4):
                Gp = Qc @ G2 @ Qc.T; Gp = (Gp + Gp.T) / 2
                off = np.abs(Gp - np.diag(np.diag(Gp)))
                nabove = int((off > thr).sum() // 2)
                rots = jacobi(Gp, thr) if it in (0, 1, 2, 3) else None
                orth = np.abs(Qc @ Qc.T - np.eye(m)).max()
                line.append(f"it{it}: off {off.max()/np.linalg.norm(G2):.1e} above {nabove} orth {orth:.1e} rot {rots}")
                Qc, nbad, emx = refine(Qc, G2, thr)
                line[-1] += f" | bad {nbad} Emax {emx:.1e}"
            print(kind, step, name); print("   " + "\n   ".join(line))
