"""RBF: align the numerically null rows of the new eigenbasis with the previous step's (orthogonal Procrustes inside the
null block) before it is stored for the next warm start -- does the extrapolated start then predict well?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'null_freeze_study.py')).read().split("for delta in")[0])
c0 = int(sys.argv[1]) if len(sys.argv) > 1 else 28         # rows >= c0 are treated as the null block
for align in (False, True):
    Qprev = Qprev2 = None
    print("align", align, "null rows from", c0)
    for t in range(7):
        G = gram(0.2 * 1.01 ** t)
        nG = np.linalg.norm(G); thr = 1e-13 * nG / m
        if Qprev is None: Qs = np.eye(m)
        elif Qprev2 is None: Qs = Qprev
        else:
            Qs = (Qprev @ Qprev2.T) @ Qprev
            Qs = 1.5 * Qs - 0.5 * (Qs @ Qs.T) @ Qs
        Gp = Qs @ G @ Qs.T; Gp = (Gp + Gp.T) / 2
        off0 = np.abs(Gp - np.diag(np.diag(Gp))).max() / nG
        Gd, Qt, rots = jacobi(Gp, Qs, thr, 0.0, 14)
        lam = np.diag(Gd).copy()
        order = np.argsort(-lam, kind="stable")
        Qt = Qt[order]
        if align and Qprev is not None:
            M = Qt[c0:] @ Qprev[c0:].T
            U, S, Vt = np.linalg.svd(M)
            Qt[c0:] = (Vt.T @ U.T) @ Qt[c0:]
        print(f"  step {t}: start off {off0:.1e} rotations {rots}")
        Qprev2, Qprev = Qprev, Qt
