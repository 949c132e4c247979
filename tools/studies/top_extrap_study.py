"""RBF warm start: extrapolate only the K leading (well-determined) eigenvectors linearly, 2 q(t-1) - q(t-2), keep the other
rows of Q(t-1), orthogonalise those against the predicted leading rows, then one Newton-Schulz step.  Compared with the
product form U Q(t-1), U = Q(t-1) Q(t-2)^T."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'null_freeze_study.py')).read().split("for delta in")[0])
kind = sys.argv[2] if len(sys.argv) > 2 else "rbf"
f = Kr.Factor("points", kind, np.linspace(0, 1, m), np.linspace(0, 1, 1024))
for K in [int(v) for v in sys.argv[1].split(",")]:
    Qprev = Qprev2 = None
    print("K", K)
    for t in range(7):
        G = gram(0.2 * 1.01 ** t)
        nG = np.linalg.norm(G); thr = 1e-13 * nG / m
        if Qprev is None: Qs = np.eye(m)
        elif Qprev2 is None: Qs = Qprev
        elif K == 0:
            Qs = (Qprev @ Qprev2.T) @ Qprev
            Qs = 1.5 * Qs - 0.5 * (Qs @ Qs.T) @ Qs
        else:
            Qs = Qprev.copy()
            T = 2 * Qprev[:K] - Qprev2[:K]
            T = 1.5 * T - 0.5 * (T @ T.T) @ T
            Qs[:K] = T
            Qs[K:] = Qs[K:] - (Qs[K:] @ T.T) @ T
            Qs = 1.5 * Qs - 0.5 * (Qs @ Qs.T) @ Qs
        Gp = Qs @ G @ Qs.T; Gp = (Gp + Gp.T) / 2
        off0 = np.abs(Gp - np.diag(np.diag(Gp))).max() / nG
        Gd, Qt, rots = jacobi(Gp, Qs, thr, 0.0, 14)
        lam = np.diag(Gd).copy()
        order = np.argsort(-lam, kind="stable")
        Qt = Qt[order]
        # sign continuity of the leading rows with the previous basis
        if Qprev is not None:
            sg = np.sign(np.sum(Qt * Qprev, axis=1)); sg[sg == 0] = 1
            Qt = Qt * sg[:, None]
        print(f"  step {t}: start off {off0:.1e} rotations {rots}")
        Qprev2, Qprev = Qprev, Qt
