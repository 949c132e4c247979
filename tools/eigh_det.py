import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
e = Engine(0)
for m in (128, 130, 150, 184):
    f = Kr.Factor("points", "matern12", np.linspace(0, 1, m), np.linspace(0, 1, 4 * m + 3))
    d = Kr.dim_prepare(f, 0.2, 1.0)
    G = d.B @ d.B.T
    Gt = torch.tensor(G, device="cuda")
    outs = []
    for it in range(6):
        lam, Qt, sw = e.eigh(Gt)
        lam, Qt = lam.cpu().numpy(), Qt.cpu().numpy()
        res = np.linalg.norm(Qt @ G @ Qt.T - np.diag(lam)) / np.linalg.norm(G)
        outs.append((lam, Qt, res, sw))
    same_lam = all(np.array_equal(outs[0][0], o[0]) for o in outs)
    same_q = all(np.array_equal(outs[0][1], o[1]) for o in outs)
    print(m, "lam bitwise same:", same_lam, "Qt bitwise same:", same_q, "residuals", ["%.1e" % o[2] for o in outs], "sweeps", [o[3] for o in outs])
