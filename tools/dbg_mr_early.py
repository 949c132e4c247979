"""Multi-rank launch sequence (RCCL communicator of size one) against the oracle along a short trajectory, at a small odd shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense as D, kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
n1, n2, m1, m2 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
X, y, x1, x2 = D.gen_grid(n1, n2); del X
f1 = Kr.Factor("points", "rbf", np.linspace(0, 1, m1), x1); f2 = Kr.Factor("points", "rbf", np.linspace(0, 1, m2), x2)
eng = Engine(0, n_ranks=1, rank=0, unique_id=Engine.unique_id())
eng.plan("rbf", "points", np.linspace(0, 1, m1), x1, "rbf", "points", np.linspace(0, 1, m2), x2, warm_start=True)
Y = torch.tensor(y.reshape(n2, n1), device="cuda"); yy = eng.sumsq(Y)
for k in range(8):
    th = np.array([0.2, 0.3, 1.0, 0.8, 0.01]) * (1.0 + 0.01 * k)
    e, g, info = eng.elbo_step(Y, yy, th)
    ref = Kr.elbo_step(y.reshape(n2, n1), f1, f2, th)
    print(k, f"elbo rel {abs(e - ref.elbo) / abs(ref.elbo):.2e} grad rel {np.abs(g - ref.grad).max() / np.abs(ref.grad).max():.2e}", {kk: info[kk] for kk in info if kk in ("rounds", "thin", "subspace", "mode", "warm")})
