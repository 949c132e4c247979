import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import Engine, datagen as D
kind, m, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
drift = float(sys.argv[4]) if len(sys.argv) > 4 else 0.02
eng = Engine(0)
X, y, x1, x2 = D.gen_grid(n, n); del X
Wn = (np.random.default_rng(3).uniform(size=(n, n)) < 0.7).astype(np.float64)
W = torch.tensor(Wn, device="cuda"); Ym = torch.tensor(y.reshape(n, n), device="cuda") * W
g = np.linspace(0, 1, m)
eng.plan(kind, "points", g, x1, kind, "points", g, x2)
yy = eng.sumsq(Ym)
for k in range(4):
    theta = np.array([0.2, 0.22, 1.0, 0.9, 0.01]) * (1.0 + drift * k)
    try:
        e, gr, info = eng.elbo_step_masked_iter(Ym, W, float(Wn.sum()), yy, theta, n_probes=16)
        print(k, "its", info["rounds"][0], "elbo", e)
    except Exception as ex:
        print(k, "FAILED", str(ex)[:150])
