import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import dense as D, kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
n, m = 96, 12
X, y, x1, x2 = D.gen_grid(n, n)
g = np.linspace(0, 1, m)
eng = Engine(0)
eng.plan("matern32", "points", g, x1, "matern32", "points", g, x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda")
yy = eng.sumsq(Y)
f1, f2 = Kr.Factor("points", "matern32", g, x1), Kr.Factor("points", "matern32", g, x2)
th = np.array([0.2, 0.25, 1.0, 1.1, 0.01])
for k in range(40):
    t = th * (1 + 0.01 * k + 0.003 * np.sin(k))
    e, gr, info = eng.elbo_step(Y, yy, t)
    ref = Kr.elbo_step(y.reshape(n, n), f1, f2, t)
    if k % 5 == 0 or k > 36: print(k, e, ref.elbo, np.abs(gr - ref.grad).max() / np.abs(ref.grad).max(), info["sweeps"], info["rounds"])
