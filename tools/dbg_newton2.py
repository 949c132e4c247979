import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import Engine, datagen as D
torch.cuda.set_device(0)
eng = Engine(0)
kind, m, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
X, y, x1, x2 = D.gen_grid(n, n)
g = np.linspace(0, 1, m)
eng.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda")
yy = eng.sumsq(Y)
base = np.array([0.2, 0.25, 1.0, 0.9, 0.01])
for k in range(14):
    theta = base * (1.0 + 0.005 * k) * (1.2 if k >= 9 else 1.0)
    e, gr, info = eng.elbo_step(Y, yy, theta)
    print(k, info["rounds"], info["sweeps"], info["polished"])
