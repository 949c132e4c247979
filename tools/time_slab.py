"""BASELINE configs[3] per-rank shape: a 1024-row slab of a 4096 x 4096 grid (n1 = 4096, n2_local = 1024), RBF, m_d = 128."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense as D, kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
n1, n2 = 4096, 1024
X, y, x1, x2 = D.gen_grid(n1, n2)
g = np.linspace(0, 1, 128)
eng = Engine(0)
eng.plan("rbf", "points", g, x1, "rbf", "points", g, x2, warm_start=True)
Y = torch.tensor(y.reshape(n2, n1), device="cuda")
yy = eng.sumsq(Y)
th = np.array([0.2, 0.2, 1.0, 1.0, 0.0025])
f1, f2 = Kr.Factor("points", "rbf", g, x1), Kr.Factor("points", "rbf", g, x2)
ref = Kr.elbo_step(y.reshape(n2, n1), f1, f2, th)
e, gr, info = eng.elbo_step(Y, yy, th)
print("parity: elbo rel", abs(e - ref.elbo) / abs(ref.elbo), "grad rel", np.abs(gr - ref.grad).max() / np.abs(ref.grad).max())
for k in range(10): eng.elbo_step(Y, yy, th * (1 + 0.002 * (k % 5)))
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(50): eng.elbo_step(Y, yy, th * (1 + 0.002 * (k % 5)))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"slab 1024 x 4096: {dt*1e3:.3f} ms per step, {n1*n2/dt:.3e} grid-points/s")
