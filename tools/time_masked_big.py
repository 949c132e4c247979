"""Masked step at m_d = 128 (M = 16384): the all-observed mask against the Kronecker path, then a 2048 x 2048 grid with 30 % missing."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import dense as D
from variational_gridded_gaussian_processes_amd import Engine
eng = Engine(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 128
th = [0.2, 0.3, 1.0, 0.8, 0.01]
g = np.linspace(0, 1, m)
n = 256
X, y, x1, x2 = D.gen_grid(n, n); del X
Y = torch.tensor(y.reshape(n, n), device="cuda")
eng.plan("matern32", "points", g, x1, "matern32", "points", g, x2)
e0, g0, _ = eng.elbo_step(Y, eng.sumsq(Y), th)
t0 = time.perf_counter()
e1, g1, _ = eng.elbo_step_masked(Y, torch.ones_like(Y), float(n * n), eng.sumsq(Y), th)
print("all-observed n=256: %.3f s  rel elbo %.2e  rel grad %.2e" % (time.perf_counter() - t0, abs(e1 - e0) / abs(e0), np.abs(g1 - g0).max() / np.abs(g0).max()), flush=True)
n = 2048
X, y, x1, x2 = D.gen_grid(n, n); del X
W = torch.tensor((np.random.default_rng(1).uniform(size=(n, n)) < 0.7).astype(np.float64), device="cuda")
Ym = torch.tensor(y.reshape(n, n), device="cuda") * W
eng.plan("matern32", "points", g, x1, "matern32", "points", g, x2)
yy = eng.sumsq(Ym); nobs = float(W.sum().item())
for k in range(3):
    t0 = time.perf_counter()
    e, gr, info = eng.elbo_step_masked(Ym, W, nobs, yy, [t * (1 + 0.01 * k) for t in th])
    print("2048^2 30%% masked m_d=%d: %.3f s  elbo %.6e" % (m, time.perf_counter() - t0, e), flush=True)
