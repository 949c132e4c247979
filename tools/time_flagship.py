"""The flagship model's configuration (Matern12GriddedGP: Matern-1/2, B0-spline cell features) on the full 1024 x 1024 grid:
ms per fit-loop step for a few inducing counts, and for the VFF / B1 feature builders."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense as D
from variational_gridded_gaussian_processes_amd import Engine
import bench
n = 1024
X, y, x1, x2 = D.gen_grid(n, n); del X
eng = Engine(0)
Y = torch.tensor(y.reshape(n, n), device="cuda"); yy = eng.sumsq(Y)
def loop(basis, g, label):
    eng.plan("matern12", basis, g, x1, "matern12", basis, g, x2, warm_start=True)
    opt = bench.FitLoop5(bench.raw_start(), lr=0.01)
    def one():
        e, gr, info = eng.elbo_step(Y, yy, opt.theta()); opt.update(gr); return info
    for _ in range(30): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): info = one()
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step  sweeps {info['sweeps']} rounds {info['rounds']} polished {info['polished']}", flush=True)
for m in (20, 32, 50, 64, 100, 128):
    loop("b0", np.linspace(0, 1, m + 1), f"b0 cells m_d={m}")
for M in (16, 31, 63):
    a, b = -0.1, 1.1
    loop("vff", np.concatenate([[a, b], D.vff_omegas(M, a, b).double().numpy()]), f"vff M={M} (m_d={2 * M + 1})")
for m in (32, 64, 128):
    loop("b1", np.linspace(-0.05, 1.05, m), f"b1 hats m_d={m}, mesh padded to [-0.05, 1.05]")
    loop("b1", np.linspace(0.0, 1.0, m), f"b1 hats m_d={m}, mesh [0, 1]")
