"""Is the run-to-run spread of the scattered step (15 vs 28 ms) tied to the allocation?  Re-plans (new workspace) in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import Engine
import bench
N, m = 100000, 32
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, 2)); y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + 0.1 * rng.standard_normal(N)
g = np.linspace(0, 1, m + 1)
yd = torch.tensor(y, device="cuda"); yy = float(y @ y)
th = bench.theta_from_raw(bench.raw_start())
keep = []
for rep in range(6):
    eng = Engine(0)
    eng.plan("matern12", "b0", g, X[:, 0].copy(), "matern12", "b0", g, X[:, 1].copy(), scattered=True)
    for _ in range(2): eng.elbo_step_scattered(yd, yy, th)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): e, gr, info = eng.elbo_step_scattered(yd, yy, th)
    torch.cuda.synchronize()
    print("rep", rep, "ms per step %.2f" % ((time.perf_counter() - t0) / 4 * 1e3), flush=True)
    keep.append(torch.empty((rep + 1) * 1237 * 1024, dtype=torch.float64, device="cuda"))     # shift later allocations
    eng.close()
