"""vggp_elbo_step_scattered at N = 100 000 points, m_d = 32 (bench.py `scattered`): a few steps for rocprofv3 --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import Engine
import bench
N, m = 100000, 32
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, 2)); y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + 0.1 * rng.standard_normal(N)
g = np.linspace(0, 1, m + 1)
eng = Engine(0)
eng.plan("matern12", "b0", g, X[:, 0].copy(), "matern12", "b0", g, X[:, 1].copy(), scattered=True)
yd = torch.tensor(y, device="cuda"); yy = float(y @ y)
th = bench.theta_from_raw(bench.raw_start())
for _ in range(3): eng.elbo_step_scattered(yd, yy, th)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): e, gr, info = eng.elbo_step_scattered(yd, yy, th)
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 5 * 1e3, e)
