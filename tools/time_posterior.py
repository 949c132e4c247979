"""Prediction throughput: vggp_posterior (mean + variance at scattered test points), vggp_qv, vggp_readout after a 1024^2 step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense as D
from variational_gridded_gaussian_processes_amd import Engine
import bench
n, m = 1024, 128
kind = sys.argv[1] if len(sys.argv) > 1 else "rbf"
X, y, x1, x2 = D.gen_grid(n, n); del X
eng = Engine(0)
g = np.linspace(0, 1, m)
eng.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda"); yy = eng.sumsq(Y)
th = bench.theta_from_raw(bench.raw_start())
for k in range(3):
    eng.elbo_step(Y, yy, th * (1 + 0.01 * k))
rng = np.random.default_rng(0)
for ns in (8192, 131072, 1048576):
    xs = torch.tensor(rng.uniform(0, 1, (ns, 2)), device="cuda")
    eng.posterior(xs)                      # (the first read-out after a warm step re-runs the finish half cold)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): mean, var = eng.posterior(xs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"posterior ns={ns}: {dt * 1e3:.3f} ms = {ns / dt / 1e6:.1f} M points/s", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): eng.qv()
torch.cuda.synchronize()
print(f"qv: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
