"""Per-step trace of a warm fit loop: ms, rounds, sweeps, status -- which chain ran (Newton chain: rounds 0, sweeps = iterations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from variational_gridded_gaussian_processes_amd import Engine, datagen as D
torch.cuda.set_device(0)
eng = Engine(0)
n = 1024
kind, m = sys.argv[1], int(sys.argv[2])
X, y, x1, x2 = D.gen_grid(n, n)
Y = torch.tensor(y.reshape(n, n), device="cuda")
yy = float((y * y).sum())
g = np.linspace(0, 1, m)
eng.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
opt = bench.Adam(bench.raw_start(), lr=0.01)
for k in range(int(sys.argv[3]) if len(sys.argv) > 3 else 14):
    raw = opt.x
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e, gr, info = eng.elbo_step(Y, yy, bench.theta_from_raw(raw.copy()))
    dt = (time.perf_counter() - t0) * 1e3
    opt.step(-(gr / (1.0 + np.exp(-raw))))
    print(k, f"{dt:8.3f} ms", "rounds", info["rounds"], "sweeps", info["sweeps"], "polished", info["polished"], f"elbo {e:.6f}")
