"""Warm fit loop at m_d = 256: per-step time and eigensolver diagnostics."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import dense as D
from variational_gridded_gaussian_processes_amd import Engine
import bench
kind = sys.argv[1] if len(sys.argv) > 1 else "rbf"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = 1024
X, y, x1, x2 = D.gen_grid(n, n); del X
eng = Engine(0)
g = np.linspace(0, 1, m)
eng.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda"); yy = eng.sumsq(Y)
opt = bench.Adam(bench.raw_start(), lr=0.01)
for it in range(14):
    raw = opt.x
    t0 = time.perf_counter()
    e, gr, info = eng.elbo_step(Y, yy, bench.theta_from_raw(raw.copy()))
    dt = time.perf_counter() - t0
    opt.step(-(gr / (1.0 + np.exp(-raw))))
    print(kind, m, it, "%.3f ms" % (dt * 1e3), info["sweeps"], info["rounds"], info["polished"], info["jitter"], flush=True)
