"""Long Adam fit loops at the headline shape (RBF, Matern-3/2): how often does the warm chain fall back (rotation rounds > 0, the
step repeated cold), and does the bound keep rising?  usage: soak_fit_loop.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from variational_gridded_gaussian_processes_amd import Engine, datagen as D
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n, m = 1024, 128
X, y, x1, x2 = D.gen_grid(n, n); del X
Y = torch.tensor(y.reshape(n, n), device="cuda")
eng = Engine(0)
for kind, lr in (("rbf", 0.01), ("rbf", 0.05), ("matern32", 0.01)):
    g = np.linspace(0, 1, m)
    eng.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
    yy = eng.sumsq(Y)
    opt = bench.FitLoop5(bench.raw_start(), lr=lr)
    slow, rounds, worst, e0 = 0, 0, 0.0, None
    torch.cuda.synchronize(); t00 = time.perf_counter()
    for k in range(steps):
        t0 = time.perf_counter()
        e, gr, info = eng.elbo_step(Y, yy, opt.theta())
        dt = time.perf_counter() - t0
        opt.update(gr)
        if k == 0: e0 = e
        if k > 5:
            if dt > 0.5e-3: slow += 1
            worst = max(worst, dt)
            rounds += 1 if sum(info["rounds"]) > 0 else 0
    torch.cuda.synchronize()
    print(f"{kind} lr {lr}: {steps} steps, {(time.perf_counter() - t00) / steps * 1e3:.4f} ms per step on average; steps over 0.5 ms: {slow}, with rotation rounds: {rounds}, "
          f"worst {worst * 1e3:.2f} ms; ELBO {e0:.1f} -> {e:.1f}; theta {np.round(opt.theta(), 4)}")
