"""B1 hats on a padded mesh (empty hats = exact null directions of the Gram matrix): fit-loop step time and eigensolver diagnostics."""
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
for m in (32, 64, 128):
    g = np.linspace(-0.05, 1.05, m)
    eng.plan("matern12", "b1", g, x1, "matern12", "b1", g, x2, warm_start=True)
    opt = bench.FitLoop5(bench.raw_start(), lr=0.01)
    hist = []
    def one():
        e, gr, info = eng.elbo_step(Y, yy, opt.theta()); opt.update(gr); return info
    for _ in range(30): hist.append(one())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): info = one()
    torch.cuda.synchronize()
    print(f"b1 padded m_d={m}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step  sweeps {info['sweeps']} rounds {info['rounds']} polished {info['polished']}  first steps: {[h['sweeps'] for h in hist[:6]]}", flush=True)
