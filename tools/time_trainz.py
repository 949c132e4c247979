import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import dense as D
from variational_gridded_gaussian_processes_amd import Engine
import bench
n, m = 1024, 128
X, y, x1, x2 = D.gen_grid(n, n); del X
eng = Engine(0)
z = [np.linspace(0, 1, m), np.linspace(0, 1, m)]
eng.plan("matern32", "points", z[0], x1, "matern32", "points", z[1], x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda"); yy = eng.sumsq(Y)
opt = bench.Adam(bench.raw_start(), lr=0.01)
T = dict(step=0.0, zgrad=0.0, d2h=0.0, setz=0.0)
DO_ZG = os.environ.get('TZ_ZGRAD', '1') == '1'; DO_SET = os.environ.get('TZ_SET', '1') == '1'
for it in range(80):
    raw = opt.x
    t0 = time.perf_counter(); e, gr, info = eng.elbo_step(Y, yy, bench.theta_from_raw(raw.copy())); t1 = time.perf_counter()
    if DO_ZG:
        g1, g2 = eng.zgrad(Y)
    t2 = time.perf_counter()
    a, b = (g1.cpu().numpy(), g2.cpu().numpy()) if DO_ZG else (np.ones(m), np.ones(m)); t3 = time.perf_counter()
    dl = float(os.environ.get('TZ_DELTA', '1e-6')); z[0] = z[0] + dl * np.sign(a); z[1] = z[1] + dl * np.sign(b)
    if DO_SET:
        eng.set_inducing(0, z[0]); eng.set_inducing(1, z[1])
    t4 = time.perf_counter()
    opt.step(-(gr / (1.0 + np.exp(-raw))))
    if it >= 20:
        T["step"] += t1 - t0; T["zgrad"] += t2 - t1; T["d2h"] += t3 - t2; T["setz"] += t4 - t3
print({k: round(v / 60 * 1e3, 4) for k, v in T.items()}, "ms per iteration", info)
