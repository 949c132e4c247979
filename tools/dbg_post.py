import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense as D, kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
e = Engine(0)
n, nk = 25, 11
X, y, x1, x2 = D.gen_grid(n, n)
mesh = np.linspace(0, 1, nk)
e.plan("matern12", "b0", mesh, x1, "matern12", "b0", mesh, x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda"); yy = e.sumsq(Y)
xs = torch.tensor(np.random.default_rng(2).uniform(0, 1, (50, 2)), device="cuda")
f1, f2 = Kr.Factor("b0", "matern12", mesh, x1), Kr.Factor("b0", "matern12", mesh, x2)
th = np.array([0.69, 0.69, 0.69, 0.69, 0.69])
for it in range(9):
    el, g, info = e.elbo_step(Y, yy, th)
    st = Kr.elbo_step(y.reshape(n, n), f1, f2, th)
    pm, pv = e.posterior(xs)
    om, ov = Kr.posterior(st, f1, f2, xs.cpu().numpy())
    pm2, pv2 = e.posterior(xs)
    print(it, "elbo rel", abs(el - st.elbo) / abs(st.elbo), "post var rel", np.abs(pv.cpu().numpy() - ov).max() / ov.max(),
          "2nd call", np.abs(pv2.cpu().numpy() - ov).max() / ov.max(), info["rounds"])
    if it % 3 == 2: th = th * 1.01
