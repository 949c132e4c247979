"""Soak test of the warm-start / extrapolation / polish path: an Adam fit loop (lr 0.01, raw = log theta) of several hundred
steps per kernel; every 20th step is compared with the CPU oracle.  Prints the worst relative errors and the polish rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import kron as Kr, dense as D
from variational_gridded_gaussian_processes_amd.engine import Engine
eng = Engine(0)
n, m, steps = 192, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 400
X, y, x1, x2 = D.gen_grid(n, n)
Yh = y.reshape(n, n); Y = torch.tensor(Yh, device="cuda:0")
g = np.linspace(0, 1, m)
rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / (np.max(np.abs(np.asarray(b))) + 1e-300))
for kind in ("matern12", "matern32", "matern52", "rbf"):
    eng.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
    yy = eng.sumsq(Y)
    f1, f2 = Kr.Factor("points", kind, g, x1), Kr.Factor("points", kind, g, x2)
    raw = np.log(np.array([0.3, 0.25, 0.9, 1.2, 0.05]))
    mom = np.zeros(5); vel = np.zeros(5)
    worst_e = worst_g = 0.0; pol = 0; sw = []
    for t in range(1, steps + 1):
        th = np.exp(raw)
        elbo, grad, info = eng.elbo_step(Y, yy, th)
        pol += int(all(info["polished"])); sw.append(sum(info["sweeps"]))
        if t % 20 == 0 or t < 5:
            st = Kr.elbo_step(Yh, f1, f2, th)
            worst_e = max(worst_e, rel(elbo, st.elbo)); worst_g = max(worst_g, rel(grad, st.grad))
        gr = -grad * th                                      # d(-elbo)/d raw
        mom = 0.9 * mom + 0.1 * gr; vel = 0.999 * vel + 0.001 * gr * gr
        raw = raw - 0.01 * (mom / (1 - 0.9 ** t)) / (np.sqrt(vel / (1 - 0.999 ** t)) + 1e-8)
    print(f"{kind:9s} steps {steps}: worst rel err elbo {worst_e:.1e} grad {worst_g:.1e}; polished {pol}/{steps}; mean sweeps (both dims) {np.mean(sw):.2f}; final theta {np.round(np.exp(raw), 4)}")
