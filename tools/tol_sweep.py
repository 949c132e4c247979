"""Accuracy / time of the ELBO step against the eigensolver threshold (VGGP_EIG_TOL is read at first launch, so one
process per value: run as  tol_sweep.py <tol>  ; prints step time and the relative error against oracle/kron.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense as D, kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
kind = sys.argv[1] if len(sys.argv) > 1 else "rbf"
n, m = 1024, 128
X, y, x1, x2 = D.gen_grid(n, n)
g = np.linspace(0, 1, m)
eng = Engine(0)
eng.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda")
yy = eng.sumsq(Y)
f1, f2 = Kr.Factor("points", kind, g, x1), Kr.Factor("points", kind, g, x2)
th = np.array([0.2, 0.2, 1.0, 1.0, 0.0025])
errs = []
for k in range(6):
    t = th * (1 + 0.01 * k)
    e, gr, info = eng.elbo_step(Y, yy, t)
    if k in (0, 3, 5):
        ref = Kr.elbo_step(y.reshape(n, n), f1, f2, t)
        mean, var = eng.qv()
        rm, rv = Kr.q_v(ref)
        errs.append((abs(e - ref.elbo) / abs(ref.elbo), np.abs(gr - ref.grad).max() / np.abs(ref.grad).max(),
                     np.abs(mean.cpu().numpy() - rm).max() / np.abs(rm).max(), np.abs(var.cpu().numpy() - rv).max() / np.abs(rv).max(),
                     info["sweeps"]))
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(50):
    eng.elbo_step(Y, yy, th * (1 + 0.002 * (k % 7)))
torch.cuda.synchronize()
print(os.environ.get("VGGP_EIG_TOL"), kind, "ms/step %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
for r in errs:
    print("   rel err elbo %.1e grad %.1e qv_mean %.1e qv_var %.1e sweeps %s" % r)
