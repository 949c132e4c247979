"""Per-step wall times and rounds of the ELBO step at fixed hyper-parameters (diagnoses bimodal timing).
usage: step_times.py <kind> [oracle] [qv]"""
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
th = np.array([0.2, 0.2, 1.0, 1.0, 0.0025])
for k in range(3):
    e, gr, info = eng.elbo_step(Y, yy, th * (1 + 0.01 * k))
    if "oracle" in sys.argv:
        f1, f2 = Kr.Factor("points", kind, g, x1), Kr.Factor("points", kind, g, x2)
        Kr.elbo_step(y.reshape(n, n), f1, f2, th)
    if "qv" in sys.argv:
        eng.qv()
out = []
for k in range(30):
    t = th * (1 + 0.002 * (k % 7))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e, gr, info = eng.elbo_step(Y, yy, t)
    dt = (time.perf_counter() - t0) * 1e3
    out.append((round(dt, 2), info["sweeps"][0], info["rounds"][0]))
ts = sorted(o[0] for o in out[3:])
print(sys.argv[1:], os.environ.get("VGGP_EIG_FAST_SWITCH"), os.environ.get("VGGP_EIG_TOL"), "median ms", ts[len(ts)//2], "min", ts[0], "max", ts[-1], "mean rounds", sum(o[2] for o in out[3:]) / len(ts))
