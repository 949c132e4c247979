"""Cold steps (warm start off) of the RBF headline shape: range finder + thin chain against the oracle, and their time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense as D, kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
n, m = int(sys.argv[1]), int(sys.argv[2])
ells = [float(v) for v in sys.argv[3:]] or [0.2]
X, y, x1, x2 = D.gen_grid(n, n); del X
g = np.linspace(0, 1, m)
f1 = Kr.Factor("points", "rbf", g, x1); f2 = Kr.Factor("points", "rbf", g, x2)
eng = Engine(0)
eng.plan("rbf", "points", g, x1, "rbf", "points", g, x2, warm_start=False)
Y = torch.tensor(y.reshape(n, n), device="cuda"); yy = eng.sumsq(Y)
for ell in ells:
    for k in range(3):
        th = np.array([ell, ell * 1.1, 1.0, 0.8, 0.01]) * (1.0 + 0.01 * k)
        e, gr, info = eng.elbo_step(Y, yy, th)
        ref = Kr.elbo_step(y.reshape(n, n), f1, f2, th)
        print(f"ell {ell} step {k}: elbo rel {abs(e - ref.elbo) / abs(ref.elbo):.2e} grad rel {np.abs(gr - ref.grad).max() / np.abs(ref.grad).max():.2e} rounds {info['rounds']}")
    mean, var = eng.qv()
    st = Kr.elbo_step(y.reshape(n, n), f1, f2, th)
    qm, qvv = Kr.q_v(st)
    print(f"   q(v): mean {np.abs(mean.cpu().numpy().ravel() - qm.ravel()).max() / np.abs(qm).max():.2e} var {np.abs(var.cpu().numpy().ravel() - qvv.ravel()).max() / np.abs(qvv).max():.2e}")
th = np.array([ells[0], ells[0] * 1.1, 1.0, 0.8, 0.01])
for _ in range(10): eng.elbo_step(Y, yy, th)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100): eng.elbo_step(Y, yy, th * (1 + 0.001 * i))
torch.cuda.synchronize()
print(f"cold step: {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms")
