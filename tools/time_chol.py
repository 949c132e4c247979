import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
e = Engine(0)
for kind in ("matern32", "rbf"):
    for m in (16, 32, 64, 96, 128):
        z = np.linspace(0, 1, m)
        K, _ = Kr.points_factor(kind, z, z, 0.2)
        Kt = torch.tensor(K, device="cuda")
        for _ in range(3): e.cholesky_inverse(Kt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): L, Li, jit = e.cholesky_inverse(Kt)
        torch.cuda.synchronize()
        print(kind, m, "%.1f us per call" % ((time.perf_counter() - t0) / 20 * 1e6), "jitter", jit)
