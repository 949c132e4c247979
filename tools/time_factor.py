"""factor_build roofline: HBM-write-bound, 2 outputs x 8 B per (k, p) for A0/dA0 and for K0/dK0.
Measured at m = n = 8192 so the kernel leaves the cache regime, and at the step's own size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import Engine
e = Engine(0)
for kind, basis in (("rbf", "points"), ("matern32", "points"), ("matern12", "b0")):
    for m, n in ((128, 1024), (8192, 8192)):
        x = torch.linspace(0, 1, n, dtype=torch.float64, device="cuda")
        g = torch.linspace(0, 1, m + 1 if basis == "b0" else m, dtype=torch.float64, device="cuda")
        for _ in range(2): e.factor_build(kind, basis, x, g, 0.2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 5
        for _ in range(reps): e.factor_build(kind, basis, x, g, 0.2)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        byt = 16.0 * m * n + 16.0 * m * m
        print(f"{kind:9s} {basis:6s} m={m:5d} n={n:5d}: {dt*1e6:9.1f} us  {byt/dt/1e9:8.1f} GB/s written (algorithmic {byt/1e6:.1f} MB; includes 4 torch.empty allocations per call)")
