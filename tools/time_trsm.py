"""Timing of the triangular-solve building blocks (vggp_trsm, vggp_kron_solve) at the shapes the step and BASELINE metric (ii)
use.  Run under `rocprofv3 --kernel-trace --stats` for per-kernel durations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
e = Engine(0)
for m, ncols in ((128, 2048), (128, 8192), (256, 2048), (1024, 1024)):
    z = np.linspace(0, 1, m)
    K, _ = Kr.points_factor("matern32", z, z, 0.1)
    L = torch.tensor(np.linalg.cholesky(K), device="cuda")
    R = torch.randn(m, ncols, dtype=torch.float64, device="cuda")
    for _ in range(3): e.trsm(L, R)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20
    for _ in range(reps): e.trsm(L, R)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"trsm m={m:5d} ncols={ncols:5d}: {dt*1e6:9.1f} us per call (host-paired, incl. diag-block inverses + copy)  {m*m*ncols/dt/1e12:6.2f} TFLOP/s")
n = 1024
z = np.linspace(0, 1, n)
K1, _ = Kr.points_factor("matern12", z, z, 0.2)
K2, _ = Kr.points_factor("matern32", z, z, 0.05)
L1, L2 = torch.tensor(np.linalg.cholesky(K1), device="cuda"), torch.tensor(np.linalg.cholesky(K2), device="cuda")
Y = torch.randn(n, n, dtype=torch.float64, device="cuda")
for _ in range(3): e.kron_solve(L1, L2, Y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): e.kron_solve(L1, L2, Y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"kron_solve n={n}: {dt*1e6:.1f} us  {4*n**3/dt/1e12:.2f} TFLOP/s")
