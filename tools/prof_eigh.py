"""Profiling helper: run the Jacobi eigensolver alone (cold start) on a 128x128 Gram factor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
m = int(sys.argv[1]) if len(sys.argv) > 1 else 128
kind = sys.argv[2] if len(sys.argv) > 2 else "matern32"
f = Kr.Factor("points", kind, np.linspace(0, 1, m), np.linspace(0, 1, 1024))
d = Kr.dim_prepare(f, 0.2, 1.0)
G = torch.tensor(d.B @ d.B.T, device="cuda")
e = Engine(0)
for _ in range(3):
    lam, Qt, sw = e.eigh(G)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    lam, Qt, sw = e.eigh(G)
torch.cuda.synchronize()
print("eigh m=%d %s: %.3f ms per call, sweeps %d" % (m, kind, (time.perf_counter() - t0) / 5 * 1e3, sw))
