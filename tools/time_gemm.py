"""Per-k-tile cost of the GEMM kernel: C = A B with one workgroup per CU or fewer and K swept (no split-K)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variational_gridded_gaussian_processes_amd import Engine
e = Engine(0)
for (M, N) in ((256, 1024), (1024, 1024), (2048, 2048)):
    res = []
    for K in (256, 1024, 4096):
        A = torch.randn(M, K, dtype=torch.float64, device="cuda")
        B = torch.randn(K, N, dtype=torch.float64, device="cuda")
        for _ in range(3): e.gemm(A, B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 20
        for _ in range(reps): e.gemm(A, B)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        res.append((K, dt))
    (k0, t0_), (k1, t1_), (k2, t2_) = res
    per_tile = (t2_ - t1_) / ((k2 - k1) / 16)
    tiles = (M // 64) * (N // 64)
    print(f"M={M} N={N}: workgroups {tiles}; times {[round(t*1e6,1) for _, t in res]} us; marginal {per_tile*1e9:.0f} ns per k-tile of 16; "
          f"large-K rate {2.0*M*N*k2/t2_/1e12:.1f} TFLOP/s")
