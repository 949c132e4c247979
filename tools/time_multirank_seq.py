"""Step time of the MULTI-RANK launch sequence (partials graph -> ncclAllReduce on the step's stream -> finish graph) on one GPU,
through an RCCL communicator of size one, next to the fused single-rank step: what the sequence itself costs before any link
latency.  1024 x 1024 RBF grid, m_d = 128, Adam fit loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense as D
from variational_gridded_gaussian_processes_amd import Engine
import bench
n, m = 1024, 128
X, y, x1, x2 = D.gen_grid(n, n); del X
g = np.linspace(0, 1, m)
for name, eng in (("fused single-rank step", Engine(0)), ("multi-rank sequence, RCCL communicator of size 1", Engine(0, n_ranks=1, rank=0, unique_id=Engine.unique_id()))):
    eng.plan("rbf", "points", g, x1, "rbf", "points", g, x2, warm_start=True)
    Y = torch.tensor(y.reshape(n, n), device="cuda")
    yy = eng.sumsq(Y)
    opt = bench.Adam(bench.raw_start(), lr=0.01)
    def one():
        raw = opt.x
        e, gr, info = eng.elbo_step(Y, yy, bench.theta_from_raw(raw.copy()))
        opt.step(-(gr / (1.0 + np.exp(-raw))))
    for _ in range(30): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): one()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per step  (transport: {eng.transport})")
    eng.close()
