"""ms per step of the Adam fit loop at m_d = 256 (1024 x 1024 grid), per kernel family, with the launch mix of one warm step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from variational_gridded_gaussian_processes_amd import Engine, datagen as D
torch.cuda.set_device(0)
eng = Engine(0)
n = 1024
X, y, x1, x2 = D.gen_grid(n, n)
Y = torch.tensor(y.reshape(n, n), device="cuda")
yy = float((y * y).sum())
for kind in sys.argv[1:] or ["rbf", "matern32"]:
    for m in (192, 256):
        ms = bench.timed_loop(eng, Y, yy, kind, x1, x2, m, warm=True, steps=20, warmup=8)
        e, g, info = eng.elbo_step(Y, yy, bench.THETA0)
        print(kind, m, "ms/step", round(ms, 3), info)
