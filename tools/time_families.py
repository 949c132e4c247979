"""ms per step of 300-step Adam fit loops for the kernel / feature families of VERDICT r2 item 4 (1024 x 1024 grid):
Matern-5/2 and Matern-3/2 points at m_d = 128, 256; B1 hats on the padded mesh (the reference's ASVGP) at 128; VFF 127."""
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

def loop(plan, steps=int(os.environ.get("TF_STEPS", 120)), warmup=30):
    plan()
    opt = bench.Adam(bench.raw_start(), lr=0.01)
    stats = {"rounds": 0, "cold": 0}
    def one():
        raw = opt.x
        e, gr, info = eng.elbo_step(Y, yy, bench.theta_from_raw(raw.copy()))
        opt.step(-(gr / (1.0 + np.exp(-raw))))
        stats["rounds"] += sum(info["rounds"])
        return e, info
    for _ in range(warmup): one()
    torch.cuda.synchronize(); stats["rounds"] = 0
    t0 = time.perf_counter()
    for _ in range(steps): e, info = one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, e, stats["rounds"] / steps, info

cases = []
for kind in ("matern52", "matern32", "matern12"):
    for m in (128, 256):
        g = np.linspace(0, 1, m)
        cases.append((f"{kind} points m={m}", lambda g=g, kind=kind: eng.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)))
m = 128
pad = 8
d = 1.0 / (m - 1 - 2 * pad)
gb1 = np.linspace(-pad * d, 1 + pad * d, m)
cases.append(("b1 padded m=128", lambda: eng.plan("matern12", "b1", gb1, x1, "matern12", "b1", gb1, x2, warm_start=True)))
M = 63
om = np.arange(M + 1) * 2 * np.pi / 1.2
gv = np.concatenate([[-0.1, 1.1], om])
cases.append(("vff 127", lambda: eng.plan("matern12", "vff", gv, x1, "matern12", "vff", gv, x2, warm_start=True)))
sel = sys.argv[1:] 
for name, plan in cases:
    if sel and not any(s in name for s in sel): continue
    ms, e, rounds, info = loop(plan)
    print(f"{name:24s} {ms:8.3f} ms/step   mean rounds/step {rounds:8.1f}   elbo {e:.6f}   last info {info}")
