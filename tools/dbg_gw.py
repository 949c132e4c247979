"""Structure of the matrix the main eigensolver launch of a warm RBF step starts from (Gw = E G E^T): how many off-diagonal
elements exceed the solver's threshold, and where (range x range, range x null, null x null).  (The Ritz-matrix part reads the
buffer Hs, which the step no longer fills since the Ritz launch forms H = T V1^T itself: run it on a commit before that change,
or ignore those lines.)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense as D
from variational_gridded_gaussian_processes_amd import Engine
import bench
n, m, kind = 1024, 128, "rbf"
X, y, x1, x2 = D.gen_grid(n, n); del X
e = Engine(0)
g = np.linspace(0, 1, m)
e.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda")
yy = e.sumsq(Y)
opt = bench.Adam(bench.raw_start(), lr=0.01)
e.lib.vggp_debug_read_gwork.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64]
for it in range(40):
    raw = opt.x
    el, gr, info = e.elbo_step(Y, yy, bench.theta_from_raw(raw.copy()))
    opt.step(-(gr / (1.0 + np.exp(-raw))))
print("info", info)
for dim in (0, 1):
    Gw = np.zeros((m, m)); lam = np.zeros(m)
    e.lib.vggp_debug_read_gwork(e._h, dim, 2, Gw.ctypes.data, 0, Gw.nbytes)
    e.lib.vggp_debug_read_gwork(e._h, dim, 4, lam.ctypes.data, 0, lam.nbytes)
    fro = np.linalg.norm(Gw)
    thr = 1e-13 * fro / m
    dg = np.diag(Gw).copy()
    r = int((lam > 1e-14 * lam.max()).sum())
    off = np.abs(Gw - np.diag(dg))
    print(f"dim {dim}: ||Gw||_F {fro:.3e} thr {thr:.2e} rank(lam) {r}; diag of Gw: max {dg.max():.3e}, [r-1] {dg[r-1]:.2e}, [r] {dg[r]:.2e}, min {dg.min():.2e}")
    for name, blk in (("range x range", off[:r, :r]), ("null x range", off[r:, :r]), ("null x null", off[r:, r:])):
        cnt = int((np.tril(blk, -1) > thr).sum()) if name != "null x range" else int((blk > thr).sum())
        print(f"   {name:14s} max |g| {blk.max():.2e}   elements above thr: {cnt}   above 100 thr: {int((blk > 100 * thr).sum())}")
    rs = max(16, ((r + 4 + 7) // 8) * 8)
    Hs = np.zeros((rs, rs))
    e.lib.vggp_debug_read_gwork(e._h, dim, 3, Hs.ctypes.data, 0, Hs.nbytes)
    hthr = 1e-13 * np.linalg.norm(Hs) / rs
    hoff = np.abs(Hs - np.diag(np.diag(Hs)))
    print(f"   Ritz matrix {rs} x {rs}: thr {hthr:.2e}, off-diagonal max {hoff.max():.2e}, elements above thr {int((np.tril(hoff, -1) > hthr).sum())} of {rs * (rs - 1) // 2}, "
          f"above 1e-6 ||H||: {int((np.tril(hoff, -1) > 1e-6 * np.linalg.norm(Hs)).sum())}; diag {np.array2string(np.diag(Hs)[:6], precision=3)} ... {np.diag(Hs)[-1]:.2e}")
    print(f"   split used by the step: sub_r = {rs}; Gw blocks with that split:")
    for name, blk in (("range x range", off[:rs, :rs]), ("null x range", off[rs:, :rs]), ("null x null", off[rs:, rs:])):
        cnt = int((np.tril(blk, -1) > thr).sum()) if name != "null x range" else int((blk > thr).sum())
        print(f"      {name:14s} max |g| {blk.max():.2e}   elements above thr: {cnt}   above 100 thr: {int((blk > 100 * thr).sum())}")
    rows = (np.tril(off, -1) > thr).sum(1) + (np.tril(off, -1) > thr).sum(0)
    print("   max above-threshold elements in one row/column:", int(rows.max()), " rows with any:", int((rows > 0).sum()))
