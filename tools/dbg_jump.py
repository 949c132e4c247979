"""RBF warm trajectory with a 30 % jump at step 15 (tests/test_gpu_elbo.py::_rbf_trajectory): error against the oracle around
the jump -- run under different VGGP_* switches to bisect."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dense as D, kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
n, m = 192, 64
X, y, x1, x2 = D.gen_grid(n, n)
g = np.linspace(0, 1, m)
e = Engine(0)
kind = sys.argv[2] if len(sys.argv) > 2 else "rbf"
e.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda")
yy = e.sumsq(Y)
f1, f2 = Kr.Factor("points", kind, g, x1), Kr.Factor("points", kind, g, x2)
th0 = np.array([0.2, 0.22, 1.0, 1.1, 0.01])
jump = float(sys.argv[1]) if len(sys.argv) > 1 else 1.3
kind = sys.argv[2] if len(sys.argv) > 2 else "rbf"
for k in range(20):
    th = th0 * (1 + 0.01 * k) * (jump if k >= 15 else 1.0)
    elbo, grad, info = e.elbo_step(Y, yy, th)
    if k >= 13:
        ref = Kr.elbo_step(y.reshape(n, n), f1, f2, th)
        print(k, "rel err elbo %.2e grad %.2e" % (abs(elbo - ref.elbo) / abs(ref.elbo), np.abs(grad - ref.grad).max() / np.abs(ref.grad).max()), info["rounds"], info["sweeps"], info["status"])
    if k == 15 and kind == 'rbf' and len(sys.argv) > 3:
        import ctypes as C
        e.lib.vggp_debug_read_gwork.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64]
        for dim in (0, 1):
            Gw = np.zeros((m, m)); lam = np.zeros(m)
            e.lib.vggp_debug_read_gwork(e._h, dim, 2, Gw.ctypes.data, 0, Gw.nbytes)
            e.lib.vggp_debug_read_gwork(e._h, dim, 4, lam.ctypes.data, 0, lam.nbytes)
            dg = np.abs(np.diag(Gw)); lmax = dg.max()
            r = int((lam > 1e-14 * lam.max()).sum()); rs = max(16, ((r + 4 + 7) // 8) * 8)
            for rr in (16, 24, 32):
                print(f"   dim {dim}: split {rr}: max |diag| of the complement block / lam_max = {dg[rr:].max() / lmax:.2e}; max |offdiag| in complement rows / lam_max = {np.abs(Gw[rr:] - np.diag(np.diag(Gw))[rr:]).max() / lmax:.2e}  (rank now {r})")
