"""Per-phase cycle stamps of the scalar Jacobi producer (library built with -DVG_EIG_STAMP as libvggp_stamp.so).
usage: stamp_eigh.py <kind> [warm]     warm: G' = Qprev^T G(1.01 ell) Qprev (the fit-loop regime)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvggp_stamp.so")
from oracle import kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
m = 128; kind = sys.argv[1] if len(sys.argv) > 1 else "matern32"
warm = len(sys.argv) > 2
f = Kr.Factor("points", kind, np.linspace(0, 1, m), np.linspace(0, 1, 1024))
d = Kr.dim_prepare(f, 0.2, 1.0)
G = d.B @ d.B.T
e = Engine(0)
if warm:
    lam, Qt, sw = e.eigh(torch.tensor(G, device="cuda"))
    Qt = Qt.cpu().numpy()
    d1 = Kr.dim_prepare(f, 0.2 * 1.01, 1.0)
    G = Qt @ (d1.B @ d1.B.T) @ Qt.T
    print("warm: max offdiag / fro", np.abs(G - np.diag(np.diag(G))).max() / np.linalg.norm(G))
torch.cuda.synchronize()
import time
Gd = torch.tensor(G, device="cuda")
lam, Qt, sw = e.eigh(Gd)
torch.cuda.synchronize(); t0 = time.perf_counter()
lam, Qt, sw = e.eigh(Gd)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
buf = (C.c_uint64 * 128)()
e.lib.vggp_debug_read_misc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
e.lib.vggp_debug_read_misc(e._h, buf, 128 * 8)
b = np.array(list(buf)).astype(float)
a = b[:64].reshape(16, 4)
f = b[64:].reshape(16, 4)
nr = sw * (m - 1)
print(f"sweeps {sw} rounds {nr} wall {dt*1e6:.0f} us (incl. launch+sync)")
for w in (0, 1, 8, 15):
    print("wave", w, "totals kcycles [P, bar1, U, bar2]:", (a[w] / 1e3).round(1), "sum", round(a[w].sum() / 1e3, 1), "per round", (a[w] / nr).round(0))
for w in (0, 1, 8, 15):
    n = max(f[w, 3], 1)
    print("fast wave", w, "rounds", int(f[w, 3]), "per round [update, angle, barrier]:", (f[w, :3] / n).round(0), "total kcycles", round(f[w, :3].sum() / 1e3, 1))
