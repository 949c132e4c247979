import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvggp_stamp.so")
from oracle import kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
m = 128; kind = sys.argv[1] if len(sys.argv) > 1 else "matern32"
f = Kr.Factor("points", kind, np.linspace(0, 1, m), np.linspace(0, 1, 1024))
d = Kr.dim_prepare(f, 0.2, 1.0)
G = torch.tensor(d.B @ d.B.T, device="cuda")
e = Engine(0)
lam, Qt, sw = e.eigh(G)
buf = (C.c_uint64 * 128)()
e.lib.vggp_debug_read_misc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
print(e.lib.vggp_debug_read_misc(e._h, buf, 128 * 8))
a = np.array(list(buf)).reshape(16, 8).astype(float)
print("sweeps", sw)
for w in (0, 1, 4, 15):
    nin, nout = a[w, 6], a[w, 7]
    print("wave", w, "inner rounds", int(nin), "outer", int(nout), "per inner [P, bar1, U, bar2]:", (a[w, :4] / nin).round(0), "per outer [gather, apply]:", (a[w, 4:6] / nout).round(0), "total Mcycles", round(a[w, :6].sum() / 1e6, 2))
