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
buf = (C.c_uint64 * 64)()
e.lib.vggp_debug_read_misc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
print(e.lib.vggp_debug_read_misc(e._h, buf, 64 * 8))
a = np.array(list(buf)).reshape(16, 4)
rounds = sw * 127
print("sweeps", sw, "rounds(total)", rounds)
print("per-round s_memtime ticks, wave: [P, barrier1, U, barrier2]")
for w in (0, 1, 7, 15):
    print(w, (a[w] / rounds).round(1), "sum", round(a[w].sum() / rounds, 1))
