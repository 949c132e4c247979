"""Phase stamps of the MFMA-blocked Cholesky (library built with -DVG_CHOL_STAMP as libvggp_stamp.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvggp_stamp.so")
from oracle import kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
e = Engine(0)
m = 128
z = np.linspace(0, 1, m); K, _ = Kr.points_factor("matern32", z, z, 0.2)
L, Li, jit = e.cholesky_inverse(torch.tensor(K, device="cuda"))
L, Li, jit = e.cholesky_inverse(torch.tensor(K, device="cuda"))
buf = (C.c_uint64 * 128)()
e.lib.vggp_debug_read_misc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
e.lib.vggp_debug_read_misc(e._h, buf, 128 * 8)
a = np.array(list(buf))[16:16 + 96].reshape(8, 12)
print("cycles since start: [load, panels, select, Lout, inverse, end] ; sums over panels [update, factor/wait, trsm+barrier]")
for w in (0, 1, 7): print("wave", w, a[w, 1:7], a[w, 7:10])
