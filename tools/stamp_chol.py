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
buf = (C.c_uint64 * 64)()
e.lib.vggp_debug_read_misc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
e.lib.vggp_debug_read_misc(e._h, buf, 64 * 8)
a = np.array(list(buf))[16:32].reshape(4, 4)
print("per step cycles [publish, barrier, update], total loop cycles:")
for w in range(4): print(w, (a[w, :3] / m).round(1), a[w, 3], "per step total", round(a[w, 3] / m, 1))
