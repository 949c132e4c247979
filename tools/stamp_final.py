"""Phase stamps of the fused final reduction (library built with -DVG_FIN_STAMP as libvggp_stamp.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvggp_stamp.so")
from oracle import dense as D
from variational_gridded_gaussian_processes_amd import Engine
n, m = 1024, 128
X, y, x1, x2 = D.gen_grid(n, n)
g = np.linspace(0, 1, m)
eng = Engine(0)
eng.plan("rbf", "points", g, x1, "rbf", "points", g, x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda")
yy = eng.sumsq(Y)
th = np.array([0.2, 0.2, 1.0, 1.0, 0.0025])
for k in range(5):
    eng.elbo_step(Y, yy, th * (1 + 0.002 * k))
    buf = (C.c_double * 8)()
    eng.lib.vggp_debug_read_out.argtypes = [C.c_void_p, C.c_void_p]
    eng.lib.vggp_debug_read_out(eng._h, buf)
    a, b = buf[6], buf[7]
    print("last block cycles: loads %d  reduce %d  store+ticket %d  final %d" % (int(a), round((a - int(a)) * 1e6), int(b), round((b - int(b)) * 1e6)))
