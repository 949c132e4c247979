"""Phase stamps of vg_thin_tail_kernel (library built with -DVGGP_DIAG as libvggp_diag.so):
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DVGGP_DIAG -o libvggp_diag.so csrc/*.hip -ldl"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from variational_gridded_gaussian_processes_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvggp_diag.so")
from variational_gridded_gaussian_processes_amd import Engine, datagen as D
e = Engine(0)
n, m = 1024, 128
X, y, x1, x2 = D.gen_grid(n, n)
g = np.linspace(0, 1, m)
e.plan("rbf", "points", g, x1, "rbf", "points", g, x2, warm_start=True)
Y = torch.tensor(y.reshape(n, n), device="cuda")
yy = e.sumsq(Y)
th = np.array([0.2, 0.2, 1.0, 1.0, 0.0025])
for k in range(8):
    el, gr, info = e.elbo_step(Y, yy, th * (1 + 0.002 * k))
buf = (C.c_uint64 * 16)()
e.lib.vggp_debug_read_gwork.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64]
e.lib.vggp_debug_read_gwork(e._h, 0, 6, buf, 3 * 32 * 32, 16 * 8)
t = np.array(list(buf), dtype=np.float64)
names = ["loads issued", "trace sum", "rot E,F (2 mm)", "rot P (2 mm)", "D-stage", "beta Grams (1 mm)", "contractions", "block sum 24", "final (lane 0)", "host burst"]
print("elbo", el, "info", info)
for i, nm in enumerate(names):
    print(f"{nm:20s} {(t[i + 1] - t[i]) / 100.0:8.2f} us")       # s_memrealtime ticks at 100 MHz
print("total", (t[10] - t[0]) / 100.0)
