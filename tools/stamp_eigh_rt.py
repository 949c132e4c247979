"""Real-time (s_memrealtime, 100 MHz) phase stamps of the scalar Jacobi producer in the warm regime
(library built with -DVG_EIG_RT as libvggp_stamp.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvggp_stamp.so")
from oracle import kron as Kr
from variational_gridded_gaussian_processes_amd import Engine
m = 128; kind = sys.argv[1] if len(sys.argv) > 1 else "rbf"
f = Kr.Factor("points", kind, np.linspace(0, 1, m), np.linspace(0, 1, 1024))
d = Kr.dim_prepare(f, 0.2, 1.0)
e = Engine(0)
lam, Qt, sw = e.eigh(torch.tensor(d.B @ d.B.T, device="cuda"))
Qt = Qt.cpu().numpy()
d1 = Kr.dim_prepare(f, 0.2 * 1.005, 1.0)
G = Qt @ (d1.B @ d1.B.T) @ Qt.T
Gd = torch.tensor(G, device="cuda")
for _ in range(3):
    lam, Q2, sw = e.eigh(Gd)
nb = m * m + 16
buf = (C.c_uint64 * nb)()
e.lib.vggp_debug_read_misc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
e.lib.vggp_debug_read_misc(e._h, buf, nb * 8)
t = np.array(list(buf))[m * m + 8:m * m + 14].astype(float) * 10.0 / 1e3      # us (stamps sit past the polish matrix E)
print("sweeps", sw, "phases (us): load+norm %.1f  dense phase %.1f  sparse-setup+sparse phase %.1f  sort+lam+DONE %.1f  | producer total %.1f  consumer-0 done at %.1f"
      % (t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[4] - t[0], t[5] - t[0]))
