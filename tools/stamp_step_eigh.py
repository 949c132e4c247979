"""Real-time phase stamps (s_memrealtime, 100 MHz) of the two eigensolver launches INSIDE the warm fit-loop step
(library built with -DVG_EIG_RT as libvggp_stamp.so):  hipcc ... -DVG_EIG_RT -o libvggp_stamp.so"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from variational_gridded_gaussian_processes_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvggp_stamp.so")
from oracle import dense as D
from variational_gridded_gaussian_processes_amd import Engine
import bench
n, m, kind = 1024, 128, (sys.argv[1] if len(sys.argv) > 1 else "rbf")
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
for which, name in ((1, "ritz (r x r)"), (0, "main")):
    for dim in (0, 1):
        buf = (C.c_uint64 * 8)()
        mm = 20 if which == 1 else m      # the Ritz problem is sub_r x sub_r (20 on this workload)
        rc = e.lib.vggp_debug_read_gwork(e._h, dim, which, buf, mm * mm + 8, 64)
        t = np.array(list(buf)).astype(float) * 10.0 / 1e3
        print(f"{name} dim {dim}: rc {rc}  load+norm {t[1]-t[0]:.1f}  dense {t[2]-t[1]:.1f}  sparse {t[3]-t[2]:.1f}  sort+lam+DONE {t[4]-t[3]:.1f} | producer {t[4]-t[0]:.1f}  replay-0 done at {t[5]-t[0]:.1f} us")

for dim in (0, 1):
    cnt = (C.c_int32 * 4)()
    e.lib.vggp_debug_read_gwork(e._h, dim, 5, cnt, 0, 16)
    print(f"ritz dim {dim}: logged rounds {cnt[0]}, sweeps {cnt[1] & 0xff}, rank {cnt[1] >> 8}, status {cnt[2]}")
