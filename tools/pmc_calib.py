"""Known-size streaming read with this library's 8 B/lane loads: vggp_sumsq over 2^24 doubles (128 MiB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from variational_gridded_gaussian_processes_amd import Engine
e = Engine(0)
y = torch.ones(1 << 24, dtype=torch.float64, device="cuda")
for _ in range(3):
    print(e.sumsq(y))
