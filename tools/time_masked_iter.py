"""ms per step of vggp_elbo_step_masked_iter on the 2048 x 2048 grid with 30 % missing (bench.py masked_iter)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from variational_gridded_gaussian_processes_amd import Engine, datagen as D
import torch
torch.cuda.set_device(0)
print(json.dumps(bench.masked_iter_bench(Engine(0), D), indent=1))
