#!/usr/bin/env python3
"""Headline benchmark: ELBO-step throughput (grid points / s) of the Kronecker-structured
collapsed-ELBO hot path on a 1024 x 1024 RBF grid (BASELINE.json `metric`), one process per GPU.

A "step" is one full ELBO evaluation -- value AND the 5-component hyper-parameter gradient --
through libvggp_hip.so, followed by a host-side Adam update of the 5 raw parameters, i.e. one
iteration of the notebooks' fit loop (5_gridded_kronecker_structure_models.ipynb cell 26).  The
hyper-parameters therefore change every step (the eigensolver's warm start is real work, not a
cached answer).  Observations are resident in HBM before the timed region.

`python3 bench.py --gpus N` with N > 1 and no launcher environment starts the N ranks ITSELF: before this process imports
torch or touches the GPU it spawns `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
bench.py <same arguments>` as a child, relays its one JSON line and exits with its code (the driver's own torchrun command
works unchanged: with WORLD_SIZE in the environment this process is a rank).  A --gpus / WORLD_SIZE mismatch is an error
(exit 2).  The collective is the library's RCCL communicator wherever the node has >= N devices; with fewer devices the ranks
share GPUs through the host-callback transport over gloo and the line says so ("rehearsal": true -- never a reported number).
`--strong` = BASELINE configs[3]'s shape: a fixed n1 x n2 grid (default 4096 x 4096) whose rows are split N ways.

N > 1 (weak scaling): every rank owns an (n2_local x n1) slab of an (n2_local*N) x n1 grid (rows of Y = dimension 2); the
step of the multi-rank context is  partials -> ONE all-reduce of the packed payload {G2,H2,C,C1,C2} (RCCL communicator
owned by the library, csrc/comm.hip) -> finish.  BASELINE configs[3] is `--gpus 4 --n1 4096 --n2-local 1024`;
configs[4] (masked grid) is `--masked --n1 2048 --n2-local 2048 --m 32` (divide --n2-local by the rank count).

Prints ONE JSON line on rank 0 (see the task contract): metric/value/... plus
  roofline          live HIP-event timing of the projection launch (the only pass over Y) vs the gfx950 fp64 MFMA roofline;
                    `traffic` = PMC bytes of the committed rocprofv3 passes, only while they match the kernel source at HEAD
  stages_us         per-launch-group breakdown of one step (HIP events on the launch stream, plain launches)
  cold_ms_per_step  the same loop with the eigensolver's warm start off
  m_d_sweep         ms per step for m_d in {32, 64, 128, 256}
  families          ms per step of Matern-3/2, 5/2 at m_d = 128 and Matern-3/2, 1/2 at m_d = 256 (refinement / Newton chain)
  fit_predict_loop  ms per iteration of a loop that calls q_v() after every step
  posterior_1M_points  vggp_posterior (mean + variance) at 2^20 scattered test points after a headline step
  svgp_train_z      ms per optimiser iteration of an SVGP whose inducing points are trained (step + Z-gradient + in-place move)
  scattered         vggp_elbo_step_scattered: ms per step for 100 000 points that form no grid (B0 cells, m_d = 32)
  masked_md128      masked 2048 x 2048 grid, 30 % missing, m_d = 128 (M = 16384): ms per step of the dense M-space solver
  masked_iter       the same grid through vggp_elbo_step_masked_iter (no M x M matrix): m_d = 128 and m_d = 256 (M = 65536)
  slab_1024x4096    the per-rank shape of BASELINE configs[3]: ms per step, and the projection kernel's MFMA fraction at that size
  kron_solve        BASELINE metric (ii): X = K1^{-1} Y K2^{-T} from Cholesky factors, GB/s and TFLOP/s
  factor_build      HBM-write rate of the factor kernel at m = n = 8192
  cpu_baseline      oracle/kron.py (the structured CPU twin, "port") timed on the host cores
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector == matrix peak
HBM_PEAK_GBS = 8000.0
PROJ = "gemm_project(S=[B2;V2]Y)"


def softplus(x):
    return np.logaddexp(0.0, x)


def inv_softplus(y):
    return y + np.log(-np.expm1(-y))


class Adam:
    """torch.optim.Adam on the 5 raw parameters (maximising the ELBO = minimising -ELBO)."""

    def __init__(self, x, lr=0.01, b1=0.9, b2=0.999, eps=1e-8):
        self.x, self.lr, self.b1, self.b2, self.eps = x.copy(), lr, b1, b2, eps
        self.m, self.v, self.t = np.zeros_like(x), np.zeros_like(x), 0

    def step(self, grad_loss):
        self.t += 1
        self.m = self.b1 * self.m + (1 - self.b1) * grad_loss
        self.v = self.b2 * self.v + (1 - self.b2) * grad_loss ** 2
        mh, vh = self.m / (1 - self.b1 ** self.t), self.v / (1 - self.b2 ** self.t)
        self.x = self.x - self.lr * mh / (np.sqrt(vh) + self.eps)
        return self.x


class FitLoop5:
    """The same Adam fit loop on the 5 raw parameters in plain Python floats (theta = softplus(raw), noise + 1e-4; gradient
    through the softplus): the arithmetic of Adam / theta_from_raw below without numpy's per-call overhead on 5-element arrays
    (11 us per step, measured -- 4 % of a 0.27 ms step)."""

    def __init__(self, raw, lr=0.01, b1=0.9, b2=0.999, eps=1e-8):
        self.x = [float(r) for r in raw]
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m, self.v, self.t = [0.0] * 5, [0.0] * 5, 0

    def theta(self):
        th = [(r if r > 0.0 else 0.0) + math.log1p(math.exp(-abs(r))) for r in self.x]
        th[4] += 1e-4
        return th

    def update(self, grad_theta):
        """One Adam step that MAXIMISES the ELBO, from the gradient w.r.t. theta."""
        self.t += 1
        b1, b2 = self.b1, self.b2
        c1, c2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        x, m, v = self.x, self.m, self.v
        if hasattr(grad_theta, "tolist"):
            grad_theta = grad_theta.tolist()          # (numpy scalars are slow operands)
        for i in range(5):
            g = -grad_theta[i] / (1.0 + math.exp(-x[i]))          # d softplus / d raw = sigmoid(raw); minimise -ELBO
            m[i] = b1 * m[i] + (1.0 - b1) * g
            v[i] = b2 * v[i] + (1.0 - b2) * g * g
            x[i] -= self.lr * (m[i] / c1) / (math.sqrt(v[i] / c2) + self.eps)


def theta_from_raw(raw):
    th = softplus(raw)
    th[4] += 1e-4
    return th


THETA0 = np.array([0.2, 0.2, 1.0, 1.0, 0.05 ** 2])


def raw_start():
    raw0 = THETA0.copy()
    raw0[4] -= 1e-4
    return inv_softplus(raw0)


def algorithmic_flops(n1, n2, m1, m2):
    """Per-rank flops of one step by launch group (dense-contraction counts)."""
    f = {}
    f["trsm_BV(L^-1[A|dA|dK])"] = (2 * m1 * m1 * n1 + 2 * m2 * m2 * n2) + (m1 ** 3 + m2 ** 3)     # substitution: m^2 per column
    f[PROJ] = 2 * 2 * m2 * n1 * n2                                                # S = [B2;V2] Y  (the only pass over Y)
    f["gemm_C(B1*S)"] = 2 * 3 * m1 * m2 * n1                                       # [C;C1;C2]
    f["gemm_gram(G,H,Mk)"] = 2 * (2 * m1 * m1 * n1 + 2 * m2 * m2 * n2) + 2 * (m1 ** 3 + m2 ** 3)
    f["gemm_rotate_right"] = 2 * 2 * (m1 ** 3 + m2 ** 3) + 2 * 3 * m1 * m2 * m2
    f["gemm_rotate_left"] = 2 * 2 * (m1 ** 3 + m2 ** 3) + 2 * 3 * m1 * m1 * m2
    f["gemm_betaGram"] = 2 * 2 * (m1 * m1 * m2 + m2 * m2 * m1)
    return f


def cpu_baseline(n1, n2, m, kind, theta, budget_s=15.0):
    """Structured CPU twin (oracle/kron.py) on the host cores: same workload, bounded sample."""
    from oracle import dense as D, kron as Kr
    X, y, x1, x2 = D.gen_grid(n1, n2)
    g = np.linspace(0, 1, m)
    f1, f2 = Kr.Factor("points", kind, g, x1), Kr.Factor("points", kind, g, x2)
    Y = y.reshape(n2, n1)
    t0 = time.perf_counter()
    Kr.elbo_step(Y, f1, f2, theta)            # warm-up (BLAS thread pools)
    one = time.perf_counter() - t0
    steps = int(max(2, min(50, budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        Kr.elbo_step(Y, f1, f2, theta)
    dt = (time.perf_counter() - t0) / steps
    try:
        import threadpoolctl
        cores = max([p["num_threads"] for p in threadpoolctl.threadpool_info()] or [os.cpu_count()])
    except Exception:
        cores = os.cpu_count()
    return {"value": n1 * n2 / dt, "unit": "grid-points/s", "cores": int(cores), "kind": "port",
            "dense_reference_literal": dense_literal_timing(kind, theta),
            "ms_per_step": dt * 1e3,
            "sample": f"{steps} ELBO steps (value+analytic gradient) of oracle/kron.py (numpy/LAPACK float64) on the "
                      f"same {n2}x{n1} {kind} grid, m_d={m}"}


def dense_literal_timing(kind, theta):
    """The reference's own dense algebra (oracle/dense.py: three N x N matrices, O(N^3) Cholesky, autograd backward) at the
    sizes it can run, with the N^3 extrapolation to the workload -- it cannot run at 256^2 and beyond (3 x 34 GB there)."""
    try:
        import torch
        from oracle import dense as D
        out = {}
        for n in (16, 32, 48):            # 16: warm-up of the BLAS / autograd machinery, not reported
            X, y, x1, x2 = D.gen_grid(n, n)
            g = torch.tensor(np.linspace(0, 1, 8))
            dm = D.DenseKron(X, y, "points", kind, g, g, raw=D.raw_from_constrained(list(theta)))
            t0 = time.perf_counter()
            dm.elbo_and_grad()
            if n > 16:
                out[f"{n}x{n}_s"] = time.perf_counter() - t0
        n_ref = 48
        out["extrapolated_1024x1024_s"] = out[f"{n_ref}x{n_ref}_s"] * ((1024 * 1024) / (n_ref * n_ref)) ** 3
        out["note"] = "value + autograd gradient, m_d = 8; N^3 extrapolation from 48x48"
        return out
    except Exception as e:          # never let the optional extra break the bench line
        return {"error": str(e)}


def source_sha():
    """Identity of the projection kernel's source: PMC traffic numbers are only reported while they were collected on this."""
    h = hashlib.sha256()
    for f in ("gemm.hip", "gemm_body.h"):
        with open(os.path.join(ROOT, "variational_gridded_gaussian_processes_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(n1, n2_loc, m):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE
    runs of this same command, profiles/r2_pmc_traffic.json written by tools/pmc_traffic.py).  None -- and the reason -- when
    the passes are absent, were taken on another kernel source than HEAD's, or on another launch shape."""
    path = os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "r2_pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        return None, "no committed PMC passes (profiles/r3_pmc_traffic.json)"
    if d.get("source_sha") != source_sha():
        return None, f"stale: passes taken on kernel source {d.get('source_sha')}, HEAD is {source_sha()}"
    if d.get("shape") != [n1, n2_loc, m]:
        return None, f"passes taken on shape {d.get('shape')}"
    return d, "profiles/" + os.path.basename(path)


def rocprof_kernel_us(kernel):
    """Average duration (us) of a kernel in the committed rocprofv3 --kernel-trace --stats summary of this same command
    (profiles/r3_step_kernel_stats.csv), or None."""
    import csv
    path = os.path.join(ROOT, "profiles", "r3_step_kernel_stats.csv")
    try:
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Name", "").startswith(kernel):
                    return float(row["AverageNs"]) / 1e3
    except Exception:
        return None
    return None


def _visible_devices():
    """Number of GPUs WITHOUT initialising HIP in this process (it may still have to spawn the ranks): the KFD topology lists
    one node per agent, GPUs are the ones with SIMDs; ROCR/HIP_VISIBLE_DEVICES narrow it."""
    n = 0
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except OSError:
                continue
    except OSError:
        n = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([t for t in v.split(",") if t.strip() != ""])) if n else len([t for t in v.split(",") if t.strip()])
    return n


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: this process has not imported torch nor touched the GPU; it becomes the supervisor of
    N fresh rank processes (torch.distributed.run spawns them; nothing is exec'ed from a GPU-initialised process)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["VGGP_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=1024, help="grid points per axis per rank (shorthand for --n1 N --n2-local N)")
    ap.add_argument("--n1", type=int, default=None, help="grid points along dimension 1 (the fast axis of Y; not sharded)")
    ap.add_argument("--n2-local", type=int, default=None, help="grid rows (dimension 2) per rank: BASELINE configs[3] is "
                                                              "--gpus 4 --n1 4096 --n2-local 1024")
    ap.add_argument("--m", type=int, default=128, help="inducing points (or B0 cells) per dimension")
    ap.add_argument("--kind", default="rbf")
    ap.add_argument("--masked", action="store_true", help="BASELINE configs[4]: Bernoulli(0.7) mask, B0 cells, masked step")
    ap.add_argument("--cold", action="store_true", help="disable the eigensolver warm start")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip cold_ms_per_step / m_d_sweep / kron_solve / factor_build")
    ap.add_argument("--strong", action="store_true", help="strong scaling on BASELINE configs[3]'s grid: a fixed n1 x n2 grid "
                                                          "(default 4096 x 4096; --n1 / --n2 change it) whose rows are split over the ranks")
    ap.add_argument("--n2", type=int, default=None, help="global number of grid rows of the --strong grid")
    ap.add_argument("--backend", default="auto", help="auto (RCCL communicator owned by the library when the node has >= N GPUs, "
                                                      "the measured path; otherwise the gloo rehearsal) | nccl | gloo (N > 1 ranks "
                                                      "sharing GPUs through the host-callback transport; never a reported number)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))            # (before torch is imported: this process never touches the GPU)
    if env_world is not None and int(env_world) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; they must agree",
              file=sys.stderr)
        sys.exit(2)

    # stdout carries exactly ONE line, the JSON: native libraries (gloo's "[Gloo] Rank 0 is connected ..." banner, RCCL
    # notices) print to file descriptor 1, so everything else is sent to stderr for the whole run
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from variational_gridded_gaussian_processes_amd import Engine
    from variational_gridded_gaussian_processes_amd.sharded import ExternalCollectiveStep, make_engine
    from variational_gridded_gaussian_processes_amd import datagen as D      # gen_2d layout + the notebooks' latent function

    collective = "none"
    ext = None
    rehearsal = False
    ndev = torch.cuda.device_count()            # (does not initialise HIP on this stack)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # torch.distributed only bootstraps (unique id, barriers, the max over ranks of the wall time): gloo on the CPU.  The
        # data path is the library's own RCCL communicator.
        dist.init_process_group("gloo")
        backend = args.backend
        if backend == "auto":
            backend = "nccl" if ndev >= world else "gloo"
            if backend == "gloo" and rank == 0:
                print(f"bench.py: {world} ranks on {ndev} GPU(s): RCCL refuses duplicate devices, so the ranks share GPUs through "
                      f"the host-callback transport over gloo -- a REHEARSAL of the multi-rank sequence, not a measurement",
                      file=sys.stderr)
        if backend == "nccl" and ndev < world:
            if rank == 0:
                print(f"bench.py: --backend nccl needs one GPU per rank ({world} ranks, {ndev} GPU(s))", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(2)
        dev = local_rank % max(ndev, 1)
        torch.cuda.set_device(dev)
        if backend == "gloo":
            eng = make_engine(dev, transport="gloo")
            collective = "host callback over gloo (rehearsal: ranks share GPUs)"
            rehearsal = True
        else:
            ok, eng = 1, None
            try:
                eng = make_engine(dev, transport="rccl")
            except Exception as e:           # all ranks must agree on the fallback
                print(f"[rank {rank}] library RCCL communicator failed: {e}", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                collective = "RCCL communicator owned by libvggp_hip.so (ncclAllReduce on the step's stream)"
            else:                            # torch.distributed carries the all-reduce between the two halves of the step
                if eng is not None:
                    eng.close()
                eng = Engine(dev)
                ext = ExternalCollectiveStep(eng, dist.new_group(backend="nccl"))
                collective = "torch.distributed NCCL all_reduce between vggp_elbo_partials / vggp_elbo_finish (fallback)"
    else:
        torch.cuda.set_device(0)
        eng = Engine(0)
    comm = eng.comm_info()                      # what the context itself says: communicator size + transport
    if ext is None and comm["n_ranks"] != world:
        print(f"bench.py: the context reports a communicator of {comm['n_ranks']} rank(s), the job has {world}", file=sys.stderr)
        sys.exit(3)

    m = args.m
    if args.strong:
        # BASELINE configs[3]: ONE fixed grid on [0,1]^2 whose rows (dimension 2) are split over the ranks
        n1 = args.n1 if args.n1 is not None else 4096
        n2_glob = args.n2 if args.n2 is not None else 4096
        if n2_glob % world:
            print(f"bench.py: --strong needs the {n2_glob} grid rows to divide over {world} ranks", file=sys.stderr)
            sys.exit(2)
        n2_loc = n2_glob // world
        x2_hi = 1.0
    else:
        n1 = args.n1 if args.n1 is not None else args.n
        n2_loc = args.n2_local if args.n2_local is not None else args.n
        n2_glob = n2_loc * world
        x2_hi = float(world)
    # weak: global grid x1 in [0,1] (n1 points), x2 in [0, world] (n2_loc*world points, same spacing per slab)
    X, y, x1, x2 = D.gen_grid(n1, n2_glob, lims2=(0.0, x2_hi), seed=0)
    del X
    Yg = y.reshape(n2_glob, n1)
    sl = slice(rank * n2_loc, (rank + 1) * n2_loc)
    basis = "b0" if args.masked else "points"
    kind = "matern12" if args.masked else args.kind
    g1 = np.linspace(0, 1, m + 1 if args.masked else m)
    g2 = np.linspace(0, x2_hi, m + 1 if args.masked else m)
    eng.plan(kind, basis, g1, x1, kind, basis, g2, x2[sl], n_total=n1 * n2_glob, warm_start=not args.cold)
    Y = torch.tensor(Yg[sl], device=eng.device)
    n_points = n1 * n2_glob
    W, n_obs, Wg = None, 0.0, 1.0
    if args.masked:
        Wg = (np.random.default_rng(1).uniform(size=(n2_glob, n1)) < 0.7).astype(np.float64)
        W = torch.tensor(Wg[sl], device=eng.device)
        Y = Y * W
        n_obs = float(Wg.sum())
        n_points = n_obs
    yy = float((Yg * Yg * Wg).sum())
    del y, Yg
    opt = FitLoop5(raw_start(), lr=0.01)

    def one_step():
        th = opt.theta()
        if args.masked:
            elbo, g, info = eng.elbo_step_masked(Y, W, n_obs, yy, th)
        elif ext is not None:
            elbo, g, info = ext.step(Y, yy, th)
        else:
            elbo, g, info = eng.elbo_step(Y, yy, th)       # multi-rank context: partials -> all-reduce -> finish inside
        opt.update(g)                                     # softplus chain rule + Adam
        return elbo, info

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        elbo, info = one_step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3

    # live per-launch-group timing (HIP events on the launch stream, plain launches), separate short loop
    stages_us = {}
    if not args.masked:
        eng.profile(True)
        nprof = max(5, min(50, args.steps))
        for _ in range(nprof):
            one_step()
        stage_ms, psteps = eng.profile_read()
        eng.profile(False)
        stages_us = {k: v / max(psteps, 1) * 1e3 for k, v in stage_ms.items() if v > 0.0}

    out = None
    if rank == 0:
        out = {
            "metric": "ELBO-step grid-points/sec (value + 5-component gradient), 1024x1024 RBF grid per GPU",
            "value": n_points / (ms_per_step * 1e-3), "unit": "grid-points/s" if not args.masked else "observed grid-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"2D Kronecker {kind} GP ({'B0 cells, 30 % of the grid masked out' if args.masked else 'points basis'}), "
                                    f"{n2_glob}x{n1} grid ({n2_loc}x{n1} per GPU), m_d={m}, Adam fit loop (lr 0.01), "
                                    f"eigensolver warm start {'off' if args.cold else 'on'}"),
                       "parallelism": (f"grid rows sharded over {world} rank(s) on {min(ndev, world)} GPU(s), one all-reduce of "
                                       f"{eng.payload_len} doubles per step: {collective}; the context reports a communicator "
                                       f"of {comm['n_ranks']} rank(s), transport {comm['transport']}") if world > 1 else "single GPU",
                       "comm": {"n_ranks": comm["n_ranks"], "transport": comm["transport"], "devices_visible": ndev,
                                "self_launched": os.environ.get("VGGP_BENCH_SELF_LAUNCHED") == "1"}},
            "rehearsal": rehearsal,
            "elbo_last": elbo,
            "jacobi": {"sweeps": info["sweeps"], "rounds": info["rounds"], "jitter": info["jitter"], "polished": info.get("polished")},
            "stages_us": stages_us,
            "stages_note": ("HIP events around plain launches in the library's profiling mode, where every launch group is its own "
                            "launch; in the timed region above the step is one graph replay in which the projection of Y and the "
                            "[C;C1;C2] products run as extra workgroups (riders) of the row-QR / Ritz launches"),
        }
        if not args.masked:
            flops = algorithmic_flops(n1, n2_loc, m, m)
            dom = max(stages_us, key=stages_us.get)
            proj_tflops = flops[PROJ] / (stages_us[PROJ] * 1e-6) / 1e12
            step_flops = sum(flops.values())
            alg_bytes = 8 * (n1 * n2_loc + 2 * m * n2_loc + 2 * m * n1)       # Y once, [B2;V2] once, the un-split output S
            tr, tr_src = pmc_traffic(n1, n2_loc, m)
            rp_us = rocprof_kernel_us(eng.project_kernel_name()) if (n1, n2_loc, m) == (1024, 1024, 128) else None
            out["roofline"] = {
                "kernel": f"{eng.project_kernel_name()} (launch group '{PROJ}': the only pass over Y)",
                "bound": "mfma", "achieved": proj_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": proj_tflops / FP64_PEAK_TFLOPS,
                "frac_events": proj_tflops / FP64_PEAK_TFLOPS,
                "frac_rocprof": (flops[PROJ] / (rp_us * 1e-6) / 1e12 / FP64_PEAK_TFLOPS) if rp_us else None,
                "avg_launch_us_rocprof": rp_us,
                "traffic": tr["bytes"] if tr else None, "traffic_source": tr_src, "traffic_detail": tr,
                "flops_per_launch": flops[PROJ], "avg_launch_us": stages_us[PROJ],
                "algorithmic_bytes_per_launch": alg_bytes,
                "traffic_over_algorithmic": (tr["bytes"] / alg_bytes) if tr else None,
                "dominant_stage_by_time": dom, "dominant_stage_us": stages_us[dom],
                "step_dense_flops": step_flops,
                "step_frac_of_fp64_peak": step_flops / (ms_per_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "step_hbm_floor_bytes": 8 * n1 * n2_loc + 8 * 2 * 2 * (m * n1 + m * n2_loc) + 8 * 4 * m * m,
            }
        if world == 1 and not args.no_extras and not args.masked:
            out["cold_ms_per_step"] = timed_loop(eng, Y, yy, args.kind, x1, x2, m, warm=False, steps=40, warmup=5)
            # (40 warm-up steps: the warm chains settle -- ranks, row counts, their captured graphs -- over the first few dozen steps)
            out["m_d_sweep"] = {str(md): timed_loop(eng, Y, yy, args.kind, x1, x2, md, warm=True, steps=100, warmup=40)
                                for md in (32, 64, 128, 256)}
            out["families"] = families_bench(eng, Y, yy, x1, x2)
            out["fit_predict_loop"] = fit_predict_bench(eng, Y, yy, args.kind, x1, x2, m)
            out["slab_1024x4096"] = slab_bench(eng, D, args.kind, m)
            out["posterior_1M_points"] = posterior_bench(eng, Y, yy, args.kind, x1, x2, m)
            out["svgp_train_z"] = trainz_bench(eng, Y, yy, x1, x2, m)
            out["scattered"] = scattered_bench(eng)
            out["masked_md128"] = masked_md128_bench(eng, D)
            out["masked_iter"] = masked_iter_bench(eng, D)
            out["kron_solve"] = kron_solve_bench(eng, 1024)
            out["factor_build"] = factor_build_bench(eng)
        if not args.no_cpu and world == 1 and not args.masked:
            out["cpu_baseline"] = cpu_baseline(n1, n2_loc, m, args.kind, THETA0)
        elif not args.no_cpu:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()


def timed_loop(eng, Y, yy, kind, x1, x2, m, warm, steps, warmup, basis="points", grid=None):
    """ms per step of the same Adam fit loop with another plan (inducing count / family / warm start) on the same data."""
    import torch
    g = np.linspace(0, 1, m) if grid is None else grid
    eng.plan(kind, basis, g, x1, kind, basis, g, x2, warm_start=warm)
    opt = FitLoop5(raw_start(), lr=0.01)          # (the headline's host loop: plain floats, no numpy per-call overhead)

    def one():
        e, gr, info = eng.elbo_step(Y, yy, opt.theta())
        opt.update(gr)

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def families_bench(eng, Y, yy, x1, x2):
    """Fit-loop ms per step of the other kernel families on the same 1024 x 1024 grid: full-rank spectra take
    the first-order refinement + polish (Matern-1/2, 3/2 at m_d = 128) or the Newton chain (Matern-5/2 at 128; everything at
    m_d = 256, which is beyond the LDS eigensolver)."""
    out = {}
    for kind, m in (("matern32", 128), ("matern52", 128), ("matern32", 256), ("matern12", 256)):
        out[f"{kind}_md{m}"] = timed_loop(eng, Y, yy, kind, x1, x2, m, warm=True, steps=60, warmup=20)
    # the reference's inter-domain families: B1 hats on a mesh padded by 8 knots beyond the data on either side
    # (gridded_kronecker_structure.py:699-724), 127 variational Fourier features on [-0.1, 1.1]; and Matern-5/2 beyond the LDS eigensolver
    m, pad = 128, 8
    d = 1.0 / (m - 1 - 2 * pad)
    out["b1_padded_md128"] = timed_loop(eng, Y, yy, "matern12", x1, x2, m, warm=True, steps=100, warmup=40, basis="b1",
                                        grid=np.linspace(-pad * d, 1 + pad * d, m))
    M = 63
    out["vff_127"] = timed_loop(eng, Y, yy, "matern12", x1, x2, 2 * M + 1, warm=True, steps=100, warmup=40, basis="vff",
                                grid=np.concatenate([[-0.1, 1.1], np.arange(M + 1) * 2 * np.pi / 1.2]))
    out["matern52_md256"] = timed_loop(eng, Y, yy, "matern52", x1, x2, 256, warm=True, steps=100, warmup=40)
    return out


def fit_predict_bench(eng, Y, yy, kind, x1, x2, m, iters=30):
    """A loop that alternates fit and predict: one ELBO step + q_v() per iteration.  Every read-out after a warm step re-runs the
    finish half with a cold eigensolve, so this is about step + 1 ms."""
    import torch
    g = np.linspace(0, 1, m)
    eng.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
    opt = Adam(raw_start(), lr=0.01)

    def one():
        raw = opt.x
        e, gr, info = eng.elbo_step(Y, yy, theta_from_raw(raw.copy()))
        opt.step(-(gr / (1.0 + np.exp(-raw))))
        return eng.qv()

    for _ in range(6):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    return {"ms_per_iteration": (time.perf_counter() - t0) / iters * 1e3, "what": "elbo_step + qv (mean and variance of q(v)) per iteration"}


def slab_bench(eng, D, kind, m, n1=4096, n2=1024):
    """BASELINE configs[3] per-rank shape (a 1024-row slab of a 4096-wide grid): ms per step of the fit loop, and the projection
    kernel -- the step's only N-proportional launch -- timed by itself (profiling mode) at a size where its fixed costs no longer
    dominate: 2 (2 m) n1 n2 flops against the fp64 MFMA peak."""
    import torch
    X, y, x1, x2 = D.gen_grid(n1, n2, seed=0)
    del X
    Y = torch.tensor(y.reshape(n2, n1), device=eng.device)
    yy = float((y * y).sum())
    ms = timed_loop(eng, Y, yy, kind, x1, x2, m, warm=True, steps=60, warmup=15)
    opt = Adam(raw_start(), lr=0.01)
    eng.profile(True)
    for _ in range(20):
        raw = opt.x
        e, gr, info = eng.elbo_step(Y, yy, theta_from_raw(raw.copy()))
        opt.step(-(gr / (1.0 + np.exp(-raw))))
    stage_ms, psteps = eng.profile_read()
    eng.profile(False)
    us = stage_ms[PROJ] / max(psteps, 1) * 1e3
    fl = 2.0 * (2 * m) * n1 * n2
    return {"ms_per_step": ms, "grid_points_per_s": n1 * n2 / (ms * 1e-3), "project_kernel": eng.project_kernel_name(),
            "project_us": us, "project_flops": fl, "project_TFLOP/s": fl / (us * 1e-6) / 1e12,
            "project_frac_of_fp64_peak": fl / (us * 1e-6) / 1e12 / FP64_PEAK_TFLOPS}


def posterior_bench(eng, Y, yy, kind, x1, x2, m, ns=1 << 20):
    """Prediction: vggp_posterior (mean + variance, kronecker_structure.py:199-230) at 2^20 scattered test points after a step
    of the headline workload; the one-off cold re-solve that follows a warm step is outside the timed region."""
    import torch
    g = np.linspace(0, 1, m)
    eng.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
    eng.elbo_step(Y, yy, THETA0)
    xs = torch.tensor(np.random.default_rng(0).uniform(0, 1, (ns, 2)), device=eng.device)
    eng.posterior(xs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        mean, var = eng.posterior(xs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    return {"points": ns, "ms": dt * 1e3, "points_per_s": ns / dt}


def trainz_bench(eng, Y, yy, x1, x2, m, kind="matern32", steps=60, warmup=15):
    """SVGP with trainable inducing points (kronecker_structure.py:303-304): ms per optimiser iteration = vggp_elbo_step +
    vggp_zgrad + vggp_set_inducing (Adam on the 5 hyper-parameters and on both coordinate vectors), same 1024 x 1024 grid."""
    import torch
    z = [np.linspace(0, 1, m), np.linspace(0, 1, m)]
    eng.plan(kind, "points", z[0], x1, kind, "points", z[1], x2, warm_start=True)
    opt = Adam(raw_start(), lr=0.01)
    oz = [Adam(z[0].copy(), lr=1e-4), Adam(z[1].copy(), lr=1e-4)]

    def one():
        raw = opt.x
        e, gr, info = eng.elbo_step(Y, yy, theta_from_raw(raw.copy()))
        g1, g2 = eng.zgrad(Y)
        opt.step(-(gr / (1.0 + np.exp(-raw))))
        for d, g in enumerate((g1, g2)):
            oz[d].step(-g.cpu().numpy())
            eng.set_inducing(d, oz[d].x)
        return e

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        e = one()
    torch.cuda.synchronize()
    return {"kind": kind, "ms_per_iteration": (time.perf_counter() - t0) / steps * 1e3, "elbo_last": e,
            "what": "elbo_step + zgrad + set_inducing x2, inducing points and hyper-parameters trained together"}


def masked_md128_bench(eng, D, n=2048, m=128, kind="matern12", steps=3):
    """BASELINE configs[4] grid (2048 x 2048, 30 % missing) at the headline's inducing count: m_d = 128, M = 16384 -- the dense
    M-space solver at the top of its range (21 GB of workspace, O(M^3) per step)."""
    import torch
    X, y, x1, x2 = D.gen_grid(n, n)
    del X
    mesh = np.linspace(0, 1, m + 1)
    W = torch.tensor((np.random.default_rng(1).uniform(size=(n, n)) < 0.7).astype(np.float64), device=eng.device)
    Ym = torch.tensor(y.reshape(n, n), device=eng.device) * W
    eng.plan(kind, "b0", mesh, x1, kind, "b0", mesh, x2)
    yy, nobs = eng.sumsq(Ym), float(W.sum().item())
    eng.elbo_step_masked(Ym, W, nobs, yy, THETA0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        e, gr, info = eng.elbo_step_masked(Ym, W, nobs, yy, [t * (1 + 0.01 * (k + 1)) for t in THETA0])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"n": n, "m_d": m, "M": m * m, "ms_per_step": dt * 1e3, "observed_points_per_s": nobs / dt, "elbo_last": e}


def masked_iter_bench(eng, D, n=2048, kind="matern12", steps=3, n_probes=16):
    """The masked step without M x M matrices (vggp_elbo_step_masked_iter: PCG with matricised Kronecker MVMs, Kronecker-eigenbasis
    preconditioner, Lanczos quadrature, control-variate traces; fixed probes) on BASELINE configs[4]'s grid (2048 x 2048, 30 %
    missing): m_d = 128 (M = 16384, where the dense solver needs 227 ms) with its deviation from the dense step, and m_d = 256
    (M = 65536: the dense solver refuses above 16384)."""
    import torch
    X, y, x1, x2 = D.gen_grid(n, n)
    del X
    W = torch.tensor((np.random.default_rng(1).uniform(size=(n, n)) < 0.7).astype(np.float64), device=eng.device)
    Ym = torch.tensor(y.reshape(n, n), device=eng.device) * W
    nobs = float(W.sum().item())
    out = {"n": n, "n_probes": n_probes}
    for m in (128, 256):
        mesh = np.linspace(0, 1, m + 1)
        eng.plan(kind, "b0", mesh, x1, kind, "b0", mesh, x2)
        yy = eng.sumsq(Ym)
        e, gr, info = eng.elbo_step_masked_iter(Ym, W, nobs, yy, THETA0, n_probes=n_probes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            e, gr, info = eng.elbo_step_masked_iter(Ym, W, nobs, yy, [t * (1 + 0.01 * (k + 1)) for t in THETA0], n_probes=n_probes)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        r = {"M": m * m, "ms_per_step": dt * 1e3, "observed_points_per_s": nobs / dt, "pcg_iterations": info["rounds"][0], "elbo_last": e}
        if m == 128:
            th = [t * (1 + 0.01 * steps) for t in THETA0]
            ed, gd, _ = eng.elbo_step_masked(Ym, W, nobs, yy, th)
            r["elbo_rel_dev_vs_dense"] = abs(e - ed) / abs(ed)
            r["grad_dev_vs_dense_rel_to_max"] = float(np.abs(gr - gd).max() / np.abs(gd).max())
        out[f"m_d_{m}"] = r
    return out


def scattered_bench(eng, N=100000, m=32, kind="matern12", steps=10, warmup=3):
    """Scattered observations (along-track points, vggp_elbo_step_scattered): ms per step for N points that form no grid,
    B0 cells with m_d = 32 (M = 1024) -- O(M^2 N) assembly + O(M^3) dense factorisation."""
    import torch
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N, 2))
    y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + 0.1 * rng.standard_normal(N)
    g = np.linspace(0, 1, m + 1)
    eng.plan(kind, "b0", g, X[:, 0].copy(), kind, "b0", g, X[:, 1].copy(), scattered=True)
    yd = torch.tensor(y, device=eng.device)
    yy = float(y @ y)
    opt = Adam(raw_start(), lr=0.01)

    def one():
        raw = opt.x
        e, gr, info = eng.elbo_step_scattered(yd, yy, theta_from_raw(raw.copy()))
        opt.step(-(gr / (1.0 + np.exp(-raw))))
        return e

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        e = one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    M = m * m
    out = {"points": N, "m_d": m, "M": M, "ms_per_step": ms, "points_per_s": N / (ms * 1e-3), "elbo_last": e,
           "assembly_flops": 3 * 2.0 * M * M * N, "assembly_TFLOP/s_if_all_time": 3 * 2.0 * M * M * N / (ms * 1e-3) / 1e12}
    # SVGP on the same points with trainable inducing coordinates: step + vggp_zgrad_scattered + in-place move
    z = [np.linspace(0, 1, m), np.linspace(0, 1, m)]
    eng.plan("matern32", "points", z[0], X[:, 0].copy(), "matern32", "points", z[1], X[:, 1].copy(), scattered=True)
    oz = [Adam(z[0].copy(), lr=1e-4), Adam(z[1].copy(), lr=1e-4)]

    def one_z():
        raw = opt.x
        e, gr, info = eng.elbo_step_scattered(yd, yy, theta_from_raw(raw.copy()))
        g1, g2 = eng.zgrad_scattered(yd)
        opt.step(-(gr / (1.0 + np.exp(-raw))))
        for d, g in enumerate((g1, g2)):
            oz[d].step(-g.cpu().numpy())
            eng.set_inducing(d, oz[d].x)

    for _ in range(2):
        one_z()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        one_z()
    torch.cuda.synchronize()
    out["svgp_train_z_ms_per_iteration"] = (time.perf_counter() - t0) / 5 * 1e3
    return out


def factor_build_bench(eng, n=8192, reps=5):
    """North-star evidence for the kernel build: HBM-write rate of vg_factor_kernel at m = n = 8192 (outside the cache regime:
    2.1 GB of outputs), pairwise RBF factors and B0 cell integrals.  Algorithmic bytes = 16 B per output element (value +
    d/d ell) for the m x n and the m x m matrix.  (The rocprofv3 kernel-trace figure of the same launch is in profiles/.)"""
    import torch
    from variational_gridded_gaussian_processes_amd._lib import BASIS, KIND, check
    res = {}
    o = dict(dtype=torch.float64, device=eng.device)
    x = torch.linspace(0, 1, n, **o)
    A, dA, K, dK = torch.empty(n, n, **o), torch.empty(n, n, **o), torch.empty(n, n, **o), torch.empty(n, n, **o)
    for kind, basis in (("rbf", "points"), ("matern12", "b0")):
        g = torch.linspace(0, 1, n + 1 if basis == "b0" else n, **o)

        def call():
            check(eng.lib.vggp_factor_build(eng._h, KIND[kind], BASIS[basis], x.data_ptr(), n, g.data_ptr(), n, 0.2, 0,
                                            A.data_ptr(), dA.data_ptr(), K.data_ptr(), dK.data_ptr(), 0))
        call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        byt = 2 * 16.0 * n * n
        res[f"{kind}_{basis}"] = {"ms": dt * 1e3, "GB/s_written": byt / dt / 1e9, "frac_of_hbm_peak": byt / dt / 1e9 / HBM_PEAK_GBS}
    res["n"] = n
    res["algorithmic_bytes"] = 2 * 16 * n * n
    return res


def kron_solve_bench(eng, n, reps=20):
    """BASELINE metric (ii): X = K1^{-1} Y K2^{-T} FROM THE CHOLESKY FACTORS (nothing pre-inverted: the whole solve --
    diagonal-block inverses, block-doubling inverse, four triangular-aware GEMMs -- is inside the timed region).  Algorithmic
    bytes and flops as SURVEY.md section 8d defines them: 16 n1 n2 + 4 (n1^2 + n2^2) bytes, 2 n1^2 n2 + 2 n2^2 n1 flops."""
    import torch
    z = torch.linspace(0, 1, n, dtype=torch.float64, device=eng.device)
    K1 = eng.factor_build("matern12", "points", z, z, 0.2)[2]
    K2 = eng.factor_build("matern32", "points", z, z, 0.05)[2]
    L1, _, _ = eng.cholesky_inverse(K1)
    L2, _, _ = eng.cholesky_inverse(K2)
    Yk = torch.randn(n, n, dtype=torch.float64, device=eng.device)
    for _ in range(3):
        eng.kron_solve(L1, L2, Yk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        Xk = eng.kron_solve(L1, L2, Yk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    resid = float((K1 @ Xk @ K2.T - Yk).abs().max())
    alg_bytes = 16 * n * n + 4 * (n * n + n * n)
    flops = 4 * n ** 3
    return {"n": n, "ms": dt * 1e3, "GB/s": alg_bytes / dt / 1e9, "algorithmic_bytes": alg_bytes,
            "TFLOP/s_algorithmic(4n^3)": flops / dt / 1e12, "frac_of_fp64_peak": flops / dt / 1e12 / FP64_PEAK_TFLOPS,
            "from": "Cholesky factors (timed region includes everything)", "max_abs_residual": resid}


if __name__ == "__main__":
    main()
