"""CPU test of the N > 1 PROTOCOL ONLY (gloo, world_size 2): every rank computes the packed payload of its row shard, ONE
all-reduce sums it, every rank finishes redundantly -- and equals the single-rank result.  The per-rank arithmetic here is
the ORACLE's (there is no GPU in the CPU suite, and the product has no CPU path): this proves that the payload
{G2, H2, C, C1, C2} + y^T y is sufficient and that its sum over row shards is exact, nothing about the HIP partials / finish.
The product's own multi-rank step (vggp_elbo_step on an n_ranks > 1 context: partials graph -> the context's all-reduce ->
finish graph) is tested on the GPU in tests/test_gpu_dist.py: 2 / 3 / 4 ranks sharing one GPU through the host-callback
transport over gloo, the failure path of a rank, an RCCL communicator of size one, and RCCL between two devices wherever two
are visible."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pack(p):
    return np.concatenate([p["G2"].ravel(), p["H2"].ravel(), p["C"].ravel(), p["C1"].ravel(), p["C2"].ravel()])


def _unpack(v, m1, m2):
    o, out = 0, {}
    for k, shp in (("G2", (m2, m2)), ("H2", (m2, m2)), ("C", (m1, m2)), ("C1", (m1, m2)), ("C2", (m1, m2))):
        n = shp[0] * shp[1]
        out[k] = v[o:o + n].reshape(shp)
        o += n
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from oracle import dense as D, kron as Kr
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n1, n2, m = 24, 20, 6
    X, y, x1, x2 = D.gen_grid(n1, n2)
    Y = y.reshape(n2, n1)
    theta = [0.2, 0.3, 1.0, 0.8, 0.01]
    g = np.linspace(0, 1, m)
    f1, f2 = Kr.Factor("points", "matern32", g, x1), Kr.Factor("points", "matern32", g, x2)
    rows = slice(rank * n2 // world, (rank + 1) * n2 // world)
    d1 = Kr.dim_prepare(f1, theta[0], theta[2])
    d2 = Kr.dim_prepare(f2, theta[1], theta[3], cols=rows)
    pay = Kr.local_partials(Y[rows], d1, d2)
    buf = torch.tensor(np.concatenate([_pack(pay), [pay["yy"]]]))
    dist.all_reduce(buf)                                  # the single collective of the step
    red = _unpack(buf.numpy()[:-1], m, m)
    red["yy"] = float(buf[-1])
    st = Kr.finish(theta, d1, Kr.dim_prepare(f2, theta[1], theta[3]), red, N=n1 * n2)
    q.put((rank, st.elbo, st.grad))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_rank():
    from oracle import dense as D, kron as Kr
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    X, y, x1, x2 = D.gen_grid(24, 20)
    g = np.linspace(0, 1, 6)
    ref = Kr.elbo_step(y.reshape(20, 24), Kr.Factor("points", "matern32", g, x1), Kr.Factor("points", "matern32", g, x2),
                       [0.2, 0.3, 1.0, 0.8, 0.01])
    for rank, elbo, grad in res:
        assert abs(elbo - ref.elbo) <= 1e-12 * abs(ref.elbo)
        assert np.abs(grad - ref.grad).max() <= 1e-10 * np.abs(ref.grad).max()
    assert res[0][1] == res[1][1]                          # every rank holds the identical value
