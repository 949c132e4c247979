"""Row-sharded ELBO step: one process per GPU, ONE all-reduce per step -- owned by the library.

The grid rows (dimension 2, the slow axis of Y[n2][n1]) are split over the ranks.  On a multi-rank context
`vggp_elbo_step` itself runs  partials -> sum all-reduce of the packed payload {G2, H2, C, C1, C2}
(2 m2^2 + 3 m1 m2 doubles) -> finish  on one stream with one host synchronisation (csrc/comm.hip), and every rank
finishes redundantly, so all ranks hold the identical value and gradient without a broadcast.  This module only
bootstraps the context: it ships the RCCL unique id from rank 0 to the other ranks over an existing
`torch.distributed` group (any backend), or -- transport "gloo" -- installs a host callback that carries the payload
with `torch.distributed.all_reduce` on the CPU: the rehearsal of several ranks sharing ONE GPU (tests; RCCL refuses
duplicate devices).  The reference has no multi-device path; the seam follows the sum structure
of Kuf Kuf^T = sum over grid rows in kronecker_structure.py:249-278.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .engine import Engine


def shard_rows(n2: int, rank: int, world: int) -> slice:
    """Contiguous row slab of rank `rank` (the last ranks get the shorter slabs when world does not divide n2)."""
    per = -(-n2 // world)
    return slice(min(rank * per, n2), min((rank + 1) * per, n2))


def make_engine(device: Optional[int] = None, group: Optional["dist.ProcessGroup"] = None, transport: str = "rccl") -> Engine:
    """One Engine per rank of the (already initialised) torch.distributed group.  transport "rccl": the context owns an RCCL
    communicator (unique id broadcast from rank 0 through the group); "gloo": host-callback transport through the group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return Engine(device)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if transport == "gloo":
        def allreduce(view: np.ndarray):
            t = torch.from_numpy(view)                  # shares the pinned staging buffer of the library
            dist.all_reduce(t, group=group)
        return Engine(device, world, rank, None, allreduce)
    # Pre-flight: ncclCommInitRank is a rendezvous -- a rank that cannot even create its local context (wrong device, no
    # memory, library not built) would leave its peers waiting in it for ever.  So every rank first proves it can create (and
    # drop) a single-rank context on its device, and the ranks agree on the outcome over the bootstrap group before any of
    # them enters the rendezvous.
    ok, err = 1, None
    try:
        Engine(device).close()
    except Exception as e:          # noqa: BLE001 -- whatever it is, the peers must hear about it
        ok, err = 0, e
    flag = torch.tensor([ok])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) != 1:
        raise RuntimeError(f"rank {rank}: a rank of the job cannot create its local context; no communicator was created"
                           + (f" (this rank: {err})" if err is not None else ""))
    box = [Engine.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return Engine(device, world, rank, box[0])


class ShardedStep:
    """Thin wrapper kept for the callers of round 1: the step of a multi-rank Engine IS the sharded step."""

    def __init__(self, engine: Engine, group: Optional["dist.ProcessGroup"] = None):
        self.engine = engine
        self.group = group
        self.world = engine.n_ranks

    def sumsq_total(self, Y_local: torch.Tensor) -> float:
        """sum(y^2) over ALL ranks (data only: call once, outside the step loop) -- vggp_sumsq reduces over the context."""
        return self.engine.sumsq(Y_local)

    def step(self, Y_local: torch.Tensor, yy_total: float, theta: Sequence[float]) -> Tuple[float, np.ndarray, dict]:
        return self.engine.elbo_step(Y_local, yy_total, theta)


class ExternalCollectiveStep:
    """The step with the collective carried by the CALLER: vggp_elbo_partials -> torch.distributed.all_reduce ->
    vggp_elbo_finish on one explicit side stream (torch's default stream is the legacy null stream, which HIP cannot capture
    into a graph).  bench.py falls back to it when the library's own RCCL communicator cannot be created."""

    def __init__(self, engine: Engine, group: Optional["dist.ProcessGroup"] = None):
        self.engine, self.group = engine, group
        self.stream = torch.cuda.Stream(device=engine.device)
        self.payload = None

    def sumsq_total(self, Y_local: torch.Tensor) -> float:
        t = torch.tensor([self.engine.sumsq(Y_local)], dtype=torch.float64, device=self.engine.device)
        dist.all_reduce(t, group=self.group)
        return float(t.item())

    def step(self, Y_local, yy_total, theta):
        eng = self.engine
        if self.payload is None or self.payload.numel() != eng.payload_len:
            self.payload = torch.empty(eng.payload_len, dtype=torch.float64, device=eng.device)
        self.stream.wait_stream(torch.cuda.current_stream(eng.device))
        with torch.cuda.stream(self.stream):
            eng.elbo_partials(Y_local, theta, self.payload)
            dist.all_reduce(self.payload, group=self.group)
            return eng.elbo_finish(self.payload, yy_total, theta)
