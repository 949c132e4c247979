"""Row-sharded ELBO step: one process per GPU, ONE all-reduce per step.

The grid rows (dimension 2, the slow axis of Y[n2][n1]) are split over the ranks.  Every rank runs
`vggp_elbo_partials` on its slab, the packed payload {G2, H2, C, C1, C2} (2 m2^2 + 3 m1 m2 doubles) is summed with a
single `torch.distributed.all_reduce` (RCCL on the GPU box, gloo in the CPU/1-GPU tests), and every rank finishes
redundantly with `vggp_elbo_finish`, so all ranks hold the identical value and gradient without a broadcast.
The reference has no multi-device path; the seam follows the sum structure of
Kuf Kuf^T = sum over grid rows in kronecker_structure.py:249-278.

Stream discipline: engine work and the collective are issued on ONE explicit side stream (torch's default stream
is the legacy null stream, which HIP cannot capture into a graph), so the ordering partials -> all_reduce -> finish
is by stream order, not by legacy-stream side effects.
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


class ShardedStep:
    def __init__(self, engine: Engine, group: Optional["dist.ProcessGroup"] = None):
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.stream = torch.cuda.Stream(device=engine.device)
        self.payload = None

    def sumsq_total(self, Y_local: torch.Tensor) -> float:
        """sum(y^2) over ALL ranks (data only: call once, outside the step loop)."""
        t = torch.tensor([self.engine.sumsq(Y_local)], dtype=torch.float64, device=self.engine.device)
        if self.world > 1:
            dist.all_reduce(t, group=self.group)
        return float(t.item())

    def step(self, Y_local: torch.Tensor, yy_total: float, theta: Sequence[float]) -> Tuple[float, np.ndarray, dict]:
        eng = self.engine
        if self.payload is None or self.payload.numel() != eng.payload_len:
            self.payload = torch.empty(eng.payload_len, dtype=torch.float64, device=eng.device)
        self.stream.wait_stream(torch.cuda.current_stream(eng.device))
        with torch.cuda.stream(self.stream):
            if self.world == 1:
                return eng.elbo_step(Y_local, yy_total, theta)
            eng.elbo_partials(Y_local, theta, self.payload)
            dist.all_reduce(self.payload, group=self.group)          # the single collective of the step
            return eng.elbo_finish(self.payload, yy_total, theta)    # synchronises the stream (returns host values)
