"""Multi-GPU: pod rows shard across ranks, node snapshot is replicated, bindings are all-gathered.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  Pods are independent given the node snapshot, so the only
exchange step is one all-gather of the int32 bindings (4 B per pod); masks stay on the GPU that
produced them.

    rank r owns pod rows [r * shard, min(P, (r + 1) * shard)),  shard = ceil(P / world)

`ShardedScheduler` is transport + bookkeeping only; the evaluation itself is the `local_eval`
callable (in the product: `Evaluator.eval_device` on this rank's GPU).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(P: int, world: int, rank: int) -> Tuple[int, int, int]:
    """-> (lo, hi, shard) for contiguous row sharding; the last ranks may own fewer (or zero) rows."""
    shard = (P + world - 1) // world if world > 0 else P
    lo = min(P, rank * shard)
    hi = min(P, lo + shard)
    return lo, hi, shard


@dataclass
class ShardedScheduler:
    """Row-sharded evaluation + all-gather of the (pod -> node) bindings."""
    P: int                                   # global number of pods
    device: torch.device
    group: Optional[dist.ProcessGroup] = None

    def __post_init__(self):
        self.world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        self.lo, self.hi, self.shard = shard_bounds(self.P, self.world, self.rank)
        # equal-size slots so one all_gather_into_tensor moves everything; tail slots hold -1
        self.local = torch.full((self.shard,), -1, dtype=torch.int32, device=self.device)
        self.gathered = torch.full((self.shard * self.world,), -1, dtype=torch.int32, device=self.device)

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def step(self, local_eval: Callable[[torch.Tensor], None]) -> torch.Tensor:
        """local_eval(binding_out) must fill binding_out[: n_local] (int32, -1 = no node) for this
        rank's rows, enqueued on the current stream.  Returns the global bindings [P] (a view)."""
        if self.n_local > 0:
            local_eval(self.local[: self.n_local])
        if self.world > 1:
            dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
            return self.gathered[: self.P]
        return self.local[: self.P]
