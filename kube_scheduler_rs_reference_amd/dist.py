"""Multi-GPU: pod rows shard across ranks, node snapshot is replicated, bindings are all-gathered.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  Pods are independent given the node snapshot, so the only
exchange step is one all-gather of the int32 bindings (4 B per pod); masks stay on the GPU that
produced them.

    rank r owns pod rows [r * shard, min(P, (r + 1) * shard)),  shard = ceil(P / world)

`ShardedScheduler` is transport + bookkeeping only; the evaluation itself is the `local_eval`
callable (in the product: `Evaluator.eval_device` on this rank's GPU).

`PipelinedScheduler` is the throughput form of the same thing: consecutive batches ("steps") are
software-pipelined with `depth` buffer slots over the two HIP streams of a `ksched_pipe`
(include/ksched.h) -- the mask kernel of step i + 1 runs while the pick kernel and the all-gather
of step i are still in flight (they do not depend on each other; per slot, events and stream order
give mask -> pick -> all-gather -> next use of the slot).  Every step's outputs are complete and
identical to the sequential form.
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


class PendingBindings:
    """Result of PipelinedScheduler.step: `.wait()` orders the current stream (CPU: the host) after the step's pick
    and all-gather and returns the global bindings [P] (a view of the slot's buffer, valid until the slot is reused,
    i.e. for `depth - 1` further steps)."""

    def __init__(self, sched: "PipelinedScheduler", slot: int):
        self._s, self._slot = sched, slot

    def wait(self) -> torch.Tensor:
        s, k = self._s, self._slot
        if s._gather:
            if s._work[k] is not None:
                s._work[k].wait()
            return s._gathered[k][: s.P]
        if s.pipe is not None and s._used[k]:
            s.pipe.wait(k)
        return s._local[k][: s.P]


class PipelinedScheduler:
    """Row-sharded evaluation, pick and all-gather of consecutive batches, `depth` steps in flight.

    step(run):  run(slot, binding_out) enqueues this rank's evaluation + pick of one batch into buffer slot `slot`,
    filling binding_out[: n_local] (int32, -1 = no node).
      * on a GPU, `pipe` is the evaluator's two-stream pipeline (Evaluator.pipe(depth)) and `run` calls
        `pipe.submit(slot, ...)`: the mask kernel goes to the pipe's mask stream, the pick to its pick stream, and
        the all-gather is enqueued (asynchronously) behind the pick on that same stream;
      * without a pipe (CPU, gloo tests) `run` executes inline and only the all-gather is asynchronous."""

    def __init__(self, P: int, device: torch.device, depth: int = 2, group: Optional[dist.ProcessGroup] = None, pipe=None,
                 gather_always: bool = False):
        if depth < 1:
            raise ValueError("depth >= 1")
        if pipe is not None and pipe.depth != depth:
            raise ValueError("pipe.depth != depth")
        self.P, self.device, self.depth, self.group, self.pipe = P, device, depth, group, pipe
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi, self.shard = shard_bounds(P, self.world, self.rank)
        # gather_always: run the all-gather even in a one-rank group (exercises the RCCL path on a single GPU; tests)
        self._gather = self.world > 1 or (gather_always and dist.is_initialized())
        self._local = [torch.full((self.shard,), -1, dtype=torch.int32, device=device) for _ in range(depth)]
        self._gathered = [torch.full((self.shard * self.world,), -1, dtype=torch.int32, device=device) if self._gather else None
                          for _ in range(depth)]
        self._work = [None] * depth
        self._used = [False] * depth
        self._pick_stream = pipe.stream(1) if pipe is not None else None
        self._i = 0

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def step(self, run: Callable[[int, torch.Tensor], None]) -> PendingBindings:
        k = self._i % self.depth
        self._i += 1
        out = self._local[k][: self.n_local]
        if self._pick_stream is not None and self._gather:
            with torch.cuda.stream(self._pick_stream):
                if self._work[k] is not None:  # the all-gather that read this slot's bindings `depth` steps ago:
                    self._work[k].wait()       # the pick stream (hence this slot's next pick) is ordered after it
                if self.n_local > 0:
                    run(k, out)
                    self._used[k] = True
                self._work[k] = dist.all_gather_into_tensor(self._gathered[k], self._local[k], group=self.group, async_op=True)
        else:
            if self._work[k] is not None:
                self._work[k].wait()
            if self.n_local > 0:
                run(k, out)
                self._used[k] = True
            if self._gather:
                self._work[k] = dist.all_gather_into_tensor(self._gathered[k], self._local[k], group=self.group, async_op=True)
        return PendingBindings(self, k)

    def drain(self) -> None:
        """Order the current stream (CPU: the host) after everything in flight."""
        for k in range(self.depth):
            PendingBindings(self, k).wait()
