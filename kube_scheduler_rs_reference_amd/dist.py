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
    and all-gather and returns the global bindings [P] (valid until the step's buffer slot is reused, i.e. for
    `(depth - 1) * gather_every` further steps)."""

    def __init__(self, sched: "PipelinedScheduler", slot: int, sub: int):
        self._s, self._slot, self._sub = sched, slot, sub

    def wait(self) -> torch.Tensor:
        s, k, g = self._s, self._slot, self._sub
        if s._gather:
            if s._fill[k] > 0 and s._cur == k:  # the step's group has not been gathered yet: do it now, short
                s._flush(k)
            if s._work[k] is not None:
                s._work[k].wait()
            # gathered[k] is [world][gather_every][shard]: this step's rows of every rank
            return s._gathered[k].view(s.world, s.gather_every, s.shard)[:, g].reshape(-1)[: s.P]
        if s.pipe is not None and s._used[k]:
            s.pipe.wait(k * s.gather_every + g)
        return s._local[k].view(s.gather_every, s.shard)[g][: s.P]


class PipelinedScheduler:
    """Row-sharded evaluation, pick and all-gather of consecutive batches, `depth` buffer slots in flight.

    step(run):  run(slot, binding_out) enqueues this rank's evaluation + pick of one batch, filling
    binding_out[: n_local] (int32, -1 = no node).
      * on a GPU with `pipe` (Evaluator.pipe(depth * gather_every)) `run(slot, ...)` gets the pipe slot of the step
        (`slot = buffer slot * gather_every + position in the gather group`) and calls `pipe.submit(slot, ...)`: the
        mask kernel goes to the pipe's mask stream, the pick to its pick stream, and the all-gather is enqueued
        (asynchronously) behind the group's last pick on that same stream;
      * without a pipe `run` enqueues on the current stream (CPU, gloo tests: executes inline) and only the
        all-gather is asynchronous.
    gather_every = G > 1: the bindings of G consecutive steps share one buffer and ONE all-gather (fewer, larger
    collectives: an RCCL call costs tens of microseconds of host and launch time whatever its size, the same order as
    a step's kernels).  A step's result is then available once its group has been gathered (`wait()` flushes a partial
    group)."""

    def __init__(self, P: int, device: torch.device, depth: int = 2, group: Optional[dist.ProcessGroup] = None, pipe=None,
                 gather_always: bool = False, gather_every: int = 1):
        if depth < 1 or gather_every < 1:
            raise ValueError("depth >= 1, gather_every >= 1")
        if pipe is not None and pipe.depth != depth * gather_every:
            raise ValueError("a pipe needs pipe.depth == depth * gather_every (one pipe slot per step in flight)")
        self.P, self.device, self.depth, self.group, self.pipe, self.gather_every = P, device, depth, group, pipe, gather_every
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi, self.shard = shard_bounds(P, self.world, self.rank)
        # gather_always: run the all-gather even in a one-rank group (exercises the RCCL path on a single GPU; tests)
        self._gather = self.world > 1 or (gather_always and dist.is_initialized())
        G = gather_every
        self._local = [torch.full((G * self.shard,), -1, dtype=torch.int32, device=device) for _ in range(depth)]
        self._gathered = [torch.full((G * self.shard * self.world,), -1, dtype=torch.int32, device=device) if self._gather else None
                          for _ in range(depth)]
        self._work = [None] * depth
        self._used = [False] * depth
        self._fill = [0] * depth   # steps written into the slot since its last gather
        self._cur = 0              # slot being filled
        self._pick_stream = pipe.stream(1) if pipe is not None else None

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def binding_buffer(self, k: int, g: int = 0) -> torch.Tensor:
        """The int32 [n_local] buffer that step `g` of buffer slot `k`'s gather group writes (for callers that pre-marshal their
        launches: the same tensor `run` receives)."""
        return self._local[k][g * self.shard: g * self.shard + self.n_local]

    def _flush(self, k: int) -> None:
        """Issue the (asynchronous) all-gather of slot k's group and move on to the next slot."""
        if self._gather and self._fill[k] > 0:
            if self._pick_stream is not None:
                with torch.cuda.stream(self._pick_stream):
                    self._work[k] = dist.all_gather_into_tensor(self._gathered[k], self._local[k], group=self.group, async_op=True)
            else:
                self._work[k] = dist.all_gather_into_tensor(self._gathered[k], self._local[k], group=self.group, async_op=True)
        self._fill[k] = 0
        self._cur = (k + 1) % self.depth

    def step(self, run: Callable[[int, torch.Tensor], None]) -> PendingBindings:
        k, g = self._cur, self._fill[self._cur]
        if g == 0 and self._work[k] is not None:
            # first step into a reused slot: the all-gather that read its bindings must be done before they are overwritten
            if self._pick_stream is not None:
                with torch.cuda.stream(self._pick_stream):
                    self._work[k].wait()
            else:
                self._work[k].wait()
            self._work[k] = None
        out = self.binding_buffer(k, g)
        if self.n_local > 0:
            run(k * self.gather_every + g if self.pipe is not None else k, out)
            self._used[k] = True
        self._fill[k] = g + 1
        pending = PendingBindings(self, k, g)
        if self._fill[k] == self.gather_every:
            self._flush(k)
        return pending

    def drain(self) -> None:
        """Gather what is still ungathered and order the current stream (CPU: the host) after everything in flight."""
        if self._fill[self._cur] > 0:
            self._flush(self._cur)
        for k in range(self.depth):
            if self._gather:
                if self._work[k] is not None:
                    self._work[k].wait()
            elif self.pipe is not None and self._used[k]:
                for g in range(self.gather_every):
                    self.pipe.wait(k * self.gather_every + g)
