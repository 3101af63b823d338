"""Multi-GPU: pod rows shard across ranks, node snapshot is replicated, bindings are all-gathered.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  Pods are independent given the node snapshot, so the only
exchange step is one all-gather of the int32 bindings (4 B per pod); masks stay on the GPU that
produced them.

    rank r owns pod rows [r * shard, min(P, (r + 1) * shard)),  shard = ceil(P / world)

`ShardedScheduler` is transport + bookkeeping only; the evaluation itself is the `local_eval`
callable (in the product: `Evaluator.eval_device` on this rank's GPU).

`PipelinedScheduler` is the throughput form of the same thing: consecutive batches ("steps") are
software-pipelined over two HIP streams with `depth` buffer slots -- the mask kernel of step i + 1
runs while the pick kernel and the all-gather of step i are still in flight (they do not depend
on each other; per slot, events order mask -> pick -> all-gather -> next use of the slot).  Every
step's outputs are complete and identical to the sequential form.
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
        if s.world > 1:
            if s._work[k] is not None:
                s._work[k].wait()
            return s._gathered[k][: s.P]
        if s._streams:
            torch.cuda.current_stream(s.device).wait_event(s._pick_done[k])
        return s._local[k][: s.P]


class PipelinedScheduler:
    """Row-sharded evaluation, pick and all-gather of consecutive batches, `depth` steps in flight.

    step(mask_fn, pick_fn):
        mask_fn(slot)                 enqueue this rank's mask evaluation into the caller's mask buffer `slot`
        pick_fn(slot, binding_out)    enqueue the pick from that mask into binding_out[: n_local] (int32)
    On a GPU the two run on two side streams; on the CPU (gloo tests) they run inline and only the all-gather is
    asynchronous."""

    def __init__(self, P: int, device: torch.device, depth: int = 2, group: Optional[dist.ProcessGroup] = None):
        if depth < 1:
            raise ValueError("depth >= 1")
        self.P, self.device, self.depth, self.group = P, device, depth, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi, self.shard = shard_bounds(P, self.world, self.rank)
        self._local = [torch.full((self.shard,), -1, dtype=torch.int32, device=device) for _ in range(depth)]
        self._gathered = [torch.full((self.shard * self.world,), -1, dtype=torch.int32, device=device) if self.world > 1 else None
                          for _ in range(depth)]
        self._work = [None] * depth
        self._streams = device.type == "cuda"
        if self._streams:
            self.s_mask, self.s_pick = torch.cuda.Stream(device), torch.cuda.Stream(device)
            self._mask_done = [torch.cuda.Event() for _ in range(depth)]
            self._pick_done = [torch.cuda.Event() for _ in range(depth)]
            cur = torch.cuda.current_stream(device)
            for e in self._mask_done + self._pick_done:  # "nothing pending" for the first use of every slot
                e.record(cur)
        self._i = 0

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def step(self, mask_fn: Callable[[int], None], pick_fn: Callable[[int, torch.Tensor], None]) -> PendingBindings:
        k = self._i % self.depth
        self._i += 1
        out = self._local[k][: self.n_local]
        if self._streams:
            cur = torch.cuda.current_stream(self.device)
            self.s_mask.wait_stream(cur)                 # inputs prepared on the caller's stream
            self.s_mask.wait_event(self._pick_done[k])   # the pick that read this slot's mask `depth` steps ago
            with torch.cuda.stream(self.s_mask):
                if self.n_local > 0:
                    mask_fn(k)
                self._mask_done[k].record(self.s_mask)
            with torch.cuda.stream(self.s_pick):
                self.s_pick.wait_event(self._mask_done[k])
                if self._work[k] is not None:            # the all-gather that read this slot's bindings `depth` steps ago
                    self._work[k].wait()
                if self.n_local > 0:
                    pick_fn(k, out)
                self._pick_done[k].record(self.s_pick)
                if self.world > 1:
                    self._work[k] = dist.all_gather_into_tensor(self._gathered[k], self._local[k], group=self.group, async_op=True)
        else:
            if self._work[k] is not None:
                self._work[k].wait()
            if self.n_local > 0:
                mask_fn(k)
                pick_fn(k, out)
            if self.world > 1:
                self._work[k] = dist.all_gather_into_tensor(self._gathered[k], self._local[k], group=self.group, async_op=True)
        return PendingBindings(self, k)

    def drain(self) -> None:
        """Order the current stream (CPU: the host) after everything in flight."""
        for k in range(self.depth):
            PendingBindings(self, k).wait()
