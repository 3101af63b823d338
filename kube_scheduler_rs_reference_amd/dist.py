"""Multi-GPU: pod rows shard across ranks, node snapshot is replicated, bindings are all-gathered.

One process per GPU.  Pods are independent given the node snapshot, so the only
exchange step is one all-gather of the int32 bindings (4 B per pod); masks stay on the GPU that
produced them.

The data path of that exchange is the C ABI's own RCCL communicator (`AbiComm` = ksched_comm_create +
ksched_allgather_bindings, include/ksched.h: ncclAllGather enqueued on the pick's stream), the same entry points a
Rust or C++ host binds.  `torch.distributed` is only the launcher-side control plane here (rank / world size, handing
the 128-byte RCCL unique id to the other ranks, the bench's barrier); without a GPU (the world-size-2/3 "gloo" tests of
the sharding arithmetic) the gather falls back to torch's collective, which is test plumbing, not the product path.

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


class _EventWork:
    """`.wait()` orders the current stream after a recorded event (the stream-ordered equivalent of a torch Work)."""

    def __init__(self, event):
        self._event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self._event)


class _StreamWork:
    """`.wait()` orders the current stream after everything enqueued on `stream` so far; nothing is recorded per step (the
    gather was enqueued on `stream` behind the pick, so the stream's own order is the only synchronisation needed until
    someone on another stream wants the result)."""

    def __init__(self, stream):
        self._stream = stream

    def wait(self):
        cur = torch.cuda.current_stream(self._stream.device)
        if cur.cuda_stream != self._stream.cuda_stream:
            cur.wait_stream(self._stream)


class AbiComm:
    """RCCL communicator behind the C ABI (ksched_comm_*), one process per GPU.

    Rank 0 draws the unique id (ksched_comm_unique_id) and the launcher's process group -- any backend -- carries its 128
    bytes to the other ranks; ksched_comm_create is then collective over all ranks.  all_gather() enqueues
    ksched_allgather_bindings on a HIP stream and returns: no host sync, no torch collective on the data path."""

    def __init__(self, evaluator, group: Optional[dist.ProcessGroup] = None):
        import ctypes as C
        from . import _lib as L
        self._lib, self._ev = evaluator._lib, evaluator
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # Every rank draws an id: rank 0's is the one used, the others only prove that RCCL can be loaded on their side.  The
        # outcomes travel in ONE collective of the launcher's group, so a rank whose RCCL is unusable makes EVERY rank raise
        # here -- before ksched_comm_create, which blocks until all ranks have joined and would otherwise hang the healthy ones.
        ident = (C.c_uint8 * L.COMM_ID_BYTES)()
        rc = self._lib.ksched_comm_unique_id(C.cast(ident, C.c_void_p))
        mine = (None if rc == 0 else f"rank {self.rank}: ksched_comm_unique_id: {self._lib.ksched_comm_last_error().decode()}", bytes(ident))
        if self.world > 1:
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine, group=group)
        else:
            everyone = [mine]
        errors = [e for e, _ in everyone if e]
        if errors:
            raise L.KschedError(L.E_RCCL, "ksched_comm_unique_id", "; ".join(errors))
        ident = (C.c_uint8 * L.COMM_ID_BYTES).from_buffer_copy(everyone[0][1])
        h = C.c_void_p()
        self._check(self._lib.ksched_comm_create(evaluator._h, C.cast(ident, C.c_void_p), self.rank, self.world, C.byref(h)), "ksched_comm_create")
        self._h = h

    def _check(self, rc: int, where: str):
        if rc != 0:
            from . import _lib as L
            raise L.KschedError(rc, where, self._lib.ksched_comm_last_error().decode())

    def all_gather(self, gathered: torch.Tensor, local: torch.Tensor, stream=None) -> None:
        """gathered[r * len(local) + i] = rank r's local[i] (int32 CUDA tensors), enqueued on `stream` (default: current)."""
        import ctypes as C
        if local.dtype != torch.int32 or gathered.dtype != torch.int32 or not local.is_contiguous() or not gathered.is_contiguous():
            raise ValueError("bindings must be contiguous int32 CUDA tensors")
        if gathered.numel() != local.numel() * self.world:
            raise ValueError("gathered must hold world * len(local) entries")
        stream = stream or torch.cuda.current_stream(local.device)
        self._check(self._lib.ksched_allgather_bindings(self._h, C.c_void_p(local.data_ptr()), C.c_void_p(gathered.data_ptr()),
                                                        local.numel(), C.c_void_p(stream.cuda_stream)), "ksched_allgather_bindings")

    def bind_all_gather(self, gathered: torch.Tensor, local: torch.Tensor, stream):
        """The same call pre-marshalled (the checks of all_gather done once): returns a zero-argument callable that enqueues
        ksched_allgather_bindings for exactly these buffers on exactly this stream.  A step of the pipelined scheduler is tens of
        microseconds of device time; per-call argument checks and ctypes conversions are a measurable part of a host loop at that rate."""
        import ctypes as C
        if local.dtype != torch.int32 or gathered.dtype != torch.int32 or not local.is_contiguous() or not gathered.is_contiguous():
            raise ValueError("bindings must be contiguous int32 CUDA tensors")
        if gathered.numel() != local.numel() * self.world:
            raise ValueError("gathered must hold world * len(local) entries")
        fn, h = self._lib.ksched_allgather_bindings, self._h
        a_local, a_gathered, a_count, a_stream = C.c_void_p(local.data_ptr()), C.c_void_p(gathered.data_ptr()), C.c_uint32(local.numel()), C.c_void_p(stream.cuda_stream)
        check = self._check

        def call():
            rc = fn(h, a_local, a_gathered, a_count, a_stream)
            if rc != 0:
                check(rc, "ksched_allgather_bindings")
        return call

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ksched_comm_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


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
    comm: Optional[AbiComm] = None           # the C ABI's RCCL communicator (GPU); None = torch collective (CPU / gloo tests)

    def __post_init__(self):
        self.world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        self.lo, self.hi, self.shard = shard_bounds(self.P, self.world, self.rank)
        # equal-size slots so one all_gather_into_tensor moves everything; tail slots hold -1
        self.local = torch.full((self.shard,), -1, dtype=torch.int32, device=self.device)
        self.gathered = torch.full((self.shard * self.world,), -1, dtype=torch.int32, device=self.device)
        # the views a step hands out, made once: slicing a tensor costs 2 - 3 us of host time, and a step is 18 us of device time
        # (at --steps 20 the host's launch loop ran 10 - 15 us per step and the device waited for it, session r5s)
        self._n_local = self.hi - self.lo
        self._local_rows = self.local[: self._n_local]
        self._gathers = self.world > 1 or self.comm is not None
        self._result = (self.gathered if self._gathers else self.local)[: self.P]

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def step(self, local_eval: Callable[[torch.Tensor], None]) -> torch.Tensor:
        """local_eval(binding_out) must fill binding_out[: n_local] (int32, -1 = no node) for this
        rank's rows, enqueued on the current stream.  Returns the global bindings [P] (a view)."""
        if self._n_local > 0:
            local_eval(self._local_rows)
        if self._gathers:
            if self.comm is not None:
                self.comm.all_gather(self.gathered, self.local)  # ksched_allgather_bindings on the current stream
            else:
                dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
        return self._result


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
                 gather_always: bool = False, gather_every: int = 1, comm: Optional[AbiComm] = None, alternate: bool = False):
        """alternate = True (with a pipe in KSCHED_OPT_PIPE_MODE 1): the WHOLE step of a slot -- ONE launch when the pick rides in the
        mask kernel -- goes onto stream (slot mod 2) and its all-gather behind it on the same stream, so the gather of batch i
        overlaps the launch of batch i + 1 on the other stream (needs gather_every == 1 and an even depth)."""
        if depth < 1 or gather_every < 1:
            raise ValueError("depth >= 1, gather_every >= 1")
        if alternate and (pipe is None or gather_every != 1 or depth % 2):
            raise ValueError("alternate needs a pipe, gather_every == 1 and an even depth (a slot keeps its stream)")
        if pipe is not None and pipe.depth != depth * gather_every:
            raise ValueError("a pipe needs pipe.depth == depth * gather_every (one pipe slot per step in flight)")
        self.P, self.device, self.depth, self.group, self.pipe, self.gather_every = P, device, depth, group, pipe, gather_every
        self.comm = comm  # the C ABI's communicator: the gather is ksched_allgather_bindings, stream-ordered behind the pick
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi, self.shard = shard_bounds(P, self.world, self.rank)
        # gather_always: run the all-gather even in a one-rank group (exercises the RCCL path on a single GPU; tests)
        self._gather = self.world > 1 or (gather_always and (dist.is_initialized() or comm is not None))
        G = gather_every
        self._local = [torch.full((G * self.shard,), -1, dtype=torch.int32, device=device) for _ in range(depth)]
        self._gathered = [torch.full((G * self.shard * self.world,), -1, dtype=torch.int32, device=device) if self._gather else None
                          for _ in range(depth)]
        self._work = [None] * depth
        self._used = [False] * depth
        self._fill = [0] * depth   # steps written into the slot since its last gather
        self._cur = 0              # slot being filled
        self._pick_stream = pipe.stream(1) if pipe is not None else None
        self._alternate = alternate
        self._streams2 = (pipe.stream(0), pipe.stream(1)) if alternate else None  # alternate: slot k lives on stream k % 2
        self._stream2_handles = tuple(int(s_.cuda_stream) for s_ in self._streams2) if alternate else None
        # single-stream form with the ABI communicator: the gather runs on a side stream behind an event, like torch's async_op
        self._side = torch.cuda.Stream(device=device) if (comm is not None and pipe is None) else None
        self._ready = [torch.cuda.Event() for _ in range(depth)] if self._side is not None else None  # reused: no event creation per step
        self._done = [torch.cuda.Event() for _ in range(depth)] if self._side is not None else None
        # the host loop runs at the rate of the device steps (tens of microseconds): views and calls that do not change are made once
        self._views = [[self._local[k][g * self.shard: g * self.shard + self.n_local] for g in range(G)] for k in range(depth)]
        self._gather_calls = None
        if self._gather and comm is not None and self._pick_stream is not None:
            self._gather_calls = [comm.bind_all_gather(self._gathered[k], self._local[k], self._streams2[k & 1] if alternate else self._pick_stream)
                                  for k in range(depth)]
            self._stream_work = [_StreamWork(self._streams2[k & 1] if alternate else self._pick_stream) for k in range(depth)]

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def binding_buffer(self, k: int, g: int = 0) -> torch.Tensor:
        """The int32 [n_local] buffer that step `g` of buffer slot `k`'s gather group writes (for callers that pre-marshal their
        launches: the same tensor `run` receives)."""
        return self._views[k][g]

    def _flush(self, k: int) -> None:
        """Issue the (asynchronous) all-gather of slot k's group and move on to the next slot."""
        if self._gather and self._fill[k] > 0:
            if self.comm is not None:
                if self._pick_stream is not None:  # stream-ordered behind the group's last pick: no event, nothing to wait for
                    self._gather_calls[k]()
                    self._work[k] = self._stream_work[k]
                else:
                    self._ready[k].record(torch.cuda.current_stream(self.device))
                    self._side.wait_event(self._ready[k])
                    self.comm.all_gather(self._gathered[k], self._local[k], stream=self._side)
                    self._done[k].record(self._side)
                    self._work[k] = _EventWork(self._done[k])
            elif self._pick_stream is not None:
                with torch.cuda.stream(self._streams2[k & 1] if self._alternate else self._pick_stream):
                    self._work[k] = dist.all_gather_into_tensor(self._gathered[k], self._local[k], group=self.group, async_op=True)
            else:
                self._work[k] = dist.all_gather_into_tensor(self._gathered[k], self._local[k], group=self.group, async_op=True)
        self._fill[k] = 0
        self._cur = (k + 1) % self.depth

    def step(self, run: Callable[[int, torch.Tensor], None]) -> PendingBindings:
        k, g = self._cur, self._fill[self._cur]
        if g == 0 and self._work[k] is not None:
            # first step into a reused slot: the all-gather that read its bindings must be done before they are overwritten
            if self._gather_calls is not None:
                pass  # the slot's next pick goes onto the very stream its gather was enqueued on: the stream's own order is the wait
            elif self._pick_stream is not None:
                with torch.cuda.stream(self._streams2[k & 1] if self._alternate else self._pick_stream):
                    self._work[k].wait()
            else:
                self._work[k].wait()
            self._work[k] = None
        out = self.binding_buffer(k, g)
        if self.n_local > 0:
            run(k * self.gather_every + g if self.pipe is not None else k, out)
            if self._alternate:
                # The slot's all-gather is pre-bound to stream (k mod 2) and relies on the pick having gone onto that very stream.
                # ksched_pipe_submit falls back to its split mode when the pick reads the mask (KSCHED_OPT_PICK_FROM_MASK, a best-fit
                # pick without the bitmap index) and deals slots over more streams when KSCHED_OPT_PIPE_MODE > 2: the pick is then
                # somewhere else.  Asked after EVERY submit (one C call; the option can change between two steps: ADVICE r5).
                used = self.pipe.slot_stream_handle(k)
                if used != self._stream2_handles[k & 1]:
                    raise RuntimeError("PipelinedScheduler(alternate=True): the pipe ran this request in its split mode (the pick reads the mask), so the "
                                       "slot's all-gather would not be ordered behind its pick; use alternate=False for this request")
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
