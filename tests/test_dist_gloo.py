"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercise the row sharding and the
all-gather of bindings of kube_scheduler_rs_reference_amd.dist.  The per-rank evaluation itself
needs a GPU, so here a stand-in fills each rank's bindings from the oracle (test-only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kube_scheduler_rs_reference_amd import synth
from kube_scheduler_rs_reference_amd.dist import ShardedScheduler, shard_bounds
from oracle import capi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, P, N, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = synth.make_cluster(P, N, n_keys=8, n_taints=0, seed=99)
        sched = ShardedScheduler(P, torch.device("cpu"))
        lo, hi = sched.lo, sched.hi
        assert (lo, hi, sched.shard) == shard_bounds(P, world, rank)

        def local_eval(out):
            _, _, b = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, c.req_cpu[lo:hi], c.req_mem[lo:hi],
                                        c.pod_sel[:, lo:hi], None, c.samples[lo:hi], capi.FIT | capi.SEL | capi.PICK_SAMPLED,
                                        want_mask=False, threads=1)
            out.copy_(torch.from_numpy(b))

        for _ in range(2):  # two steps: buffers are reused
            got = sched.step(local_eval).clone().numpy()
        _, _, want = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, c.req_cpu, c.req_mem, c.pod_sel, None,
                                       c.samples, capi.FIT | capi.SEL | capi.PICK_SAMPLED, want_mask=False, threads=1)
        q.put((rank, bool(np.array_equal(got, want)), int((got >= 0).sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,P", [(2, 1000), (2, 1001), (3, 10), (2, 1)])
def test_sharded_allgather_matches_single_process(world, P):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, 200, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({n for _, _, n in res}) == 1  # every rank holds the same gathered bindings


def test_shard_bounds_cover_rows_exactly_once():
    for P in (0, 1, 7, 8, 9, 1000, 1_000_000):
        for world in (1, 2, 3, 4, 8):
            seen = 0
            prev_hi = 0
            for r in range(world):
                lo, hi, shard = shard_bounds(P, world, r)
                assert lo == min(P, prev_hi) or lo == prev_hi
                assert hi - lo <= shard
                seen += hi - lo
                prev_hi = hi
            assert seen == P


def _pipe_worker(rank, world, port, P, N, depth, steps, q, gather_every=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kube_scheduler_rs_reference_amd.dist import PipelinedScheduler
        c = synth.make_cluster(P, N, n_keys=8, n_taints=0, seed=7)
        sched = PipelinedScheduler(P, torch.device("cpu"), depth=depth, gather_every=gather_every)
        lo, hi = sched.lo, sched.hi
        flags = capi.FIT | capi.SEL | capi.PICK_SAMPLED
        state = {"step": 0}

        def samples_of(j):  # every step is a different batch: the draws rotate
            return np.ascontiguousarray(np.roll(c.samples, j, axis=0))

        def run(slot, out):  # stand-in for the device evaluation + pick (test-only: the oracle)
            j = state["step"]
            _, _, b = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, c.req_cpu[lo:hi], c.req_mem[lo:hi],
                                        c.pod_sel[:, lo:hi], None, samples_of(j)[lo:hi], flags, want_mask=False, threads=1)
            out.copy_(torch.from_numpy(b))

        ok = True
        pend = []
        for j in range(steps):
            state["step"] = j
            pend.append((j, sched.step(run)))
            if len(pend) >= max(1, (depth - 1) * gather_every):  # consume the oldest while newer steps are in flight (its slot is reused soon)
                jj, pb = pend.pop(0)
                got = pb.wait().clone().numpy()
                _, _, want = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, c.req_cpu, c.req_mem, c.pod_sel, None,
                                               samples_of(jj), flags, want_mask=False, threads=1)
                ok = ok and bool(np.array_equal(got, want))
        for jj, pb in pend:
            got = pb.wait().clone().numpy()
            _, _, want = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, c.req_cpu, c.req_mem, c.pod_sel, None,
                                           samples_of(jj), flags, want_mask=False, threads=1)
            ok = ok and bool(np.array_equal(got, want))
        sched.drain()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,P,depth,every", [(2, 501, 2, 1), (2, 64, 3, 1), (3, 10, 2, 1), (2, 1, 2, 1), (2, 301, 2, 3), (3, 77, 2, 4), (2, 50, 1, 2)])
def test_pipelined_allgather_matches_single_process(world, P, depth, every):
    """PipelinedScheduler: `depth` slots in flight, asynchronous all-gather (one per `every` steps), slots reused, partial last
    group -- every step's global bindings equal the single-process result of THAT step's batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, P, 150, depth, 7, q, every)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
