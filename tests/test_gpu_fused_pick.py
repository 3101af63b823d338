"""The sampled pick riding in the fused mask launch (KSCHED_OPT_FUSED_PICK, kernels_fused.hpp "PICK") and the
no-unwind rule of the C ABI (KSCHED_OPT_FAULT).

select_node_for_pod (src/main.rs:51-71): ATTEMPTS draws with replacement, the first feasible one wins, none -> NoNodeFound.
Three implementations must agree on every pod: the pick inside the mask launch, the stand-alone launch (k_select_sampled)
and the oracle's scalar loop -- and the mask written by the same launch must not notice that the pick rode along.
"""
import os

import numpy as np
import pytest

from kube_scheduler_rs_reference_amd import FIT, PICK_SAMPLED, SEL, TAINT, WANT_FIT_MASK, KschedError, _lib, synth
from oracle import capi

pytestmark = pytest.mark.gpu


def oracle_eval(c, flags):
    return capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels if c.n_keys else None, c.node_taints if c.n_taints else None,
                             c.req_cpu, c.req_mem, c.pod_sel if c.n_keys else None, c.pod_tol if c.n_taints else None, c.samples, flags)


def run(ev, c, flags, ride):
    """ride: False / 0 = the pick is its own launch; True / 1 = it rides, the library chooses the form; 2 = as waves of the fill
    (select_one_pod on the node records); 3 = as tile tests in phase 1 (KSCHED_E_UNSUPPORTED where that form does not apply)"""
    ev.set_option(_lib.OPT_FUSED_PICK, int(ride))
    try:
        pc = c.pod_columns()
        r = ev.eval(pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], pc["tolerations"], pc["samples"], flags)
        return r, ev.last_pick, ev.last_kernel
    finally:
        ev.set_option(_lib.OPT_FUSED_PICK, 1)


@pytest.mark.parametrize("P,N,attempts", [(1, 1, 5), (63, 65, 5), (300, 200, 5), (1000, 4097, 5), (5000, 1000, 1), (4099, 2500, 8),
                                          (2500, 700, 11), (20_000, 5_000, 5), (70_000, 1_100, 5), (3_000, 12_000, 5), (130, 40_000, 5)])
@pytest.mark.parametrize("flags", [FIT, FIT | SEL, FIT | SEL | TAINT, SEL, 0])
def test_riding_pick_equals_standalone_pick_equals_oracle(evaluator, P, N, attempts, flags):
    ev = evaluator
    ev.set_kernel("fused")
    try:
        c = synth.make_cluster(P, N, n_keys=8, n_taints=16, seed=P * 31 + N * 7 + attempts, attempts=attempts)
        ev.set_nodes(**c.node_columns())
        f = flags | PICK_SAMPLED
        feas, _, bind = oracle_eval(c, f)
        # the tile-test form: ATTEMPTS draws, the reference's two predicates (a taint predicate no node's taints make active is none)
        tile_ok = attempts == 5 and not ((flags & TAINT) and c.node_taints.any())
        a, how_a, kern_a = run(ev, c, f, ride=1)
        assert kern_a == "fused" and how_a == ("fused-tile" if (tile_ok and 1024 < N <= 12 * 1024) else "fused")  # (the library picks the tile form from two to twelve tiles)
        w, how_w, _ = run(ev, c, f, ride=2)
        b, how_b, _ = run(ev, c, f, ride=0)
        assert how_w == "fused" and how_b == "select"
        for r, what in ((a, "riding pick (form chosen by the library)"), (w, "riding pick, waves of the fill"), (b, "stand-alone pick")):
            assert np.array_equal(r.binding, bind), what + " vs oracle"
            assert np.array_equal(r.feasible, feas), what + ": the mask does not notice the pick"
        if tile_ok:
            for rep in range(3):  # the accumulators are left at zero by every launch: the same call again gives the same bindings
                t, how_t, _ = run(ev, c, f, ride=3)
                assert how_t == "fused-tile" and np.array_equal(t.binding, bind) and np.array_equal(t.feasible, feas), rep
        else:
            with pytest.raises(KschedError) as ei:
                run(ev, c, f, ride=3)
            assert ei.value.code == _lib.E_UNSUPPORTED
    finally:
        ev.set_kernel("auto")


def test_riding_pick_out_of_range_draws_and_no_feasible_node(evaluator):
    """Draws >= N are infeasible draws (include/ksched.h); a pod no draw fits gets -1 (NoNodeFound, src/main.rs:117)."""
    ev = evaluator
    c = synth.make_cluster(3000, 900, n_keys=8, n_taints=0, seed=4242)
    rng = np.random.default_rng(7)
    c.samples[rng.random(c.samples.shape) < 0.3] = np.uint32(900 + 5)   # out of range
    c.samples[:50, :] = np.uint32(0xFFFFFFFF)                          # a pod whose every draw is out of range
    c.req_cpu[100:150] = np.int64(1) << 60                             # pods nothing can hold
    ev.set_nodes(**c.node_columns())
    ev.set_kernel("fused")
    try:
        _, _, bind = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
        for form, name in ((3, "fused-tile"), (2, "fused")):
            a, how, _ = run(ev, c, FIT | SEL | PICK_SAMPLED, ride=form)
            assert how == name
            assert np.array_equal(a.binding, bind), name
            assert (a.binding[:50] == -1).all() and (a.binding[100:150] == -1).all()
    finally:
        ev.set_kernel("auto")


def test_the_pick_does_not_ride_where_it_cannot(evaluator):
    """A second mask (WANT_FIT_MASK), the direct kernel, a bindings-only request: the stand-alone launch, same bindings."""
    ev = evaluator
    c = synth.make_cluster(2000, 1500, n_keys=8, n_taints=0, seed=99)
    ev.set_nodes(**c.node_columns())
    pc = c.pod_columns()
    _, _, bind = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
    r = ev.eval(pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], FIT | SEL | PICK_SAMPLED | WANT_FIT_MASK)
    assert ev.last_pick == "select" and np.array_equal(r.binding, bind)
    r = ev.eval(pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], FIT | SEL | PICK_SAMPLED, want_mask=False)
    assert ev.last_pick == "select" and np.array_equal(r.binding, bind)
    ev.set_kernel("direct")
    try:
        r = ev.eval(pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], FIT | SEL | PICK_SAMPLED)
        assert ev.last_pick == "select" and ev.last_kernel == "direct" and np.array_equal(r.binding, bind)
    finally:
        ev.set_kernel("auto")
    r = ev.eval(pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], FIT | SEL | PICK_SAMPLED)
    assert ev.last_pick == "fused-tile" and np.array_equal(r.binding, bind)  # (1500 nodes: two tiles)


def test_tile_pick_more_than_eight_keys_and_partial_tiles(evaluator):
    """More than eight label keys: the tile-test form does not apply (a pod's selector must fit the eight slots of its record) and the
    library falls back to the waves of the fill.  Node counts around the 1024-node tile edge: draws on the padding bits of the
    last, partial tile, and in tiles this block does not own."""
    ev = evaluator
    rng = np.random.default_rng(3)
    ev.set_kernel("fused")
    try:
        for N in (1, 63, 1023, 1024, 1025, 2049, 3000):
            c = synth.make_cluster(1200, N, n_keys=8, n_taints=0, seed=50 + N)
            c.samples[:] = rng.integers(0, N + 40, size=c.samples.shape, dtype=np.uint32)  # some draws fall past the last node
            ev.set_nodes(**c.node_columns())
            _, _, bind = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
            a, how, _ = run(ev, c, FIT | SEL | PICK_SAMPLED, ride=3)
            assert how == "fused-tile" and np.array_equal(a.binding, bind), N
        c = synth.make_cluster(900, 2500, n_keys=8, n_taints=0, seed=8)
        lab = np.concatenate([c.node_labels, rng.integers(0, 3, size=(3, c.N), dtype=np.uint32)])
        sel = np.concatenate([c.pod_sel, np.where(rng.random((3, c.P)) < 0.3, rng.integers(1, 4, size=(3, c.P)), 0).astype(np.uint32)])
        ev.set_nodes(c.avail_cpu, c.avail_mem, lab, None)
        want = capi.eval_encoded(c.avail_cpu, c.avail_mem, lab, None, c.req_cpu, c.req_mem, sel, None, c.samples, FIT | SEL | PICK_SAMPLED)[2]
        r = ev.eval(c.req_cpu, c.req_mem, sel, None, c.samples, FIT | SEL | PICK_SAMPLED)
        assert ev.last_pick == "fused" and np.array_equal(r.binding, want)
    finally:
        ev.set_kernel("auto")


def test_riding_pick_on_device_buffers_repeated_steps(evaluator):
    """The bench's form: device-resident inputs, pitched mask rows, the same launch over and over on one stream."""
    import torch
    ev = evaluator
    c = synth.make_config("C3", P=30_000)
    ev.set_nodes(**c.node_columns())
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d_cpu, d_mem, d_sel, d_smp = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), t(c.samples, np.int32)
    mask = ev.alloc_mask(c.P, pitched=True)
    out = torch.full((c.P,), -7, dtype=torch.int32, device=dev)
    feas, _, bind = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
    for _ in range(5):
        out.fill_(-7)
        ev.eval_device(d_cpu, d_mem, d_sel, None, d_smp, FIT | SEL | PICK_SAMPLED, out_feasible=mask, out_binding=out)
        torch.cuda.synchronize()
        assert ev.last_pick == "fused-tile"
        assert np.array_equal(out.cpu().numpy(), bind)
    assert np.array_equal(mask.cpu().numpy().view(np.uint64), feas)
    # two streams in turn (what the pipe's alternate mode does): each stream has its own accumulators, launches may overlap
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    outs = [torch.full((c.P,), -7, dtype=torch.int32, device=dev) for _ in range(6)]
    masks = [ev.alloc_mask(c.P, pitched=True) for _ in range(2)]
    torch.cuda.synchronize()
    for j in range(6):
        ev.eval_device(d_cpu, d_mem, d_sel, None, d_smp, FIT | SEL | PICK_SAMPLED, out_feasible=masks[j % 2], out_binding=outs[j], stream=(s1, s2)[j % 2])
    torch.cuda.synchronize()
    for j in range(6):
        assert np.array_equal(outs[j].cpu().numpy(), bind), j
    ev.forget_stream(s1)
    ev.forget_stream(s2)


def test_how_long_a_launch_keeps_its_pick(evaluator):
    """The tile-test pick rides in the mask launch up to 524 288 pods per call (as long as an XCD's share of the batch's operands and draws stays in its
    L2, the tile-blocks of a pod range re-read them from there: session r7a, 400k x 5k 53.4 us riding against 65.9 us with the pick as its own launch,
    800k x 5k 143.4 against 125.3); beyond that the default is the stand-alone kernel.  Every form at both sizes == oracle."""
    import torch
    ev = evaluator
    c = synth.make_config("C3", P=540_000)  # one step beyond the limit
    ev.set_nodes(**c.node_columns())
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d_cpu, d_mem, d_sel, d_smp = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), t(c.samples, np.int32)
    mask = ev.alloc_mask(c.P, pitched=True)
    out = torch.full((c.P,), -7, dtype=torch.int32, device=dev)
    _, _, bind = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
    try:
        for mode, want in ((1, "select"), (3, "fused-tile"), (2, "fused")):
            ev.set_option(_lib.OPT_FUSED_PICK, mode)
            out.fill_(-7)
            ev.eval_device(d_cpu, d_mem, d_sel, None, d_smp, FIT | SEL | PICK_SAMPLED, out_feasible=mask, out_binding=out)
            torch.cuda.synchronize()
            assert ev.last_pick == want, (mode, ev.last_pick)
            assert np.array_equal(out.cpu().numpy(), bind), mode
        ev.set_option(_lib.OPT_FUSED_PICK, 1)
        for small in (524_288, 400_000):  # (inside the limit: rides)
            out.fill_(-7)
            ev.eval_device(d_cpu[:small].contiguous(), d_mem[:small].contiguous(), d_sel[:, :small].contiguous(), None, d_smp[:small].contiguous(),
                           FIT | SEL | PICK_SAMPLED, out_feasible=ev.alloc_mask(small, pitched=True), out_binding=out[:small])
            torch.cuda.synchronize()
            assert ev.last_pick == "fused-tile" and np.array_equal(out[:small].cpu().numpy(), bind[:small]), small
    finally:
        ev.set_option(_lib.OPT_FUSED_PICK, 1)


# ---- nothing unwinds across the C ABI (include/ksched.h "Conventions") -------------------------------------

FAULT_CASE = r'''
import sys
import numpy as np
from kube_scheduler_rs_reference_amd import Evaluator, FIT, SEL, PICK_SAMPLED, _lib, synth
from kube_scheduler_rs_reference_amd._lib import KschedError
kind, code = int(sys.argv[1]), int(sys.argv[2])
assert _lib.load().ksched_test_hooks_linked() == 1  # the TEST build (KSCHED_LIB): the shipped library has no fault injection
ev = Evaluator(0)
c = synth.make_cluster(500, 300, n_keys=8, n_taints=0, seed=5)
ev.set_nodes(**c.node_columns())
pc = c.pod_columns()
args = (pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], FIT | SEL | PICK_SAMPLED)
good = ev.eval(*args)


def raises(fn, want=None, text=None):
    try:
        fn()
    except KschedError as e:
        assert want is None or e.code == want, (e.code, want)
        assert text is None or text in str(e), str(e)
        return
    raise AssertionError("no KschedError")


# (a) inside an evaluation
ev.set_option(_lib.OPT_FAULT, kind)
raises(lambda: ev.eval(*args), code, "exception inside the library")
again = ev.eval(*args)  # one shot: the next call is normal, on an intact snapshot
assert np.array_equal(again.feasible, good.feasible) and np.array_equal(again.binding, good.binding)
# (b) inside ksched_update_nodes, before anything changed: the snapshot stays valid and unchanged
ev.set_option(_lib.OPT_FAULT, kind)
raises(lambda: ev.update_nodes(np.array([1, 2], dtype=np.uint32), np.array([0, 0], dtype=np.int64), np.array([0, 0], dtype=np.int64)), code)
again = ev.eval(*args)
assert np.array_equal(again.feasible, good.feasible)
# (c) inside ksched_set_nodes: a half-built snapshot is never evaluated (KSCHED_E_STATE) until the next successful set_nodes
ev.set_option(_lib.OPT_FAULT, kind)
raises(lambda: ev.set_nodes(**c.node_columns()), code)
raises(lambda: ev.eval(*args), _lib.E_STATE)
ev.set_nodes(**c.node_columns())
again = ev.eval(*args)
assert np.array_equal(again.feasible, good.feasible) and np.array_equal(again.binding, good.binding)
# (d) a deeper fault point: skip one (the evaluation's own), throw at the next (the following call)
ev.set_option(_lib.OPT_FAULT, kind | (1 << 8))
ev.eval(*args)
raises(lambda: ev.eval(*args))
ev.eval(*args)
# (e) ksched_mask_alloc is an entry into the library's C++ like any other
ev.set_option(_lib.OPT_FAULT, kind)
raises(lambda: ev.alloc_mask(100), code)
assert ev.alloc_mask(100).shape[0] == 100
print("fault case ok")
'''


@pytest.mark.parametrize("kind,code", [(1, _lib.E_NOMEM), (2, _lib.E_INVAL)])
def test_an_exception_inside_the_library_comes_back_as_a_code(built, kind, code):
    """KSCHED_OPT_FAULT makes the next entry into the library's C++ throw (std::bad_alloc / std::runtime_error): the call returns
    KSCHED_E_NOMEM / KSCHED_E_INVAL with ksched_last_error set -- the process is still alive to assert it -- and the ctx works on.  Fault
    injection exists in the TEST build of the library only (tests/cpp/hooks/libksched_hip.so: the shipped object code + tests/cpp/test_hooks.cpp),
    so the case runs in a process of its own that loads that build ($KSCHED_LIB)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", FAULT_CASE, str(kind), str(code)], capture_output=True, text=True, timeout=300, cwd=root,
                       env=dict(os.environ, KSCHED_LIB=os.path.join(root, "tests", "cpp", "hooks", "libksched_hip.so")))
    assert r.returncode == 0 and "fault case ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_the_shipped_library_refuses_fault_injection(evaluator):
    """... and the shipped library answers KSCHED_E_UNSUPPORTED: nothing in a production process can make it throw on request."""
    with pytest.raises(KschedError) as ei:
        evaluator.set_option(_lib.OPT_FAULT, 1)
    assert ei.value.code == _lib.E_UNSUPPORTED


# ---- the pipe's alternate mode (KSCHED_OPT_PIPE_MODE = 1): whole steps on stream (slot mod 2) --------------------------------------

@pytest.mark.parametrize("with_gather", [False, True])
def test_pipe_alternate_mode_equals_oracle(evaluator, with_gather):
    """Consecutive batches alternate between the pipe's two streams, each batch ONE launch (the pick rides) -- and, with the C ABI's
    RCCL communicator, each batch's all-gather behind it on its own stream (the N > 1 default of bench.py, here in a one-rank group).
    Slots reused over nine steps; every step's bindings and masks == oracle; wait_mask orders a consumer behind the slot's launch."""
    import torch
    from kube_scheduler_rs_reference_amd.dist import AbiComm, PipelinedScheduler
    ev = evaluator
    c = synth.make_cluster(6000, 2600, n_keys=8, n_taints=0, seed=314)
    ev.set_nodes(**c.node_columns())
    dev = torch.device("cuda", ev.device)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    steps, depth = 9, 4
    rolled = [np.roll(np.arange(c.P), 17 * j) for j in range(steps)]
    batches = [dict(cpu=t(c.req_cpu[r], np.int64), mem=t(c.req_mem[r], np.int64), sel=t(c.pod_sel[:, r], np.int32), smp=t(c.samples[r], np.int32)) for r in rolled]
    feas, _, base = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
    comm = AbiComm(ev) if with_gather else None
    ev.set_option(_lib.OPT_PIPE_MODE, 1)
    pipe = ev.pipe(depth)
    try:
        sched = PipelinedScheduler(c.P, dev, depth=depth, pipe=pipe, gather_always=with_gather, gather_every=1, comm=comm, alternate=True)
        masks = [ev.alloc_mask(c.P) for _ in range(depth)]
        state = {"j": 0}

        def run(slot, out):
            b = batches[state["j"]]
            pipe.submit(slot, b["cpu"], b["mem"], b["sel"], None, b["smp"], FIT | SEL | PICK_SAMPLED, masks[slot], out)

        got, pend = [], []
        for j in range(steps):
            state["j"] = j
            pend.append((j, sched.step(run)))
            assert ev.last_pick == "fused-tile"  # (2600 nodes: three tiles -- one launch per step)
            if len(pend) >= depth:
                jj, p = pend.pop(0)
                got.append((jj, p.wait().clone()))
        got += [(jj, p.wait().clone()) for jj, p in pend]
        sched.drain()
        # the last step's mask, read by a consumer stream ordered by wait_mask
        consumer = torch.cuda.Stream(device=dev)
        last_slot = (steps - 1) % depth
        pipe.wait_mask(last_slot, stream=consumer)
        with torch.cuda.stream(consumer):
            m = masks[last_slot].clone()
        consumer.synchronize()
        torch.cuda.synchronize()
        for jj, b in got:
            assert np.array_equal(b.cpu().numpy(), base[rolled[jj]]), f"step {jj}"
        assert np.array_equal(m.cpu().numpy().view(np.uint64), feas[rolled[steps - 1]])
    finally:
        pipe.close()
        ev.set_option(_lib.OPT_PIPE_MODE, 0)
        if comm is not None:
            comm.close()


# ---- several batches in flight on shares of the chip (KSCHED_OPT_PIPE_MODE = k streams, KSCHED_OPT_GRID_CUS) ------------------------------

@pytest.mark.parametrize("k,cus,depth", [(2, 128, 4), (3, 88, 6), (4, 64, 8), (3, 96, 4), (8, 32, 8), (2, 0, 3)])
def test_pipe_k_streams_on_a_share_of_the_chip_equals_oracle(evaluator, k, cus, depth):
    """The pipe's alternate mode over k streams with every mask launch kept to `cus` compute units (the next batches' launches fill on the
    others while this one stores): every step's bindings and masks == oracle, also when depth is not a multiple of k (a slot then moves
    between streams and is ordered by its event) and when the mode changes back to split on the same pipe."""
    import torch
    ev = evaluator
    c = synth.make_cluster(9000, 5000, n_keys=8, n_taints=0, seed=2718)
    ev.set_nodes(**c.node_columns())
    dev = torch.device("cuda", ev.device)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    steps = 3 * depth + 1
    rolled = [np.roll(np.arange(c.P), 29 * j) for j in range(steps)]
    batches = [dict(cpu=t(c.req_cpu[r], np.int64), mem=t(c.req_mem[r], np.int64), sel=t(c.pod_sel[:, r], np.int32), smp=t(c.samples[r], np.int32)) for r in rolled]
    feas, _, base = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
    ev.set_option(_lib.OPT_PIPE_MODE, k)
    ev.set_option(_lib.OPT_GRID_CUS, cus)
    pipe = ev.pipe(depth)
    try:
        masks = [ev.alloc_mask(c.P) for _ in range(depth)]
        outs = [torch.full((c.P,), -7, dtype=torch.int32, device=dev) for _ in range(depth)]
        torch.cuda.synchronize()  # (the fills run on torch's stream; the pipe's streams are non-blocking ones and are not ordered behind it)
        assert pipe.slot_stream(0) is None
        for j in range(steps):
            slot = j % depth
            if j >= depth:  # the slot's previous batch, before it is overwritten
                pipe.wait(slot, host=True)
                jj = j - depth
                assert np.array_equal(outs[slot].cpu().numpy(), base[rolled[jj]]), f"step {jj}"
                pipe.wait_mask(slot, host=True)
                if jj % 4 == 0:
                    assert np.array_equal(masks[slot].cpu().numpy().view(np.uint64), feas[rolled[jj]]), f"mask of step {jj}"
            b = batches[j]
            if j == steps - 2:
                ev.set_option(_lib.OPT_PIPE_MODE, 0)  # one split-mode batch in between: ordered against the slot's alternate-mode past
            pipe.submit(slot, b["cpu"], b["mem"], b["sel"], None, b["smp"], FIT | SEL | PICK_SAMPLED, masks[slot], outs[slot])
            if j == steps - 2:
                ev.set_option(_lib.OPT_PIPE_MODE, k)
            assert pipe.slot_stream(slot) is not None
        torch.cuda.synchronize()
        for j in range(steps - depth, steps):
            assert np.array_equal(outs[j % depth].cpu().numpy(), base[rolled[j]]), f"step {j}"
            assert np.array_equal(masks[j % depth].cpu().numpy().view(np.uint64), feas[rolled[j]]), f"mask of step {j}"
    finally:
        pipe.close()
        ev.set_option(_lib.OPT_PIPE_MODE, 0)
        ev.set_option(_lib.OPT_GRID_CUS, 0)


def test_grid_cus_option_is_validated_and_changes_nothing_in_the_results(evaluator):
    ev = evaluator
    for bad in (1, 7, 257, -1):
        with pytest.raises(KschedError):
            ev.set_option(_lib.OPT_GRID_CUS, bad)
    with pytest.raises(KschedError):
        ev.set_option(_lib.OPT_PIPE_MODE, _lib.PIPE_MAX_STREAMS + 1)
    c = synth.make_cluster(3000, 2100, n_keys=8, n_taints=0, seed=99)
    ev.set_nodes(**c.node_columns())
    feas, _, base = oracle_eval(c, FIT | SEL | PICK_SAMPLED)
    try:
        for cus in (0, 8, 24, 100, 256):
            ev.set_option(_lib.OPT_GRID_CUS, cus)
            r = ev.eval(c.req_cpu, c.req_mem, c.pod_sel, None, c.samples, FIT | SEL | PICK_SAMPLED)
            assert np.array_equal(r.feasible, feas) and np.array_equal(r.binding, base), cus
    finally:
        ev.set_option(_lib.OPT_GRID_CUS, 0)


def test_alternate_scheduler_refuses_a_request_the_pipe_runs_in_split_mode(evaluator):
    """with a pick that reads the mask (KSCHED_OPT_PICK_FROM_MASK) ksched_pipe_submit falls back to its split mode -- the pick on the
    pick stream whatever the slot -- so an all-gather pre-bound to stream (slot mod 2) would not be ordered behind it.  The scheduler asks the
    pipe which stream carried the slot (ksched_pipe_slot_stream) and refuses loudly instead of gathering stale bindings."""
    import torch
    from kube_scheduler_rs_reference_amd.dist import AbiComm, PipelinedScheduler
    ev = evaluator
    c = synth.make_cluster(3000, 2600, n_keys=8, n_taints=0, seed=77)
    ev.set_nodes(**c.node_columns())
    dev = torch.device("cuda", ev.device)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d = dict(cpu=t(c.req_cpu, np.int64), mem=t(c.req_mem, np.int64), sel=t(c.pod_sel, np.int32), smp=t(c.samples, np.int32))
    comm = AbiComm(ev)
    ev.set_option(_lib.OPT_PIPE_MODE, 1)
    ev.set_option(_lib.OPT_PICK_FROM_MASK, 1)
    pipe = ev.pipe(2)
    try:
        sched = PipelinedScheduler(c.P, dev, depth=2, pipe=pipe, gather_always=True, gather_every=1, comm=comm, alternate=True)
        masks = [ev.alloc_mask(c.P) for _ in range(2)]

        def run(slot, out):
            pipe.submit(slot, d["cpu"], d["mem"], d["sel"], None, d["smp"], FIT | SEL | PICK_SAMPLED, masks[slot], out)
        with pytest.raises(RuntimeError, match="split mode"):
            sched.step(run)  # slot 0: its pick went onto the pick stream, not onto stream 0
        torch.cuda.synchronize()
    finally:
        pipe.close()
        ev.set_option(_lib.OPT_PICK_FROM_MASK, 0)
        ev.set_option(_lib.OPT_PIPE_MODE, 0)
        comm.close()

