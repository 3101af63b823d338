"""BASELINE.json's full single-GPU sizes (C3 100k x 5k, the C4 shard 125k x 10k, the C5 shard 125k x 50k):

  * EVERY mask word (feasible and fit) and EVERY binding == the oracle's encoded restatement (ora_eval_encoded, OpenMP over
    pods: 5e8 .. 6.25e9 pairs take 0.05 .. 0.6 s on the GPU box's host cores), compared in row chunks so that only one
    chunk of either side is in host memory at a time;
  * fused == direct on every mask word (two independent HIP implementations: bitmap index vs per-pair compares);
  * feasible is a subset of fit; padding bits beyond node N are zero;
  * pod-permutation equivariance of a checksum of row checksums (rows do not depend on their neighbours or on
    which block / wave / round evaluated them);
  * bindings are consistent with the mask (sampled: the bound draw's bit is set and every earlier draw's is clear;
    best fit: a pod is bound iff its row is non-empty, and the bound node's bit is set).

The property checks stay on the device; the exhaustive comparison streams the masks back chunk by chunk.
"""
import numpy as np
import pytest

from kube_scheduler_rs_reference_amd import FIT, PICK_BESTFIT, PICK_SAMPLED, SEL, TAINT, WANT_FIT_MASK, synth
from oracle import capi

pytestmark = pytest.mark.gpu

CASES = {
    "C3": ("C3", 100_000, 5_000, FIT | SEL, PICK_SAMPLED),
    "C4s": ("C4", 125_000, 10_000, FIT | SEL, PICK_SAMPLED),
    "C5s": ("C5", 125_000, 50_000, FIT | SEL | TAINT, PICK_BESTFIT),
}
CHUNK_ROWS = 16384  # rows of the masks compared with the oracle per pass (C5 shard: 100 MB per mask and chunk)


def _row_checksums(mask):
    """[P, W] int64 device tensor -> [P] int64: position-weighted wrapping sum of the row's words."""
    import torch
    W = mask.shape[1]
    w = (torch.arange(1, W + 1, device=mask.device, dtype=torch.int64) * 0x9E3779B97F4A7C1) | 1
    return (mask * w).sum(dim=1)


@pytest.mark.parametrize("name", list(CASES))
def test_full_size_properties(evaluator, name):
    import torch
    cfg, P, N, preds, pick = CASES[name]
    c = synth.make_config(cfg, P=P, N=N)
    ev = evaluator
    dev = torch.device("cuda", ev.device)
    ev.set_nodes(**c.node_columns())
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d_cpu, d_mem = t(c.req_cpu, np.int64), t(c.req_mem, np.int64)
    d_sel = t(c.pod_sel, np.int32)
    d_tol = t(c.pod_tol, np.int64) if preds & TAINT else None
    d_smp = t(c.samples, np.int32)
    flags = preds | WANT_FIT_MASK | pick
    W = ev.W

    def run(kernel, cpu, mem, sel, tol, smp):
        ev.set_kernel(kernel)
        feas, fit = ev.alloc_mask(P), ev.alloc_mask(P)
        bind = torch.empty((P,), dtype=torch.int32, device=dev)
        ev.eval_device(cpu, mem, sel, tol, smp if pick == PICK_SAMPLED else None, flags, out_feasible=feas, out_fit=fit, out_binding=bind)
        torch.cuda.synchronize()
        assert ev.last_kernel == kernel
        return feas, fit, bind

    feas, fit, bind = run("fused", d_cpu, d_mem, d_sel, d_tol, d_smp)
    feas_d, fit_d, bind_d = run("direct", d_cpu, d_mem, d_sel, d_tol, d_smp)
    # two independent implementations agree on every word and every binding
    assert torch.equal(feas, feas_d), "fused != direct (feasible)"
    assert torch.equal(fit, fit_d), "fused != direct (fit)"
    assert torch.equal(bind, bind_d)
    del feas_d, fit_d, bind_d
    # feasible is a subset of fit; padding bits are zero
    assert not (feas & ~fit).any()
    if N % 64:
        pad = torch.tensor(-1 << (N % 64), dtype=torch.int64, device=dev)
        assert not (feas[:, W - 1] & pad).any() and not (fit[:, W - 1] & pad).any()
    density = float(sum(int((feas[i:i + 4096] != 0).sum()) for i in range(0, P, 4096))) / (P * W)
    assert density > 0.2, "mask is almost empty: the workload is degenerate"

    # every row == oracle, word for word (feasible, fit) and every binding
    lab = c.node_labels
    tnt = c.node_taints if preds & TAINT else None
    for lo in range(0, P, CHUNK_ROWS):
        hi = min(P, lo + CHUNK_ROWS)
        o_feas, o_fit, o_bind = capi.eval_encoded(c.avail_cpu, c.avail_mem, lab, tnt, c.req_cpu[lo:hi], c.req_mem[lo:hi],
                                                  np.ascontiguousarray(c.pod_sel[:, lo:hi]), c.pod_tol[lo:hi] if preds & TAINT else None,
                                                  np.ascontiguousarray(c.samples[lo:hi]), flags)
        assert np.array_equal(feas[lo:hi].contiguous().cpu().numpy().view(np.uint64), o_feas), f"feasible != oracle in rows [{lo}, {hi})"
        assert np.array_equal(fit[lo:hi].contiguous().cpu().numpy().view(np.uint64), o_fit), f"fit != oracle in rows [{lo}, {hi})"
        assert np.array_equal(bind[lo:hi].cpu().numpy(), o_bind), f"bindings != oracle in rows [{lo}, {hi})"
        del o_feas, o_fit, o_bind

    # bindings consistent with the mask, for every pod
    b = bind.to(torch.int64)
    has = b >= 0
    bc = b.clamp(min=0)
    word = feas.gather(1, (bc >> 6).unsqueeze(1)).squeeze(1)
    bit = (word >> (bc & 63)) & 1
    assert bool((bit[has] == 1).all()), "a bound node's bit is clear"
    assert bool((b < N).all())
    if pick == PICK_SAMPLED:
        s = d_smp.to(torch.int64)  # [P, attempts]
        ok = s < N
        sc = s.clamp(max=N - 1)
        bits = ((feas.gather(1, sc >> 6) >> (sc & 63)) & 1).bool() & ok  # feasibility of every draw
        first = torch.where(bits.any(dim=1), bits.int().argmax(dim=1), torch.full((P,), -1, device=dev))
        want = torch.where(first >= 0, s.gather(1, first.clamp(min=0).unsqueeze(1)).squeeze(1), torch.full((P,), -1, device=dev))
        assert torch.equal(want, b), "sampled pick is not the first feasible draw (src/main.rs:53-66)"
    else:
        # no feasible node <=> -1
        nonempty = torch.cat([(feas[i:i + 8192] != 0).any(dim=1) for i in range(0, P, 8192)])
        assert torch.equal(nonempty, has)

    # pod-permutation equivariance of the checksum of row checksums
    cs = _row_checksums(feas)
    perm = torch.from_numpy(np.random.default_rng(7).permutation(P)).to(dev)
    feas_p, _, bind_p = run("fused", d_cpu[perm].contiguous(), d_mem[perm].contiguous(), d_sel[:, perm].contiguous(),
                            d_tol[perm].contiguous() if d_tol is not None else None, d_smp[perm].contiguous())
    assert torch.equal(_row_checksums(feas_p), cs[perm])
    assert torch.equal(bind_p, bind[perm])
    assert int(_row_checksums(feas_p).sum()) == int(cs.sum())
    ev.set_kernel("auto")


@pytest.mark.parametrize("name", ["C3", "C4s"])
def test_full_size_the_variant_the_bench_times(evaluator, name):
    """`test_full_size_properties` always asks for the fit mask too, so at full size it exercises the stand-alone pick + the
    two-mask kernel -- not the launch bench.py times.  This is that launch: ONE mask, pitched rows, the sampled pick riding in the fused
    kernel as tile tests (`ksched_last_pick` == "fused-tile": C3 has 5 tiles, the C4 shard 10), at 100k x 5k and 125k x 10k.  Every
    feasible word and every binding == the oracle, twice in a row into two different buffers (the accumulators of the tile-test pick must
    be back at zero after a launch)."""
    import torch
    cfg, P, N, preds, pick = CASES[name]
    c = synth.make_config(cfg, P=P, N=N)
    ev = evaluator
    dev = torch.device("cuda", ev.device)
    ev.set_nodes(**c.node_columns())
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d_cpu, d_mem, d_sel, d_smp = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), t(c.samples, np.int32)
    flags = preds | pick  # no WANT_FIT_MASK
    ev.set_kernel("auto")
    runs = []
    for _ in range(2):
        feas = ev.alloc_mask(P, pitched=True)
        bind = torch.full((P,), -7, dtype=torch.int32, device=dev)
        ev.eval_device(d_cpu, d_mem, d_sel, None, d_smp, flags, out_feasible=feas, out_binding=bind)
        torch.cuda.synchronize()
        assert ev.last_kernel == "fused" and ev.last_pick == "fused-tile", (ev.last_kernel, ev.last_pick)
        runs.append((feas, bind))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    feas, bind = runs[1]
    for lo in range(0, P, CHUNK_ROWS):
        hi = min(P, lo + CHUNK_ROWS)
        o_feas, _, o_bind = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, c.req_cpu[lo:hi], c.req_mem[lo:hi],
                                              np.ascontiguousarray(c.pod_sel[:, lo:hi]), None, np.ascontiguousarray(c.samples[lo:hi]), flags)
        assert np.array_equal(feas[lo:hi].contiguous().cpu().numpy().view(np.uint64), o_feas), f"feasible != oracle in rows [{lo}, {hi})"
        assert np.array_equal(bind[lo:hi].cpu().numpy(), o_bind), f"bindings != oracle in rows [{lo}, {hi})"


@pytest.mark.parametrize("name", ["C3", "C4s"])
def test_operand_prefetch_on_and_off_write_the_same_words(evaluator, name):
    """the operand prefetch (kernels_fused.hpp `prefetch_rounds`) issues LDS-DMA loads whose destination comes from M0, written by the
    same inline-asm statement; a load that read a stale M0 would overwrite 256 bytes of the staged tile index and some mask words would come out
    wrong.  The launch the bench times (pick riding as tile tests), with the prefetch (default) and without it (KSCHED_OPT_DEBUG bit 0x20000000),
    six times each over different pod orders: every word and every binding identical between the two, and the first pair == the oracle."""
    import torch
    from kube_scheduler_rs_reference_amd import _lib
    cfg, P, N, preds, pick = CASES[name]
    c = synth.make_config(cfg, P=P, N=N)
    ev = evaluator
    dev = torch.device("cuda", ev.device)
    ev.set_nodes(**c.node_columns())
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d_cpu, d_mem, d_sel, d_smp = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), t(c.samples, np.int32)
    flags = preds | pick
    ev.set_kernel("auto")
    rng = np.random.default_rng(11)
    try:
        for rep in range(6):
            if rep == 0:
                cols = (d_cpu, d_mem, d_sel, d_smp)
            else:
                perm = torch.from_numpy(rng.permutation(P)).to(dev)
                cols = (d_cpu[perm].contiguous(), d_mem[perm].contiguous(), d_sel[:, perm].contiguous(), d_smp[perm].contiguous())
            out = []
            for dbg in (0, 0x20000000):
                ev.set_option(_lib.OPT_DEBUG, dbg)
                feas = ev.alloc_mask(P, pitched=True)
                bind = torch.full((P,), -7, dtype=torch.int32, device=dev)
                ev.eval_device(cols[0], cols[1], cols[2], None, cols[3], flags, out_feasible=feas, out_binding=bind)
                torch.cuda.synchronize()
                assert ev.last_kernel == "fused" and ev.last_pick == "fused-tile", (ev.last_kernel, ev.last_pick)
                out.append((feas, bind))
            assert torch.equal(out[0][0], out[1][0]), f"repeat {rep}: the mask differs with the operand prefetch on / off"
            assert torch.equal(out[0][1], out[1][1]), f"repeat {rep}: the bindings differ with the operand prefetch on / off"
            if rep == 0:
                feas, bind = out[0]
                for lo in range(0, P, CHUNK_ROWS):
                    hi = min(P, lo + CHUNK_ROWS)
                    o_feas, _, o_bind = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, c.req_cpu[lo:hi], c.req_mem[lo:hi],
                                                          np.ascontiguousarray(c.pod_sel[:, lo:hi]), None, np.ascontiguousarray(c.samples[lo:hi]), flags)
                    assert np.array_equal(feas[lo:hi].contiguous().cpu().numpy().view(np.uint64), o_feas), f"feasible != oracle in rows [{lo}, {hi})"
                    assert np.array_equal(bind[lo:hi].cpu().numpy(), o_bind), f"bindings != oracle in rows [{lo}, {hi})"
            del out
    finally:
        ev.set_option(_lib.OPT_DEBUG, 0)


@pytest.mark.parametrize("name,P", [("C3", 100_000), ("C4s", 125_000), ("C5s", 125_000), ("C3", 1), ("C3", 63), ("C3", 65), ("C3", 8_191), ("C4s", 12_345), ("C5s", 20_001)])
def test_round_orders_write_the_same_words(evaluator, name, P):
    """KSCHED_OPT_ROUND_ORDER (round 6): the fused kernel's waves take the batch's rounds interleaved (wave-major: the default; chunk-major) or as
    one contiguous pod range each (blocked, the order of rounds 1 - 5).  Same words, same bindings, whatever the order -- full BASELINE sizes and
    ragged ones (fewer rounds than streams, a short last round); the default order == the oracle on the ragged sizes here (the full sizes are
    compared with the oracle word for word by the tests above, which run the default order)."""
    import torch
    from kube_scheduler_rs_reference_amd import _lib
    cfg, _, N, preds, pick = CASES[name]
    c = synth.make_config(cfg, P=P, N=N)
    ev = evaluator
    dev = torch.device("cuda", ev.device)
    ev.set_nodes(**c.node_columns())
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d_cpu, d_mem, d_sel = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32)
    d_tol = t(c.pod_tol, np.int64) if preds & TAINT else None
    d_smp = t(c.samples, np.int32) if pick == PICK_SAMPLED else None
    flags = preds | pick
    ev.set_kernel("fused")
    out = {}
    try:
        for order in (0, 1, 2):
            ev.set_option(_lib.OPT_ROUND_ORDER, order)
            feas = ev.alloc_mask(P, pitched=True)
            bind = torch.full((P,), -7, dtype=torch.int32, device=dev)
            ev.eval_device(d_cpu, d_mem, d_sel, d_tol, d_smp, flags, out_feasible=feas, out_binding=bind)
            torch.cuda.synchronize()
            assert ev.last_kernel == "fused"
            out[order] = (feas, bind)
    finally:
        ev.set_option(_lib.OPT_ROUND_ORDER, 0)
        ev.set_kernel("auto")
    for order in (1, 2):
        assert torch.equal(out[0][0], out[order][0]), f"round order {order}: the mask differs from the default order's"
        assert torch.equal(out[0][1], out[order][1]), f"round order {order}: the bindings differ from the default order's"
    if P <= 20_001:
        o_feas, _, o_bind = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, c.node_taints if preds & TAINT else None, c.req_cpu, c.req_mem,
                                              np.ascontiguousarray(c.pod_sel), c.pod_tol if preds & TAINT else None,
                                              np.ascontiguousarray(c.samples) if pick == PICK_SAMPLED else None, flags)
        assert np.array_equal(out[0][0].contiguous().cpu().numpy().view(np.uint64), o_feas)
        assert np.array_equal(out[0][1].cpu().numpy(), o_bind)


@pytest.mark.parametrize("name,P", [("C3", 100_000), ("C3", 52_225), ("C3", 104_448), ("C3", 60_001), ("C4s", 40_000), ("C3", 19_999)])
def test_chunk_count_rule_writes_the_same_words(evaluator, name, P):
    """Round 6, `run_fused`: a launch whose waves take exactly two rounds each uses the SMALLEST chunk count that still needs no third round
    (C3: 49 chunks instead of the 51 that fit) -- fewer blocks filled for the same two rounds.  KSCHED_OPT_DEBUG bit 31 switches the rule off
    (the largest resident chunk count: the A/B of `profiles/r06_r7i_r7k_chunk_count.txt`).  Sizes on both sides of the rule's edges (one round
    per wave <-> two <-> three at C3: 816 streams of 64 pods), with the pick riding: every word and binding identical with and without the rule,
    and == the oracle on the smaller ones (the full size is compared with the oracle by the tests above, which run the rule)."""
    import torch
    from kube_scheduler_rs_reference_amd import _lib
    cfg, _, N, preds, pick = CASES[name]
    c = synth.make_config(cfg, P=P, N=N)
    ev = evaluator
    dev = torch.device("cuda", ev.device)
    ev.set_nodes(**c.node_columns())
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d_cpu, d_mem, d_sel, d_smp = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), t(c.samples, np.int32)
    flags = preds | pick
    ev.set_kernel("auto")
    out = []
    try:
        for dbg in (0, 0x80000000):
            ev.set_option(_lib.OPT_DEBUG, dbg)
            feas = ev.alloc_mask(P, pitched=True)
            bind = torch.full((P,), -7, dtype=torch.int32, device=dev)
            ev.eval_device(d_cpu, d_mem, d_sel, None, d_smp, flags, out_feasible=feas, out_binding=bind)
            torch.cuda.synchronize()
            assert ev.last_kernel == "fused" and ev.last_pick == "fused-tile", (ev.last_kernel, ev.last_pick)
            out.append((feas, bind))
    finally:
        ev.set_option(_lib.OPT_DEBUG, 0)
    assert torch.equal(out[0][0], out[1][0]), "the mask differs with the chunk-count rule on / off"
    assert torch.equal(out[0][1], out[1][1]), "the bindings differ with the chunk-count rule on / off"
    if P <= 60_001:
        o_feas, _, o_bind = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, c.req_cpu, c.req_mem, np.ascontiguousarray(c.pod_sel), None,
                                              np.ascontiguousarray(c.samples), flags)
        assert np.array_equal(out[0][0].contiguous().cpu().numpy().view(np.uint64), o_feas)
        assert np.array_equal(out[0][1].cpu().numpy(), o_bind)


def test_mask_larger_than_4_gib(evaluator):
    """One evaluation whose mask does not fit 32-bit byte offsets: 700k pods x 50k nodes (C5's predicates, pitched rows of 784 words)
    = 4.39 GB of feasible mask, more than half of BASELINE.json's configs[4] on ONE GPU.  Every word and every binding == the oracle
    (rows past the 4 GiB mark included), fused == direct on the device."""
    import torch
    P, N = 700_000, 50_000
    c = synth.make_config("C5", P=P, N=N)
    ev = evaluator
    dev = torch.device("cuda", ev.device)
    ev.set_nodes(**c.node_columns())
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)  # noqa: E731
    d_cpu, d_mem, d_sel, d_tol = t(c.req_cpu, np.int64), t(c.req_mem, np.int64), t(c.pod_sel, np.int32), t(c.pod_tol, np.int64)
    flags = FIT | SEL | TAINT | PICK_BESTFIT
    out = {}
    for kernel in ("fused", "direct"):
        ev.set_kernel(kernel)
        feas = ev.alloc_mask(P, pitched=True)
        bind = torch.empty((P,), dtype=torch.int32, device=dev)
        ev.eval_device(d_cpu, d_mem, d_sel, d_tol, None, flags, out_feasible=feas, out_binding=bind)
        torch.cuda.synchronize()
        assert ev.last_kernel == kernel
        out[kernel] = (feas, bind)
    ev.set_kernel("auto")
    feas, bind = out["fused"]
    assert feas.stride(0) * 8 * P > (1 << 32), "the mask must cross the 4 GiB mark for this test to mean anything"
    for lo in range(0, P, 65536):  # (chunked: torch.equal on strided 4 GB views would materialise copies)
        assert torch.equal(feas[lo:lo + 65536], out["direct"][0][lo:lo + 65536]), f"fused != direct in rows from {lo}"
    assert torch.equal(bind, out["direct"][1])
    del out
    for lo in range(0, P, 4 * CHUNK_ROWS):
        hi = min(P, lo + 4 * CHUNK_ROWS)
        o_feas, _, o_bind = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, c.node_taints, c.req_cpu[lo:hi], c.req_mem[lo:hi],
                                              np.ascontiguousarray(c.pod_sel[:, lo:hi]), c.pod_tol[lo:hi], None, flags)
        assert np.array_equal(feas[lo:hi].contiguous().cpu().numpy().view(np.uint64), o_feas), f"feasible != oracle in rows [{lo}, {hi})"
        assert np.array_equal(bind[lo:hi].cpu().numpy(), o_bind), f"bindings != oracle in rows [{lo}, {hi})"
