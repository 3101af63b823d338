#!/usr/bin/env python
"""Randomised differential test of the HIP path against the oracle (GPU box): random shapes, key counts and cardinalities (incl.
list keys and > 8 keys), taints, predicate subsets, both picks, snapshot updates between evaluations, both kernels, per-pair reasons (ksched_explain),
the two halves of ksched_eval over 1 .. 5 row shards (ksched_shard_bounds / ksched_eval_begin / ksched_eval_end) and -- with the test hooks on
(KSCHED_TEST_HOOKS=1 KSCHED_RCCL_LIB=tests/cpp/libfake_rccl.so) -- the whole multi-device sequence over 2 .. 4 evaluators on the one GPU:
ksched_comm_create_local, ksched_eval_begin on every replica, ksched_gather_buffer, ksched_allgather_bindings_local, ksched_eval_end(gathered_0).
usage: python tools/fuzz_parity.py [seconds] [seed]       prints one line per failure and a summary; exit code 1 on any failure"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kube_scheduler_rs_reference_amd import Evaluator, _lib as L
from oracle import capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ev = Evaluator(0)
# the multi-device sequence with n > 1 needs the TEST-ONLY librccl stand-in (one GPU stands for n ranks): replicas + one clique per n, made on first use
HOOKS = os.environ.get("KSCHED_TEST_HOOKS") == "1" and bool(os.environ.get("KSCHED_RCCL_LIB"))
if HOOKS and not hasattr(L.load(), "ksched_test_hooks_linked"):  # the stand-in is loadable by the TEST build of the library only (KSCHED_LIB=tests/cpp/hooks/libksched_hip.so)
    print("fuzz: KSCHED_TEST_HOOKS is set but the loaded library is the shipped one (no hooks): the n > 1 multi-device cases are left out; set KSCHED_LIB=tests/cpp/hooks/libksched_hip.so", flush=True)
    HOOKS = False
replicas, cliques = [], {}
t_end = time.time() + budget
cases = fails = 0
picks = {}
I64 = np.iinfo(np.int64)
while time.time() < t_end:
    cases += 1
    cs = int(rng.integers(0, 1 << 31))
    r = np.random.default_rng(cs)
    N = int(r.choice([1, 2, 63, 64, 65, 300, 1023, 1024, 1025, 2500, 4097, 6000]))
    P = int(r.choice([1, 7, 64, 65, 500, 1500, 3000]))
    if r.random() < 0.05:  # now and then a batch long enough for two or more rounds per wave of a whole-chip launch (run_fused's chunk-count rule, the riding pick's longer forms)
        P = int(r.choice([12_000, 20_011, 40_000, 52_225]))
    K = int(r.choice([0, 1, 3, 8, 9, 12]))
    cards = [int(r.choice([1, 2, 5, 40, 300, N, 4 * N + 7])) for _ in range(K)]
    scale = int(r.choice([10, 1000, 1 << 20]))
    cpu = r.integers(-scale, 64 * scale, N).astype(np.int64)
    mem = r.integers(-scale, 64 * scale, N).astype(np.int64)
    if r.random() < 0.1:
        cpu[r.integers(0, N, 3)] = r.choice([I64.min, I64.max, 0])
    lab = np.stack([r.integers(0, c + 1, N).astype(np.uint32) for c in cards]) if K else None
    nt = int(r.choice([0, 0, 3, 16, 60]))
    taints = (r.integers(0, 1 << nt, N).astype(np.uint64) & r.integers(0, 1 << nt, N).astype(np.uint64)) if nt else None
    rc = r.integers(-5, 70 * scale, P).astype(np.int64)
    rm = r.integers(-5, 70 * scale, P).astype(np.int64)
    pk = float(r.choice([0.05, 0.3, 0.9]))
    sel = np.stack([np.where(r.random(P) < pk, r.integers(1, c + 3, P), 0).astype(np.uint32) for c in cards]) if K else None
    if K and r.random() < 0.5:
        sel[r.integers(0, K), r.integers(0, P, max(1, P // 20))] = L.SEL_NEVER
    tol = r.integers(0, 1 << nt, P).astype(np.uint64) if nt else None
    smp = r.integers(0, N + (2 if r.random() < 0.2 else 0), (P, 5)).astype(np.uint32)
    preds = int(r.choice([L.FIT, L.FIT | L.SEL, L.FIT | L.SEL | L.TAINT, L.SEL, L.SEL | L.TAINT, L.FIT | L.TAINT]))
    if not K:
        preds &= ~L.SEL
    if not nt:
        preds &= ~L.TAINT
    if preds == 0:
        preds = L.FIT
    pick = int(r.choice([0, L.PICK_SAMPLED, L.PICK_BESTFIT]))
    flags = preds | pick | (L.WANT_FIT_MASK if r.random() < 0.5 else 0)
    try:
        ev.set_option(L.OPT_BESTFIT_STAGES, int(r.choice([0, 1, 2])))
        ev.set_option(L.OPT_GRID_CUS, int(r.choice([0, 0, 0, 8, 17, 96, 200])))  # fewer compute units per launch: same results
        ev.set_option(L.OPT_ROUND_ORDER, int(r.choice([0, 0, 1, 2])))  # which wave takes which round: same results
        ev.set_option(L.OPT_FUSED_PICK, int(r.choice([0, 1, 1, 2])))  # 3 (tile tests or E_UNSUPPORTED) below, where it applies
        ev.set_nodes(cpu, mem, lab, taints)
        for step in range(int(r.choice([1, 1, 3]))):
            if step:  # a snapshot update between evaluations
                idx = r.integers(0, N, int(r.choice([1, 5, 40, N]))).astype(np.uint32)
                nc, nm = r.integers(-scale, 64 * scale, idx.size).astype(np.int64), r.integers(-scale, 64 * scale, idx.size).astype(np.int64)
                ev.update_nodes(idx, nc, nm)
                for j in range(idx.size):
                    cpu[idx[j]], mem[idx[j]] = nc[j], nm[j]
            want = capi.eval_encoded(cpu, mem, lab, taints if (preds & L.TAINT) else None, rc, rm, sel, tol if (preds & L.TAINT) else None, smp, flags)
            for kernel in ("auto", "direct"):
                ev.set_kernel(kernel)
                got = ev.eval(rc, rm, sel if K else None, tol if (preds & L.TAINT) else None, smp if pick == L.PICK_SAMPLED else None, flags)
                ok = np.array_equal(got.feasible, want[0]) and (not (flags & L.WANT_FIT_MASK) or np.array_equal(got.fit, want[1])) and \
                    (not pick or np.array_equal(got.binding, want[2]))
                if not ok:
                    fails += 1
                    print(f"FAIL case seed {cs}: N={N} P={P} K={K} cards={cards} nt={nt} flags={flags:#x} kernel={kernel}/{ev.last_kernel} pick={ev.last_pick} step={step}", flush=True)
                picks[ev.last_pick] = picks.get(ev.last_pick, 0) + 1
            if pick == L.PICK_SAMPLED and K <= 8 and not (preds & L.TAINT):  # the tile-test form of the riding pick, forced
                ev.set_kernel("fused")
                ev.set_option(L.OPT_FUSED_PICK, 3)
                try:
                    got = ev.eval(rc, rm, sel if K else None, None, smp, flags)
                    if not (np.array_equal(got.feasible, want[0]) and np.array_equal(got.binding, want[2])):
                        fails += 1
                        print(f"FAIL tile pick case seed {cs}: N={N} P={P} K={K} cards={cards} flags={flags:#x} pick={ev.last_pick} step={step}", flush=True)
                    picks[ev.last_pick] = picks.get(ev.last_pick, 0) + 1
                except L.KschedError as e:
                    if e.code != L.E_UNSUPPORTED:
                        raise
                    picks["tile-unsupported"] = picks.get("tile-unsupported", 0) + 1
                ev.set_option(L.OPT_FUSED_PICK, 1)
        ev.set_kernel("auto")
        if pick and r.random() < 0.4:
            # the pick alone from HOST masks (ksched_pick, what a selector evaluated in key groups uses): from the oracle's mask it is the oracle's
            # binding; from that mask thinned by random words (an AND with another group's mask) the sampled pick is the first draw whose bit is set
            got_b = ev.pick(want[0], pick | (preds & L.FIT), req_mem_bytes=rm if (preds & L.FIT) else None, samples=smp if pick == L.PICK_SAMPLED else None)
            ok = np.array_equal(got_b, want[2])
            if ok and pick == L.PICK_SAMPLED:
                thin = want[0] & r.integers(0, 1 << 63, want[0].shape, dtype=np.uint64)
                got_t = ev.pick(thin, L.PICK_SAMPLED, samples=smp)
                bit = lambda row, n: n < N and bool((int(thin[row, n >> 6]) >> (n & 63)) & 1)  # noqa: E731
                want_t = np.array([next((int(x) for x in smp[i] if bit(i, int(x))), -1) for i in range(P)], dtype=np.int32)
                ok = np.array_equal(got_t, want_t)
            if not ok:
                fails += 1
                print(f"FAIL ksched_pick case seed {cs}: N={N} P={P} K={K} nt={nt} flags={flags:#x}", flush=True)
            picks["host-masks"] = picks.get("host-masks", 0) + 1
        if pick and r.random() < 0.35:
            # the host-side row shard (include/ksched.h "one host thread, several devices"): the batch cut into 1 .. 5 shards with
            # ksched_shard_bounds, every shard through ksched_eval_begin (selector columns addressed inside the whole batch's array with its
            # stride, bindings padded to ceil(P / n) with -1) + ksched_eval_end, merged like an all-gathered table -- on this one evaluator,
            # shard after shard
            import ctypes as C
            n_sh = int(r.choice([1, 2, 3, 5]))
            W = ev.W
            lo_, hi_, cpr_ = C.c_uint32(), C.c_uint32(), C.c_uint32()
            rc_c, rm_c = np.ascontiguousarray(rc), np.ascontiguousarray(rm)
            sel_c = np.ascontiguousarray(sel) if K else None
            tol_c = np.ascontiguousarray(tol) if (preds & L.TAINT) else None
            smp_c = np.ascontiguousarray(smp)
            feas_s = np.zeros((P, W), dtype=np.uint64)
            fit_s = np.zeros((P, W), dtype=np.uint64)
            bind_s = np.full((P,), 777, dtype=np.int32)
            for rank in range(n_sh):
                ev._lib.ksched_shard_bounds(P, n_sh, rank, C.byref(lo_), C.byref(hi_), C.byref(cpr_))
                lo, hi, cpr = lo_.value, hi_.value, cpr_.value
                dev_b, stream = C.c_void_p(), C.c_void_p()
                rcode = ev._lib.ksched_eval_begin(
                    ev._h, hi - lo, C.c_void_p(rc_c.ctypes.data + 8 * lo), C.c_void_p(rm_c.ctypes.data + 8 * lo),
                    C.c_void_p(sel_c.ctypes.data + 4 * lo) if K else None, P, C.c_void_p(tol_c.ctypes.data + 8 * lo) if tol_c is not None else None,
                    C.c_void_p(smp_c.ctypes.data + 20 * lo) if pick == L.PICK_SAMPLED else None, 5 if pick == L.PICK_SAMPLED else 0, flags,
                    C.c_void_p(feas_s.ctypes.data + 8 * W * lo), C.c_void_p(fit_s.ctypes.data + 8 * W * lo) if flags & L.WANT_FIT_MASK else None, cpr,
                    C.byref(dev_b), C.byref(stream))
                if rcode != 0:
                    raise L.KschedError(rcode, "ksched_eval_begin", ev._lib.ksched_last_error(ev._h).decode())
                part = np.full((cpr,), 555, dtype=np.int32)
                rcode = ev._lib.ksched_eval_end(ev._h, dev_b, cpr, part.ctypes.data_as(C.c_void_p))
                if rcode != 0:
                    raise L.KschedError(rcode, "ksched_eval_end", ev._lib.ksched_last_error(ev._h).decode())
                bind_s[lo:hi] = part[:hi - lo]
                if not (part[hi - lo:] == -1).all():
                    fails += 1
                    print(f"FAIL shard padding case seed {cs}: P={P} shards={n_sh} rank={rank}", flush=True)
            if not (np.array_equal(feas_s, want[0]) and (not (flags & L.WANT_FIT_MASK) or np.array_equal(fit_s, want[1])) and np.array_equal(bind_s, want[2])):
                fails += 1
                print(f"FAIL sharded halves case seed {cs}: N={N} P={P} K={K} nt={nt} flags={flags:#x} shards={n_sh}", flush=True)
            picks["sharded-halves"] = picks.get("sharded-halves", 0) + 1
        if HOOKS and pick and r.random() < 0.35:
            import ctypes as C
            n_sh = int(r.choice([2, 3, 4]))
            while len(replicas) < n_sh:
                replicas.append(Evaluator(0))
            reps = replicas[:n_sh]
            lib = ev._lib
            if n_sh not in cliques:
                ctxs = (C.c_void_p * n_sh)(*[e._h for e in reps])
                comms = (C.c_void_p * n_sh)()
                rcode = lib.ksched_comm_create_local(ctxs, n_sh, comms)
                if rcode != 0:
                    raise L.KschedError(rcode, "ksched_comm_create_local", lib.ksched_comm_last_error().decode())
                cliques[n_sh] = comms
            comms = cliques[n_sh]
            for e in reps:  # the snapshot is replicated (the current values: the updates above are in cpu / mem)
                e.set_option(L.OPT_BESTFIT_STAGES, int(r.choice([0, 1, 2])))
                e.set_nodes(cpu, mem, lab, taints)
            W = ev.W
            lo_, hi_, cpr_ = C.c_uint32(), C.c_uint32(), C.c_uint32()
            rc_c, rm_c = np.ascontiguousarray(rc), np.ascontiguousarray(rm)
            sel_c = np.ascontiguousarray(sel) if K else None
            tol_c = np.ascontiguousarray(tol) if (preds & L.TAINT) else None
            smp_c = np.ascontiguousarray(smp)
            feas_s, fit_s = np.zeros((P, W), dtype=np.uint64), np.zeros((P, W), dtype=np.uint64)
            local, gathered, streams = (C.c_void_p * n_sh)(), (C.c_void_p * n_sh)(), (C.c_void_p * n_sh)()
            lib.ksched_shard_bounds(P, n_sh, 0, C.byref(lo_), C.byref(hi_), C.byref(cpr_))
            cpr = cpr_.value
            for rank, e in enumerate(reps):
                lib.ksched_shard_bounds(P, n_sh, rank, C.byref(lo_), C.byref(hi_), C.byref(cpr_))
                lo, hi = lo_.value, hi_.value
                dev_b, stream = C.c_void_p(), C.c_void_p()
                rcode = lib.ksched_eval_begin(
                    e._h, hi - lo, C.c_void_p(rc_c.ctypes.data + 8 * lo), C.c_void_p(rm_c.ctypes.data + 8 * lo),
                    C.c_void_p(sel_c.ctypes.data + 4 * lo) if K else None, P, C.c_void_p(tol_c.ctypes.data + 8 * lo) if tol_c is not None else None,
                    C.c_void_p(smp_c.ctypes.data + 20 * lo) if pick == L.PICK_SAMPLED else None, 5 if pick == L.PICK_SAMPLED else 0, flags,
                    C.c_void_p(feas_s.ctypes.data + 8 * W * lo), C.c_void_p(fit_s.ctypes.data + 8 * W * lo) if flags & L.WANT_FIT_MASK else None, cpr,
                    C.byref(dev_b), C.byref(stream))
                if rcode != 0:
                    raise L.KschedError(rcode, "ksched_eval_begin", lib.ksched_last_error(e._h).decode())
                local[rank], streams[rank] = dev_b.value, stream.value
                g = C.c_void_p()
                rcode = lib.ksched_gather_buffer(e._h, n_sh * cpr, C.byref(g))
                if rcode != 0:
                    raise L.KschedError(rcode, "ksched_gather_buffer", lib.ksched_last_error(e._h).decode())
                gathered[rank] = g.value
            rcode = lib.ksched_allgather_bindings_local(comms, n_sh, local, gathered, cpr, streams)
            if rcode != 0:
                raise L.KschedError(rcode, "ksched_allgather_bindings_local", lib.ksched_comm_last_error().decode())
            table = np.full((n_sh * cpr,), 555, dtype=np.int32)
            for rank, e in enumerate(reps):
                rcode = lib.ksched_eval_end(e._h, C.c_void_p(gathered[0]) if rank == 0 else None, n_sh * cpr if rank == 0 else 0,
                                            table.ctypes.data_as(C.c_void_p) if rank == 0 else None)
                if rcode != 0:
                    raise L.KschedError(rcode, "ksched_eval_end", lib.ksched_last_error(e._h).decode())
            bind_s = np.full((P,), 777, dtype=np.int32)
            pad_ok = True
            for rank in range(n_sh):
                lib.ksched_shard_bounds(P, n_sh, rank, C.byref(lo_), C.byref(hi_), C.byref(cpr_))
                bind_s[lo_.value:hi_.value] = table[rank * cpr: rank * cpr + hi_.value - lo_.value]
                pad_ok = pad_ok and bool((table[rank * cpr + hi_.value - lo_.value: (rank + 1) * cpr] == -1).all())
            if not (pad_ok and np.array_equal(feas_s, want[0]) and (not (flags & L.WANT_FIT_MASK) or np.array_equal(fit_s, want[1])) and np.array_equal(bind_s, want[2])):
                fails += 1
                print(f"FAIL multi-device sequence case seed {cs}: N={N} P={P} K={K} nt={nt} flags={flags:#x} replicas={n_sh}", flush=True)
            picks[f"gathered-over-{n_sh}"] = picks.get(f"gathered-over-{n_sh}", 0) + 1
        if r.random() < 0.3:  # ksched_explain on random pairs == the reason rebuilt from three single-predicate oracle masks
            from kube_scheduler_rs_reference_amd.evaluator import unpack_mask
            one = lambda f: unpack_mask(capi.eval_encoded(cpu, mem, lab, taints, rc, rm, sel, tol, None, f)[0], N)  # noqa: E731
            ok_f = one(L.FIT) if preds & L.FIT else np.ones((P, N), bool)
            ok_s = one(L.SEL) if preds & L.SEL else np.ones((P, N), bool)
            ok_t = one(L.TAINT) if preds & L.TAINT else np.ones((P, N), bool)
            want_r = np.where(~ok_f, L.REASON_NOT_ENOUGH_RESOURCES, np.where(~ok_s, L.REASON_NODE_SELECTOR_MISMATCH, np.where(~ok_t, L.REASON_TAINT_NOT_TOLERATED, L.REASON_OK)))
            pp, pn = r.integers(0, P, 4000).astype(np.uint32), r.integers(0, N, 4000).astype(np.uint32)
            got_r = ev.explain(rc, rm, sel if K else None, tol if (preds & L.TAINT) else None, pp, pn, preds)
            if not np.array_equal(got_r, want_r[pp, pn]):
                fails += 1
                print(f"FAIL explain case seed {cs}: N={N} P={P} K={K} cards={cards} nt={nt} preds={preds:#x}", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"EXCEPTION case seed {cs}: N={N} P={P} K={K} cards={cards} nt={nt} flags={flags:#x}: {e}", flush=True)
        ev.set_kernel("auto")
print(f"fuzz: {cases} cases, {fails} failures, seed {seed}, pick launches {picks}")
sys.exit(1 if fails else 0)
