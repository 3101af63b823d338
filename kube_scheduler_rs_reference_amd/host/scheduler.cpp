#include "scheduler.hpp"

#include "pool.hpp"

#include <chrono>
#include <exception>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <thread>
#include <tuple>

namespace ksched_host {

std::optional<size_t> SplitMixChooser::choose(size_t n) {
    if (n == 0) return std::nullopt;
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (size_t)(z % n);
}

std::optional<size_t> ScriptedChooser::choose(size_t n) {
    if (n == 0 || next >= script.size()) return std::nullopt;
    return script[next++] % n;
}

const char *error_text(ReconcileError e) {
    switch (e) {
        case ReconcileError::CreateBindingFailed: return "create-binding-failed";
        case ReconcileError::CreateBindingObjectFailed: return "create-binding-object-failed";
        case ReconcileError::NoNodeFound: return "no-node-found";
    }
    return "?";
}

std::optional<corev1::Node> select_node_for_pod(const corev1::Pod &pod, Context &ctx, NodeChooser &chooser,
                                                std::vector<RejectedCandidate> *rejected) {
    std::optional<corev1::Node> node;
    for (uint32_t attempt = 0; attempt < ATTEMPTS; ++attempt) {  // src/main.rs:53
        // ctx.node_store.state().choose(&mut rng): a uniform draw with replacement; an empty store yields None (:56)
        const std::optional<size_t> idx = chooser.choose(ctx.node_store.size());
        if (!idx) continue;
        const corev1::Node &candidate = ctx.node_store[*idx];
        const predicates::Validity v = predicates::check_node_validity(pod, candidate, ctx);  // :61
        if (v) {
            const RejectedCandidate r{corev1::name_any(candidate.metadata), *v};
            if (ctx.warn) ctx.warn(rejected_line(pod, r));  // the WARN line at :62
            if (rejected) rejected->push_back(r);
        } else {
            node = candidate;  // :64-65
            break;
        }
    }
    return node;  // :70
}

BatchSelection select_nodes_for_pods(const std::vector<const corev1::Pod *> &pods, Context &ctx, NodeChooser &chooser, bool want_rejected) {
    if (!ctx.snapshot) ctx.refresh_snapshot();
    Snapshot &snap = *ctx.snapshot;
    const uint32_t p = (uint32_t)pods.size(), n = snap.n();
    BatchSelection out;
    out.node_store_index.assign(p, -1);
    if (want_rejected) out.rejected.resize(p);
    // the draws, in the reference's order: pod by pod, attempt by attempt, over the store's own ordering;
    // converted to canonical column indices for the device (an empty store gives "no draw" = index n)
    PhaseClock clock("select");
    std::vector<uint32_t> samples((size_t)p * ATTEMPTS, n);
    auto draw = [&] {
        const size_t store = ctx.node_store.size();
        for (uint32_t i = 0; i < p; ++i)
            for (uint32_t t = 0; t < ATTEMPTS; ++t) {
                const std::optional<size_t> idx = chooser.choose(store);
                if (idx) samples[(size_t)i * ATTEMPTS + t] = snap.canonical_index((uint32_t)*idx);
            }
    };
    // A large batch draws on a thread of its own while this one plans and encodes the batch (the draws are a serial walk -- one
    // generator, the reference's order -- of 2 ms per 100 000 pods; the device call is the first reader).  Joined on every path out.
    std::thread drawer;
    std::exception_ptr draw_error;
    struct Join {
        std::thread &t;
        ~Join() {
            if (t.joinable()) t.join();
        }
    } join_guard{drawer};
    const std::function<void()> samples_ready = [&] {
        if (drawer.joinable()) drawer.join();
        if (draw_error) std::rethrow_exception(draw_error);
    };
    if (p >= 4096 && n != 0) {
        drawer = std::thread([&] {
            try {
                draw();
            } catch (...) {
                draw_error = std::current_exception();
            }
        });
    } else {
        draw();
    }
    clock.lap("draws (a large batch: on their own thread from here)");
    if (n == 0 || p == 0) {
        out.samples = samples;
        return out;
    }
    // (the masks only when the caller wants the rejected draws' reasons: a batch's bindings alone need no mask kernel and no mask copy)
    out.validity = predicates::check_node_validity_batch(pods, ctx, /*taints=*/false, KSCHED_PICK_SAMPLED, &samples, ATTEMPTS, /*want_masks=*/want_rejected, &samples_ready);
    samples_ready();
    out.samples = samples;
    clock.lap("check_node_validity_batch");
    for (uint32_t i = 0; i < p; ++i) {
        const int32_t b = out.validity.binding[i];
        if (b >= 0) out.node_store_index[i] = (int32_t)snap.store_index((uint32_t)b);
        if (want_rejected) {
            for (uint32_t t = 0; t < ATTEMPTS; ++t) {
                const uint32_t s = samples[(size_t)i * ATTEMPTS + t];
                if (s >= n) continue;
                const predicates::Validity v = out.validity.validity(i, s);
                if (!v) break;  // this draw won
                out.rejected[i].push_back({snap.columns().names[s], *v});
            }
        }
    }
    return out;
}

std::vector<std::vector<RejectedCandidate>> explain_rejected(const std::vector<const corev1::Pod *> &pods, Context &ctx, const BatchSelection &sel) {
    const uint32_t p = (uint32_t)pods.size();
    std::vector<std::vector<RejectedCandidate>> out(p);
    if (!ctx.snapshot || p == 0 || sel.samples.size() != (size_t)p * ATTEMPTS || sel.validity.binding.size() != p) return out;
    Snapshot &snap = *ctx.snapshot;
    const uint32_t n = snap.n();
    std::vector<std::pair<uint32_t, uint32_t>> pairs;
    for (uint32_t i = 0; i < p; ++i) {
        const int32_t won = sel.validity.binding[i];
        for (uint32_t t = 0; t < ATTEMPTS; ++t) {
            const uint32_t s = sel.samples[(size_t)i * ATTEMPTS + t];
            if (s >= n) continue;                       // no draw (empty store)
            if (won >= 0 && s == (uint32_t)won) break;  // the first feasible draw wins (src/main.rs:61-65): everything before it was refused
            pairs.emplace_back(i, s);
        }
    }
    const std::vector<predicates::Validity> why = predicates::explain_pairs(pods, ctx, pairs);
    for (size_t k = 0; k < pairs.size(); ++k)
        if (why[k]) out[pairs[k].first].push_back({snap.columns().names[pairs[k].second], *why[k]});
    return out;
}

std::string rejected_line(const corev1::Pod &pod, const RejectedCandidate &r) {
    return "Node " + r.node_name + " failed validity check for pod " + full_name(pod.metadata) + ": " + predicates::debug_name(r.reason);
}

void warn_rejected(const std::vector<const corev1::Pod *> &pods, Context &ctx, const BatchSelection &sel) {
    if (!ctx.warn) return;  // the level is off: a quiet cluster pays nothing
    const auto rejected = sel.rejected.size() == pods.size() ? sel.rejected : explain_rejected(pods, ctx, sel);
    for (size_t i = 0; i < pods.size(); ++i)
        for (const RejectedCandidate &r : rejected[i]) ctx.warn(rejected_line(*pods[i], r));
}

const char *debug_name(ReconcileError e) {
    switch (e) {
        case ReconcileError::CreateBindingFailed: return "CreateBindingFailed";
        case ReconcileError::CreateBindingObjectFailed: return "CreateBindingObjectFailed";
        case ReconcileError::NoNodeFound: return "NoNodeFound";
    }
    return "?";
}

Action error_policy(const corev1::Pod &, ReconcileError) { return Action::RequeueAfter5Min; }  // src/main.rs:122-125
Action error_policy(const corev1::Pod &pod, ReconcileError error, Context &ctx) {
    if (ctx.warn) ctx.warn("reconcile failed on pod " + full_name(pod.metadata) + ": " + debug_name(error));  // :123
    return error_policy(pod, error);
}

namespace {
// what the Controller does with a failed reconcile: error_policy (src/main.rs:141-144) -- here for every failed outcome of a batch, in batch order
void warn_failed(const std::vector<const corev1::Pod *> &pods, const std::vector<ReconcileOutcome> &out, Context &ctx) {
    if (!ctx.warn) return;
    for (size_t i = 0; i < pods.size() && i < out.size(); ++i)
        if (!out[i].ok) (void)error_policy(*pods[i], out[i].error, ctx);
}
}  // namespace

namespace {

ReconcileOutcome bind(const corev1::Pod &pod, const corev1::Node *chosen, BindingSink &sink) {
    ReconcileOutcome r;
    if (!chosen) {  // src/main.rs:116-118
        r.ok = false;
        r.error = ReconcileError::NoNodeFound;
        r.action = error_policy(pod, r.error);
        return r;
    }
    const std::string pod_name = corev1::name_any(pod.metadata);
    if (!pod.metadata.namespace_) {
        // `pod.namespace().unwrap()` panics in the reference (src/main.rs:80); a pod without a namespace
        // cannot come from the API server.  Map it to the binding-object failure instead of unwinding.
        r.ok = false;
        r.error = ReconcileError::CreateBindingObjectFailed;
        r.action = error_policy(pod, r.error);
        return r;
    }
    Binding b;
    b.metadata = pod.metadata;  // src/main.rs:88-91
    b.target_name = corev1::name_any(chosen->metadata);
    if (!sink.create_pod_binding(pod_name, *pod.metadata.namespace_, b)) {  // :103-108
        r.ok = false;
        r.error = ReconcileError::CreateBindingFailed;
        r.action = error_policy(pod, r.error);
        return r;
    }
    r.bound_to = b.target_name;
    return r;  // Ok(Action::await_change()), :119
}

}  // namespace

std::vector<ReconcileOutcome> post_bindings(const std::vector<const corev1::Pod *> &pods, const std::vector<const corev1::Node *> &chosen,
                                            BindingSink &sink, unsigned post_concurrency) {
    const size_t p = pods.size();
    std::vector<ReconcileOutcome> out(p);
    auto one = [&](size_t i) {
        try {
            out[i] = bind(*pods[i], i < chosen.size() ? chosen[i] : nullptr, sink);
        } catch (...) {  // a throwing sink is a failed POST (src/main.rs:105-108), never an exception crossing a worker thread
            ReconcileOutcome r;
            r.ok = false;
            r.error = ReconcileError::CreateBindingFailed;
            r.action = error_policy(*pods[i], r.error);
            out[i] = r;
        }
    };
    const size_t workers = std::min<size_t>(post_concurrency, p);
    if (workers <= 1) {
        for (size_t i = 0; i < p; ++i) one(i);
        return out;
    }
    // pods are handed out one at a time (a POST is a network round trip: no point in static ranges), each outcome slot is
    // written by exactly one worker
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    pool.reserve(workers);
    for (size_t w = 0; w < workers; ++w)
        pool.emplace_back([&] {
            for (size_t i = next.fetch_add(1, std::memory_order_relaxed); i < p; i = next.fetch_add(1, std::memory_order_relaxed)) one(i);
        });
    for (auto &t : pool) t.join();
    return out;
}

ReconcileOutcome reconcile(const corev1::Pod &pod, Context &ctx, NodeChooser &chooser, BindingSink &sink) {
    if (is_pod_bound(pod)) return ReconcileOutcome{};  // src/main.rs:74-76
    const std::optional<corev1::Node> chosen = select_node_for_pod(pod, ctx, chooser);
    const ReconcileOutcome r = bind(pod, chosen ? &*chosen : nullptr, sink);
    if (!r.ok) (void)error_policy(pod, r.error, ctx);  // (the Controller's error_policy call: the WARN line of :123)
    return r;
}

std::vector<ReconcileOutcome> reconcile_batch(const std::vector<const corev1::Pod *> &pods, Context &ctx, NodeChooser &chooser,
                                              BindingSink &sink, unsigned post_concurrency) {
    PhaseClock clock("reconcile_batch");
    std::vector<ReconcileOutcome> out(pods.size());
    std::vector<const corev1::Pod *> pending;
    std::vector<size_t> where;
    pending.reserve(pods.size());
    where.reserve(pods.size());
    for (size_t i = 0; i < pods.size(); ++i) {
        if (is_pod_bound(*pods[i])) continue;  // Ok(await_change) without touching the evaluator
        pending.push_back(pods[i]);
        where.push_back(i);
    }
    clock.lap("outcome slots + the pending pods");
    const bool timing = std::getenv("KSCHED_HOST_TIMING") != nullptr;  // (tools/host_loop.py: where a batch's host time goes, to stderr)
    const auto t0 = std::chrono::steady_clock::now();
    BatchSelection sel = select_nodes_for_pods(pending, ctx, chooser);
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<const corev1::Node *> chosen(pending.size(), nullptr);
    for (size_t j = 0; j < pending.size(); ++j)
        if (sel.node_store_index[j] >= 0) chosen[j] = &ctx.node_store[(size_t)sel.node_store_index[j]];
    // The bindings count against their nodes for the NEXT batch (the reference gets that from re-LISTing on every evaluation,
    // src/predicates.rs:34-38; here the snapshot is patched in one device update).  Everything of that update that only READS the
    // snapshot -- the pods' keys, what the bookkeeping holds for them, the per-node sums -- is worked out WHILE THE POSTS ARE IN FLIGHT
    // (they run on their own thread; the reference's are network round trips), for every pod that was given a node, from the node index
    // the device returned and the request sums the encoder made (no name lookup, no second parse).  It is committed afterwards, and only
    // if every one of those POSTs landed; otherwise it is dropped and the ones that did land are observed the plain way.
    std::shared_ptr<Snapshot::StagedUpdate> staged;
    std::vector<ReconcileOutcome> posted;
    {
        std::exception_ptr post_error;
        std::thread poster([&] {
            try {
                posted = post_bindings(pending, chosen, sink, post_concurrency);  // src/main.rs:94-108, overlapped
            } catch (...) {
                post_error = std::current_exception();
            }
        });
        try {
            if (ctx.snapshot && sel.validity.exact_requests && sel.validity.binding.size() == pending.size()) {
                std::vector<Snapshot::Bound> all;
                all.reserve(pending.size());
                for (size_t j = 0; j < pending.size(); ++j)
                    if (chosen[j] && pending[j]->metadata.namespace_)  // (a pod without a namespace is never POSTed: bind())
                        all.push_back({pending[j], (uint32_t)sel.validity.binding[j], sel.validity.req_cpu_nanos[j], sel.validity.req_mem_nanos[j]});
                if (!all.empty()) staged = ctx.snapshot->stage_bound(all);
            }
        } catch (const EncodeError &) {
            staged.reset();  // (decided after the POSTs, on the plain path)
        } catch (...) {
            poster.join();
            throw;
        }
        poster.join();
        if (post_error) std::rethrow_exception(post_error);
    }
    const auto t2 = std::chrono::steady_clock::now();
    std::vector<std::pair<const corev1::Pod *, const std::string *>> landed;  // (pod, the node it was bound to), for the snapshot update
    size_t expected = 0;
    for (size_t j = 0; j < pending.size(); ++j) {
        out[where[j]] = std::move(posted[j]);
        if (chosen[j] && pending[j]->metadata.namespace_) ++expected;
        if (out[where[j]].ok && out[where[j]].bound_to) landed.emplace_back(pending[j], &*out[where[j]].bound_to);  // (`out` is not resized any more)
    }
    // The whole batch was evaluated against ONE snapshot (the reference's racing reconciles all see the same API-server state).
    if (ctx.snapshot && !landed.empty()) {
        // (the tracked form: when the watch later echoes these bindings as MODIFIED events they change nothing)
        // The bindings EXIST by now: whatever happens to the snapshot update, the caller gets `out` (an exception here would make
        // a batching caller lose the outcomes, or worse reconcile -- and POST -- the batch again).  A device failure leaves the
        // snapshot marked stale (it uploads everything before the next evaluation); a bookkeeping failure (int64 overflow of
        // `available`) drops the snapshot, so that the next batch starts from fresh LISTs.
        try {
            // (a snapshot that changed while the POSTs ran -- a watch event applied by another caller -- takes the plain path too)
            if (staged && landed.size() == expected && ctx.snapshot->staged_is_current(*staged)) (void)ctx.snapshot->commit_staged(*staged);
            else ctx.snapshot->observe_bound(landed);
        } catch (const EncodeError &) {
            if (!ctx.snapshot->device_stale()) ctx.snapshot.reset();
        }
    }
    const auto t3 = std::chrono::steady_clock::now();
    // src/main.rs:62, for the whole batch (one ksched_explain call; skipped when nobody listens).  Behind the POSTs and outside the
    // evaluation's span: the line is a log line -- a failure to produce it (the explain call, an encode error of the re-encoded pods)
    // degrades to one warning and never costs the batch its bindings.
    try {
        warn_rejected(pending, ctx, sel);
    } catch (const std::exception &e) {
        if (ctx.warn) ctx.warn(std::string("the rejected candidates of this batch could not be listed: ") + e.what());
    }
    warn_failed(pods, out, ctx);
    if (timing) {
        const auto t4 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "reconcile_batch %zu pods: draws + encode + device %.2f ms, bindings (POST) with the snapshot update staged beside them %.2f ms, snapshot update committed %.2f ms, WARN lines %.2f ms\n",
                     pending.size(), ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4));
    }
    // (100 000 outcomes' strings, the draws, the staged update's keys: freeing them takes 4 ms of a C3-size batch's 21 -- not on the caller's thread)
    Reaper::instance().discard_later(std::make_tuple(std::move(sel), std::move(posted), std::move(staged), std::move(chosen), std::move(landed), std::move(pending), std::move(where)));
    clock.lap("everything up to the return");
    return out;
}

std::vector<ReconcileOutcome> reconcile_batch_sequential(const std::vector<const corev1::Pod *> &pods, Context &ctx, NodeChooser &chooser,
                                                         BindingSink &sink, uint32_t max_rounds, SequentialStats *stats) {
    std::vector<ReconcileOutcome> out(pods.size());
    std::vector<size_t> pending;
    for (size_t i = 0; i < pods.size(); ++i)
        if (!is_pod_bound(*pods[i])) pending.push_back(i);  // src/main.rs:74-76
    if (!ctx.snapshot) ctx.refresh_snapshot();
    SequentialStats st;
    while (!pending.empty() && st.rounds < max_rounds) {
        ++st.rounds;
        std::vector<const corev1::Pod *> batch;
        for (size_t i : pending) batch.push_back(pods[i]);
        const BatchSelection sel = select_nodes_for_pods(batch, ctx, chooser);  // one device evaluation + pick
        warn_rejected(batch, ctx, sel);
        std::vector<size_t> next;
        std::vector<bool> taken(ctx.node_store.size(), false);
        std::vector<std::pair<const corev1::Pod *, const std::string *>> landed;  // (pod, the node it was bound to), for the snapshot update
        for (size_t j = 0; j < pending.size(); ++j) {
            const size_t i = pending[j];
            const int32_t idx = sel.node_store_index[j];
            if (idx < 0) {  // no feasible draw against the current state
                out[i] = bind(*pods[i], nullptr, sink);
                continue;
            }
            if (taken[(size_t)idx]) {  // an earlier pod of this round shrank that node: look again next round
                ++st.conflicts;
                next.push_back(i);
                continue;
            }
            out[i] = bind(*pods[i], &ctx.node_store[(size_t)idx], sink);
            if (!out[i].ok) continue;  // the POST failed: nothing landed on the node
            taken[(size_t)idx] = true;
            landed.emplace_back(pods[i], &*out[i].bound_to);  // (`out` keeps its size: the pointer stays valid)
        }
        ctx.snapshot->observe_bound(landed);
        pending.swap(next);
    }
    for (size_t i : pending) out[i] = bind(*pods[i], nullptr, sink);  // still colliding after max_rounds
    if (stats) *stats = st;
    warn_failed(pods, out, ctx);
    return out;
}

}  // namespace ksched_host
