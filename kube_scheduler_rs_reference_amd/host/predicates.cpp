#include "predicates.hpp"

#include "pool.hpp"
#include "sharded.hpp"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <set>

namespace ksched_host {
namespace predicates {

const char *debug_name(InvalidNodeReason r) {
    switch (r) {
        case InvalidNodeReason::NotEnoughResources: return "NotEnoughResources";
        case InvalidNodeReason::NodeSelectorMismatch: return "NodeSelectorMismatch";
        case InvalidNodeReason::TaintNotTolerated: return "TaintNotTolerated";
    }
    return "?";
}

namespace {

std::vector<corev1::Pod> split_wide_pod(const corev1::Pod &pod);
bool is_wide(const corev1::Pod &pod) { return pod.spec && pod.spec->node_selector && pod.spec->node_selector->size() > KSCHED_MAX_KEYS; }

// One-pod x one-node evaluation on the device: bit 0 of the single mask word.
bool eval_pair(Snapshot &snap, const corev1::Pod &pod, uint32_t flags) {
    if (is_wide(pod)) {  // more selector keys than one call takes: group by group, every group must hold (a conjunction, src/predicates.rs:48-53)
        bool ok = true;
        for (const corev1::Pod &part : split_wide_pod(pod)) ok = eval_pair(snap, part, flags) && ok;
        return ok;
    }
    PodColumns pc = snap.encode_pods({&pod});  // may add label columns -> re-upload
    uint64_t word = 0;
    DeviceEvaluator &dev = snap.device();
    dev.check(ksched_eval(dev.handle(), 1, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(),
                          pc.n_keys ? pc.sel_val_ids.data() : nullptr, nullptr, nullptr, 0, flags, &word, nullptr, nullptr),
              "ksched_eval");
    return word & 1ull;
}

Snapshot &pair_snapshot(Context &ctx) {
    if (!ctx.pair_snapshot) ctx.pair_snapshot = std::make_shared<Snapshot>(ctx.device);
    return *ctx.pair_snapshot;
}

// does_node_selector_match has no Context argument (src/predicates.rs:45), so its evaluator is a
// process-wide one on device $KSCHED_DEVICE (default 0), serialised by a mutex.
std::mutex g_sel_mu;
std::shared_ptr<Snapshot> g_sel_snapshot;

}  // namespace

bool can_pod_fit(const corev1::Pod &pod, const corev1::Node &node, Context &ctx) {
    // src/predicates.rs:21-38: available = allocatable - every pod the LIST for this node returns.
    // Snapshot::rebuild issues exactly that LIST (one call) and uploads the one-node columns.
    Snapshot &snap = pair_snapshot(ctx);
    snap.rebuild({node}, ctx.client.get());
    return eval_pair(snap, pod, KSCHED_FIT);  // :40-42 on the device
}

bool does_node_selector_match(const corev1::Pod &pod, const corev1::Node &node) {
    std::lock_guard<std::mutex> lk(g_sel_mu);
    if (!g_sel_snapshot) {
        const char *d = std::getenv("KSCHED_DEVICE");
        g_sel_snapshot = std::make_shared<Snapshot>(d ? std::atoi(d) : 0);
    }
    g_sel_snapshot->rebuild({node}, nullptr, /*with_resources=*/false);
    return eval_pair(*g_sel_snapshot, pod, KSCHED_SEL);
}

Validity check_node_validity(const corev1::Pod &pod, const corev1::Node &node, Context &ctx) {
    // One device call gives both masks; the order of the reasons is the reference's (:68-74).
    Snapshot &snap = pair_snapshot(ctx);
    snap.rebuild({node}, ctx.client.get());
    uint64_t feas = ~0ull, fit = 0;
    DeviceEvaluator &dev = snap.device();
    std::vector<corev1::Pod> groups;
    if (is_wide(pod)) groups = split_wide_pod(pod);  // (more selector keys than one call takes: the groups' feasible bits ANDed)
    for (size_t g = 0; g < std::max<size_t>(1, groups.size()); ++g) {
        const corev1::Pod &part = groups.empty() ? pod : groups[g];
        PodColumns pc = snap.encode_pods({&part});
        uint64_t f = 0;
        dev.check(ksched_eval(dev.handle(), 1, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(),
                              pc.n_keys ? pc.sel_val_ids.data() : nullptr, nullptr, nullptr, 0,
                              KSCHED_FIT | KSCHED_SEL | KSCHED_WANT_FIT_MASK, &f, &fit, nullptr),
                  "ksched_eval");
        feas &= f;
    }
    const int r = ksched_reason(&feas, &fit, 0, KSCHED_FIT | KSCHED_SEL);
    if (r == KSCHED_REASON_OK) return std::nullopt;
    return r == KSCHED_REASON_NOT_ENOUGH_RESOURCES ? InvalidNodeReason::NotEnoughResources : InvalidNodeReason::NodeSelectorMismatch;
}

Validity BatchValidity::validity(uint32_t pod, uint32_t node) const {
    const int r = ksched_reason(feasible.data() + (size_t)pod * W, fit.data() + (size_t)pod * W, node, flags);
    switch (r) {
        case KSCHED_REASON_OK: return std::nullopt;
        case KSCHED_REASON_NOT_ENOUGH_RESOURCES: return InvalidNodeReason::NotEnoughResources;
        case KSCHED_REASON_TAINT_NOT_TOLERATED: return InvalidNodeReason::TaintNotTolerated;
        default: return InvalidNodeReason::NodeSelectorMismatch;
    }
}

uint64_t BatchValidity::feasible_count(uint32_t pod) const {
    uint64_t c = 0;
    for (uint32_t w = 0; w < W; ++w) c += (uint64_t)__builtin_popcountll(feasible[(size_t)pod * W + w]);
    return c;
}

namespace {

// The device takes at most KSCHED_MAX_KEYS label columns per call; the reference has no limit on selector keys
// (src/predicates.rs:48-53).  A batch is therefore cut into consecutive pod ranges whose distinct selector keys fit the budget (a new
// range re-uploads the label columns it needs), and a single pod with MORE keys than one call takes is a range of its own, marked wide:
// it is evaluated once per group of KSCHED_MAX_KEYS keys and the groups' masks are ANDed (a selector is a conjunction).
struct KeyRange {
    size_t lo, hi;
    bool wide;
};
std::vector<KeyRange> key_ranges(const std::vector<const corev1::Pod *> &pods, std::optional<std::set<std::string>> *whole_batch_keys = nullptr) {
    std::vector<KeyRange> out;
    // The common case first, on the worker threads: the whole batch names at most KSCHED_MAX_KEYS distinct keys and no pod is wide -- ONE
    // range, and its key set goes on to the encoder.  (The serial walk below was 17 of a C3-size batch's 25 ms on the way to the device,
    // profiles/r06_host_loop.txt; it is still what cuts a batch that needs cutting.)
    if (pods.size() >= 4096) {
        bool any_wide = false;
        std::set<std::string> all = Snapshot::batch_selector_keys(pods, &any_wide);
        if (!any_wide && all.size() <= KSCHED_MAX_KEYS) {
            out.push_back({0, pods.size(), false});
            if (whole_batch_keys) *whole_batch_keys = std::move(all);
            return out;
        }
    }
    size_t lo = 0;
    std::set<std::string> keys;
    auto close = [&](size_t hi) {
        if (hi > lo) out.push_back({lo, hi, false});
        lo = hi;
        keys.clear();
    };
    for (size_t i = 0; i < pods.size(); ++i) {
        const corev1::Pod &pod = *pods[i];
        if (!pod.spec || !pod.spec->node_selector || pod.spec->node_selector->empty()) continue;  // (most pods: nothing is built for them)
        const auto &selector = *pod.spec->node_selector;  // (a map: its keys are distinct)
        if (selector.size() > KSCHED_MAX_KEYS) {
            close(i);
            out.push_back({i, i + 1, true});
            lo = i + 1;
            continue;
        }
        size_t adds = 0;
        for (const auto &kv : selector) adds += keys.find(kv.first) == keys.end() ? 1u : 0u;
        if (!adds) continue;
        if (keys.size() + adds > KSCHED_MAX_KEYS) close(i);
        for (const auto &kv : selector) keys.insert(kv.first);
    }
    close(pods.size());
    return out;
}
// the pod once per group of at most KSCHED_MAX_KEYS of its selector's keys (every copy carries the whole pod otherwise: requests, tolerations)
std::vector<corev1::Pod> split_wide_pod(const corev1::Pod &pod) {
    std::vector<corev1::Pod> out;
    const auto &selector = *pod.spec->node_selector;
    corev1::StringMap group;
    auto flush = [&] {
        corev1::Pod part = pod;
        part.spec->node_selector = group;
        out.push_back(std::move(part));
        group.clear();
    };
    for (const auto &kv : selector) {
        group.insert(kv);
        if (group.size() == KSCHED_MAX_KEYS) flush();
    }
    if (!group.empty()) flush();
    return out;
}

// A pod whose selector has more keys than one call takes (row `i` of `out`): one device evaluation per key group against that group's
// label columns, the groups' feasible masks ANDed (does_node_selector_match is a conjunction over the keys, src/predicates.rs:48-53;
// the fit mask is the same in every group), and the pick made by the device from the combined mask (ksched_pick).  On the first
// device of a multi-device snapshot: such pods are rare, and every device holds the whole snapshot.
void eval_wide_pod(Snapshot &snap, const corev1::Pod &pod, size_t i, uint32_t pick, const std::vector<uint32_t> *samples, uint32_t attempts,
                   BatchValidity &out, bool want_masks, const std::function<void()> *samples_ready = nullptr) {
    const uint32_t W = out.W;
    std::vector<uint64_t> feas(W, ~0ull), fit(W, 0ull), f(W), ft(W);
    DeviceEvaluator &dev = snap.device();
    int64_t req_mem = 0;
    for (const corev1::Pod &part : split_wide_pod(pod)) {
        PodColumns pc = snap.encode_pods({&part});  // (re-uploads the label columns of this group's keys)
        dev.check(ksched_eval(dev.handle(), 1, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(), pc.n_keys ? pc.sel_val_ids.data() : nullptr,
                              (out.flags & KSCHED_TAINT) ? pc.tolerations.data() : nullptr, nullptr, 0, out.flags | KSCHED_WANT_FIT_MASK, f.data(), ft.data(), nullptr),
                  "ksched_eval");
        for (uint32_t w = 0; w < W; ++w) feas[w] &= f[w];
        fit = ft;
        req_mem = pc.req_mem_bytes[0];
    }
    if (want_masks) {
        std::copy(feas.begin(), feas.end(), out.feasible.begin() + (std::ptrdiff_t)(i * W));
        std::copy(fit.begin(), fit.end(), out.fit.begin() + (std::ptrdiff_t)(i * W));
    }
    if (pick && samples_ready) (*samples_ready)();
    if (pick)
        dev.check(ksched_pick(dev.handle(), 1, feas.data(), &req_mem, (pick & KSCHED_PICK_SAMPLED) ? samples->data() + i * attempts : nullptr, attempts,
                              pick | (out.flags & (KSCHED_FIT | KSCHED_SEL | KSCHED_TAINT)), out.binding.data() + i),
                  "ksched_pick");
}

// One device call for pods [lo, hi) of the batch, written into rows [lo, hi) of `out`.
void eval_range(Snapshot &snap, const std::vector<const corev1::Pod *> &pods, size_t lo, size_t hi, uint32_t pick,
                const std::vector<uint32_t> *samples, uint32_t attempts, BatchValidity &out, bool want_masks,
                const std::set<std::string> *known_keys = nullptr, const std::function<void()> *samples_ready = nullptr) {
    PhaseClock clock("eval_range");
    const bool whole = lo == 0 && hi == pods.size();
    const std::vector<const corev1::Pod *> part = whole ? std::vector<const corev1::Pod *>() : std::vector<const corev1::Pod *>(pods.begin() + (std::ptrdiff_t)lo, pods.begin() + (std::ptrdiff_t)hi);
    PodColumns pc = snap.encode_pods(whole ? pods : part, whole ? known_keys : nullptr);
    clock.lap("encode_pods");
    if (out.req_cpu_nanos.size() == out.p) {
        std::copy(pc.req_cpu_nanos.begin(), pc.req_cpu_nanos.end(), out.req_cpu_nanos.begin() + (std::ptrdiff_t)lo);
        std::copy(pc.req_mem_nanos.begin(), pc.req_mem_nanos.end(), out.req_mem_nanos.begin() + (std::ptrdiff_t)lo);
    }
    const uint32_t flags = out.flags | (want_masks ? KSCHED_WANT_FIT_MASK : 0u) | pick;
    if (samples_ready) (*samples_ready)();  // (the draws are read from here on)
    clock.lap("waiting for the draws");
    if (ShardedContext *sh = snap.sharded()) {
        // several devices (or KSCHED_SHARDED): the range's rows are cut over them, each device evaluates its rows against its
        // replica of the snapshot, the bindings meet in one RCCL all-gather (sharded.hpp); masks land in this range's rows directly
        sh->eval(pc, (pick & KSCHED_PICK_SAMPLED) ? samples->data() + lo * attempts : nullptr, attempts, flags, out.W,
                 want_masks ? out.feasible.data() + lo * out.W : nullptr, want_masks ? out.fit.data() + lo * out.W : nullptr,
                 pick ? out.binding.data() + lo : nullptr);
        return;
    }
    DeviceEvaluator &dev = snap.device();
    dev.check(ksched_eval(dev.handle(), pc.p, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(),
                          pc.n_keys ? pc.sel_val_ids.data() : nullptr, (out.flags & KSCHED_TAINT) ? pc.tolerations.data() : nullptr,
                          (pick & KSCHED_PICK_SAMPLED) ? samples->data() + lo * attempts : nullptr, attempts,
                          flags, want_masks ? out.feasible.data() + lo * out.W : nullptr,
                          want_masks ? out.fit.data() + lo * out.W : nullptr, pick ? out.binding.data() + lo : nullptr),
              "ksched_eval");
    clock.lap("ksched_eval (copies in, device, bindings out)");
}

}  // namespace

BatchValidity check_node_validity_batch(const std::vector<const corev1::Pod *> &pods, Context &ctx, bool taints, uint32_t pick_flags,
                                        const std::vector<uint32_t> *samples, uint32_t attempts, bool want_masks, const std::function<void()> *samples_ready) {
    if (!ctx.snapshot) ctx.refresh_snapshot();
    Snapshot &snap = *ctx.snapshot;
    if (taints && snap.has_taints()) snap.enable_taints();  // extension E2 is opt-in: interning happens (and can fail) only here
    BatchValidity out;
    out.p = (uint32_t)pods.size();
    out.n = snap.n();
    out.W = snap.mask_words();
    out.flags = KSCHED_FIT | KSCHED_SEL | ((taints && snap.has_taints()) ? KSCHED_TAINT : 0u);
    const uint32_t pick = pick_flags & (KSCHED_PICK_SAMPLED | KSCHED_PICK_BESTFIT);
    if (!want_masks && !pick) throw EncodeError("check_node_validity_batch: nothing asked for (no masks, no pick)");
    if (want_masks) {
        out.feasible.assign((size_t)out.p * out.W, 0ull);
        out.fit.assign((size_t)out.p * out.W, 0ull);
    }
    if (pick) out.binding.assign(out.p, -1);
    if (out.p == 0 || out.n == 0) return out;  // no pods, or an empty store: nothing is feasible
    if ((pick & KSCHED_PICK_SAMPLED) && (!samples || samples->size() != (size_t)out.p * attempts))
        throw EncodeError("check_node_validity_batch: samples must hold p * attempts indices");
    // (key_ranges: consecutive pod ranges within the device's budget of label columns per call; a pod with more keys than that is
    // evaluated group by group and ANDed -- no input the reference schedules is refused)
    PhaseClock clock("check_node_validity_batch");
    std::optional<std::set<std::string>> whole_keys;  // set when the batch is ONE range whose keys the plan has collected already
    const std::vector<KeyRange> ranges = key_ranges(pods, &whole_keys);
    clock.lap("key_ranges");
    out.req_cpu_nanos.assign(out.p, 0);
    out.req_mem_nanos.assign(out.p, 0);
    out.exact_requests = true;
    for (const KeyRange &r : ranges) {
        if (r.wide) {
            out.exact_requests = false;  // (its rows come from the pod's key groups; a caller that needs the sums parses that pod itself)
            eval_wide_pod(snap, *pods[r.lo], r.lo, pick, samples, attempts, out, want_masks, samples_ready);
        } else {
            eval_range(snap, pods, r.lo, r.hi, pick, samples, attempts, out, want_masks, whole_keys ? &*whole_keys : nullptr, samples_ready);
        }
    }
    return out;
}

std::vector<DeviceCall> device_calls(const std::vector<const corev1::Pod *> &pods) {
    std::vector<DeviceCall> out;
    for (const KeyRange &r : key_ranges(pods)) {
        DeviceCall c;
        c.lo = r.lo;
        c.hi = r.hi;
        if (r.wide) c.groups = split_wide_pod(*pods[r.lo]);
        out.push_back(std::move(c));
    }
    return out;
}

std::vector<Validity> explain_pairs(const std::vector<const corev1::Pod *> &pods, Context &ctx,
                                    const std::vector<std::pair<uint32_t, uint32_t>> &pairs, bool taints) {
    if (!ctx.snapshot) ctx.refresh_snapshot();
    Snapshot &snap = *ctx.snapshot;
    if (taints && snap.has_taints()) snap.enable_taints();
    std::vector<Validity> out(pairs.size());
    if (pairs.empty()) return out;
    const uint32_t flags = KSCHED_FIT | KSCHED_SEL | ((taints && snap.has_taints()) ? KSCHED_TAINT : 0u);
    DeviceEvaluator &dev = snap.device();
    auto to_validity = [](int32_t r) -> Validity {
        switch (r) {
            case KSCHED_REASON_OK: return std::nullopt;
            case KSCHED_REASON_NOT_ENOUGH_RESOURCES: return InvalidNodeReason::NotEnoughResources;
            case KSCHED_REASON_TAINT_NOT_TOLERATED: return InvalidNodeReason::TaintNotTolerated;
            default: return InvalidNodeReason::NodeSelectorMismatch;
        }
    };
    // pairs by pod range (the same key-budget ranges the evaluation uses: one ksched_explain per range that has pairs)
    std::vector<size_t> order(pairs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return pairs[a].first < pairs[b].first; });
    size_t next = 0;
    for (const KeyRange &r : key_ranges(pods)) {
        std::vector<size_t> mine;
        while (next < order.size() && pairs[order[next]].first < r.hi) mine.push_back(order[next++]);
        if (mine.empty()) continue;
        std::vector<uint32_t> pp(mine.size()), pn(mine.size());
        std::vector<int32_t> reason(mine.size());
        for (size_t k = 0; k < mine.size(); ++k) {
            if (pairs[mine[k]].first < r.lo) throw EncodeError("explain_pairs: a pair names a pod outside the batch");
            pp[k] = pairs[mine[k]].first - (uint32_t)r.lo;
            pn[k] = pairs[mine[k]].second;
        }
        // a wide pod: group by group; the first failure in the reference's order wins (resources, src/predicates.rs:68-70, say the same
        // in every group; then any group's selector mismatch, :72-74)
        std::vector<corev1::Pod> groups;
        std::vector<std::vector<const corev1::Pod *>> calls;
        if (r.wide) {
            groups = split_wide_pod(*pods[r.lo]);
            for (const corev1::Pod &g : groups) calls.push_back({&g});
        } else {
            calls.emplace_back(pods.begin() + (std::ptrdiff_t)r.lo, pods.begin() + (std::ptrdiff_t)r.hi);
        }
        std::vector<int32_t> combined(mine.size(), KSCHED_REASON_OK);
        for (const auto &part : calls) {
            PodColumns pc = snap.encode_pods(part);
            dev.check(ksched_explain(dev.handle(), pc.p, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(), pc.n_keys ? pc.sel_val_ids.data() : nullptr,
                                     (flags & KSCHED_TAINT) ? pc.tolerations.data() : nullptr, (uint32_t)mine.size(), pp.data(), pn.data(), flags, reason.data()),
                      "ksched_explain");
            // (over the groups of a wide pod the strongest reason in check_node_validity's order wins: resources, then ANY group's selector
            // mismatch, then -- extension E2 -- the taints; a taint verdict of an early group must not hide a later group's selector mismatch)
            auto rank = [](int32_t r) { return r == KSCHED_REASON_NOT_ENOUGH_RESOURCES ? 3 : r == KSCHED_REASON_NODE_SELECTOR_MISMATCH ? 2 : r == KSCHED_REASON_OK ? 0 : 1; };
            for (size_t k = 0; k < mine.size(); ++k)
                if (rank(reason[k]) > rank(combined[k])) combined[k] = reason[k];
        }
        for (size_t k = 0; k < mine.size(); ++k) out[mine[k]] = to_validity(combined[k]);
    }
    if (next != order.size()) throw EncodeError("explain_pairs: a pair names a pod outside the batch");
    return out;
}

}  // namespace predicates
}  // namespace ksched_host
