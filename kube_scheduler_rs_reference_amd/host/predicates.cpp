#include "predicates.hpp"

#include "sharded.hpp"

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

// One-pod x one-node evaluation on the device: bit 0 of the single mask word.
bool eval_pair(Snapshot &snap, const corev1::Pod &pod, uint32_t flags) {
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
    PodColumns pc = snap.encode_pods({&pod});
    uint64_t feas = 0, fit = 0;
    DeviceEvaluator &dev = snap.device();
    dev.check(ksched_eval(dev.handle(), 1, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(),
                          pc.n_keys ? pc.sel_val_ids.data() : nullptr, nullptr, nullptr, 0,
                          KSCHED_FIT | KSCHED_SEL | KSCHED_WANT_FIT_MASK, &feas, &fit, nullptr),
              "ksched_eval");
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

// One device call for pods [lo, hi) of the batch, written into rows [lo, hi) of `out`.
void eval_range(Snapshot &snap, const std::vector<const corev1::Pod *> &pods, size_t lo, size_t hi, uint32_t pick,
                const std::vector<uint32_t> *samples, uint32_t attempts, BatchValidity &out, bool want_masks) {
    const std::vector<const corev1::Pod *> part(pods.begin() + (std::ptrdiff_t)lo, pods.begin() + (std::ptrdiff_t)hi);
    PodColumns pc = snap.encode_pods(part);
    const uint32_t flags = out.flags | (want_masks ? KSCHED_WANT_FIT_MASK : 0u) | pick;
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
}

}  // namespace

BatchValidity check_node_validity_batch(const std::vector<const corev1::Pod *> &pods, Context &ctx, bool taints, uint32_t pick_flags,
                                        const std::vector<uint32_t> *samples, uint32_t attempts, bool want_masks) {
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
    // The device takes at most KSCHED_MAX_KEYS label columns per call.  The reference has no limit on selector keys, so a batch
    // that uses more distinct keys is evaluated in consecutive pod ranges, each within the budget (a new range re-uploads the
    // label columns it needs).  Only a single pod with more than KSCHED_MAX_KEYS selector keys is refused.
    size_t lo = 0;
    std::set<std::string> keys;
    for (size_t i = 0; i < pods.size(); ++i) {
        const corev1::Pod &pod = *pods[i];
        if (!pod.spec || !pod.spec->node_selector || pod.spec->node_selector->empty()) continue;  // (most pods: nothing is built for them)
        const auto &selector = *pod.spec->node_selector;  // (a map: its keys are distinct)
        if (selector.size() > KSCHED_MAX_KEYS)  // a per-pod failure raised before anything of the batch was evaluated or POSTed: PodEncodeError, so that
                                                // run_batches isolates the offender instead of the exception taking the scheduling loop down
            throw PodEncodeError("pod " + full_name(pod.metadata) + ": more than KSCHED_MAX_KEYS nodeSelector keys on one pod");
        size_t adds = 0;
        for (const auto &kv : selector) adds += keys.find(kv.first) == keys.end() ? 1u : 0u;
        if (!adds) continue;
        if (keys.size() + adds > KSCHED_MAX_KEYS) {
            eval_range(snap, pods, lo, i, pick, samples, attempts, out, want_masks);
            lo = i;
            keys.clear();
        }
        for (const auto &kv : selector) keys.insert(kv.first);
    }
    eval_range(snap, pods, lo, pods.size(), pick, samples, attempts, out, want_masks);
    return out;
}

std::vector<Validity> explain_pairs(const std::vector<const corev1::Pod *> &pods, Context &ctx,
                                    const std::vector<std::pair<uint32_t, uint32_t>> &pairs, bool taints) {
    if (!ctx.snapshot) ctx.refresh_snapshot();
    Snapshot &snap = *ctx.snapshot;
    if (taints && snap.has_taints()) snap.enable_taints();
    std::vector<Validity> out(pairs.size());
    if (pairs.empty()) return out;
    PodColumns pc = snap.encode_pods(pods);  // (one call: at most KSCHED_MAX_KEYS distinct selector keys among `pods`)
    std::vector<uint32_t> pp(pairs.size()), pn(pairs.size());
    for (size_t i = 0; i < pairs.size(); ++i) {
        pp[i] = pairs[i].first;
        pn[i] = pairs[i].second;
    }
    std::vector<int32_t> reason(pairs.size());
    const uint32_t flags = KSCHED_FIT | KSCHED_SEL | ((taints && snap.has_taints()) ? KSCHED_TAINT : 0u);
    DeviceEvaluator &dev = snap.device();
    dev.check(ksched_explain(dev.handle(), pc.p, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(), pc.n_keys ? pc.sel_val_ids.data() : nullptr,
                             (flags & KSCHED_TAINT) ? pc.tolerations.data() : nullptr, (uint32_t)pairs.size(), pp.data(), pn.data(), flags,
                             reason.data()),
              "ksched_explain");
    for (size_t i = 0; i < pairs.size(); ++i) {
        switch (reason[i]) {
            case KSCHED_REASON_OK: out[i] = std::nullopt; break;
            case KSCHED_REASON_NOT_ENOUGH_RESOURCES: out[i] = InvalidNodeReason::NotEnoughResources; break;
            case KSCHED_REASON_TAINT_NOT_TOLERATED: out[i] = InvalidNodeReason::TaintNotTolerated; break;
            default: out[i] = InvalidNodeReason::NodeSelectorMismatch;
        }
    }
    return out;
}

}  // namespace predicates
}  // namespace ksched_host
