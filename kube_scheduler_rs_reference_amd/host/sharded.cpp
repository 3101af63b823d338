#include "sharded.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <set>

namespace ksched_host {

ShardBounds shard_bounds(uint32_t p, uint32_t nranks, uint32_t rank) {
    ShardBounds b;
    ksched_shard_bounds(p, nranks, rank, &b.lo, &b.hi, &b.count_per_rank);  // (one definition: the C ABI's)
    return b;
}

void merge_gathered(const int32_t *table, uint32_t p, uint32_t nranks, int32_t *out) {
    for (uint32_t r = 0; r < nranks; ++r) {
        const ShardBounds b = shard_bounds(p, nranks, r);
        if (b.hi > b.lo) std::memcpy(out + b.lo, table + (size_t)r * b.count_per_rank, (size_t)(b.hi - b.lo) * sizeof(int32_t));
    }
}

std::vector<int> devices_from_env(const char *value, int fallback) {
    if (!value || !*value) return {fallback};
    const int visible = ksched_device_count();
    std::vector<int> out;
    if (std::strcmp(value, "all") == 0) {
        for (int d = 0; d < visible; ++d) out.push_back(d);
        if (out.empty()) throw EncodeError("KSCHED_DEVICES=all: the process sees no HIP device");
        return out;
    }
    std::set<int> seen;
    const char *s = value;
    // the same reading as the Rust twin's parse_device_ids (rust/src/ksched.rs): entries separated by commas, blanks around an entry ignored,
    // an empty entry ("0,1," or "0,,1") is an error like anything else that is not a number
    while (*s) {
        while (*s == ' ' || *s == '\t') ++s;
        char *end = nullptr;
        const bool digit = *s >= '0' && *s <= '9';  // (strtol would also take a sign and leading white space of its own)
        const long d = digit ? std::strtol(s, &end, 10) : -1;
        while (end && (*end == ' ' || *end == '\t')) ++end;
        if (!digit || end == s || d < 0 || (*end != ',' && *end != '\0') || (*end == ',' && end[1] == '\0'))
            throw EncodeError(std::string("KSCHED_DEVICES: cannot read '") + value + "' (expected e.g. 0,1,2,3 or all)");
        if (d >= visible) throw EncodeError("KSCHED_DEVICES names device " + std::to_string(d) + ", the process sees " + std::to_string(visible));
        if (!seen.insert((int)d).second) throw EncodeError("KSCHED_DEVICES lists device " + std::to_string(d) + " twice");
        out.push_back((int)d);
        s = *end ? end + 1 : end;
    }
    if (out.empty()) return {fallback};
    return out;
}

ShardedContext::ShardedContext(std::vector<std::shared_ptr<DeviceEvaluator>> devs, Exchange exchange) : devs_(std::move(devs)), exchange_(exchange) {
    if (devs_.empty()) throw EncodeError("ShardedContext: no device");
    for (const auto &d : devs_)
        if (!d) throw EncodeError("ShardedContext: null evaluator");
    if (exchange_ == Exchange::Rccl) {
        std::vector<ksched_ctx *> ctxs;
        for (const auto &d : devs_) ctxs.push_back(d->handle());
        comms_.assign(devs_.size(), nullptr);
        const int rc = ksched_comm_create_local(ctxs.data(), (int)ctxs.size(), comms_.data());  // ncclCommInitAll over the evaluators' devices
        if (rc != KSCHED_OK) {
            comms_.clear();
            throw EncodeError(std::string("ksched_comm_create_local: ") + ksched_strerror(rc) + " (" + ksched_comm_last_error() + ")");
        }
    }
}

ShardedContext::~ShardedContext() {
    for (ksched_comm *c : comms_) ksched_comm_destroy(c);
}

void ShardedContext::eval(const PodColumns &pc, const uint32_t *samples, uint32_t attempts, uint32_t flags, uint32_t W, uint64_t *out_feasible,
                          uint64_t *out_fit, int32_t *out_binding) {
    const uint32_t n = size(), p = pc.p;
    const bool pick = flags & (KSCHED_PICK_SAMPLED | KSCHED_PICK_BESTFIT);
    if (pick && !out_binding) throw EncodeError("ShardedContext::eval: a pick needs out_binding");
    if (p == 0) return;
    const uint32_t cpr = shard_bounds(p, n, 0).count_per_rank;
    std::vector<int32_t *> local(n, nullptr), gathered(n, nullptr);
    std::vector<void *> streams(n, nullptr);
    uint32_t touched = 0;  // devices a call of this batch has entered -- the failing one included: its copies may already be under way
    std::string failure;
    if (broken_) throw EncodeError("ShardedContext: the communicator was aborted after a failed exchange; build a new context");
    // 1. every device gets its rows: copies in and kernels enqueued on the device's own stream, the host does not wait
    for (uint32_t r = 0; r < n && failure.empty(); ++r) {
        const ShardBounds b = shard_bounds(p, n, r);
        const uint32_t lo = b.lo, rows = b.hi - b.lo;
        const int rc = ksched_eval_begin(devs_[r]->handle(), rows, pc.req_cpu_milli.data() + lo, pc.req_mem_bytes.data() + lo,
                                         pc.n_keys ? pc.sel_val_ids.data() + lo : nullptr, p,  // rows [lo, hi) of the [n_keys][p] array
                                         (flags & KSCHED_TAINT) && !pc.tolerations.empty() ? pc.tolerations.data() + lo : nullptr,
                                         (flags & KSCHED_PICK_SAMPLED) ? samples + (size_t)lo * attempts : nullptr, attempts, flags,
                                         out_feasible ? out_feasible + (size_t)lo * W : nullptr, out_fit ? out_fit + (size_t)lo * W : nullptr, cpr, &local[r],
                                         &streams[r]);
        ++touched;
        if (rc != KSCHED_OK)
            failure = "ksched_eval_begin on shard " + std::to_string(r) + ": " + ksched_strerror(rc) + " (" + ksched_last_error(devs_[r]->handle()) + ")";
    }
    // 2. the exchange: one all-gather of ceil(p / n) int32 per device over xGMI, enqueued behind each device's pick on its own stream
    if (failure.empty() && pick && exchange_ == Exchange::Rccl) {
        for (uint32_t r = 0; r < n && failure.empty(); ++r) {
            const int rc = ksched_gather_buffer(devs_[r]->handle(), n * cpr, &gathered[r]);
            if (rc != KSCHED_OK) failure = "ksched_gather_buffer on shard " + std::to_string(r) + ": " + ksched_strerror(rc);
        }
        if (failure.empty()) {
            const int rc = ksched_allgather_bindings_local(comms_.data(), (int)n, local.data(), gathered.data(), cpr, streams.data());
            if (rc != KSCHED_OK) {  // the library has aborted the clique (nothing half issued is left on the streams): this context is done for
                failure = std::string("ksched_allgather_bindings_local: ") + ksched_strerror(rc) + " (" + ksched_comm_last_error() + ")";
                broken_ = true;
            }
        }
    }
    // 3. the table comes back in one copy from device 0 (every device holds it); the other devices only finish their streams --
    //    whatever happened above, every device a call has entered is waited for before this function returns or throws (the
    //    inputs must stay alive until ksched_eval_end, include/ksched.h -- also for a shard whose ksched_eval_begin failed half way)
    if (pick) table_.resize((size_t)n * cpr);
    for (uint32_t r = 0; r < touched; ++r) {
        const int32_t *src = nullptr;
        int32_t *dst = nullptr;
        uint32_t count = 0;
        if (failure.empty() && pick) {
            if (exchange_ == Exchange::Rccl) {
                if (r == 0) src = gathered[0], dst = table_.data(), count = n * cpr;
            } else {
                src = local[r], dst = table_.data() + (size_t)r * cpr, count = cpr;
            }
        }
        const int rc = ksched_eval_end(devs_[r]->handle(), src, count, dst);
        if (rc != KSCHED_OK && failure.empty())
            failure = "ksched_eval_end on shard " + std::to_string(r) + ": " + ksched_strerror(rc) + " (" + ksched_last_error(devs_[r]->handle()) + ")";
    }
    if (!failure.empty()) throw EncodeError(failure);
    if (pick) merge_gathered(table_.data(), p, n, out_binding);
    ++batches_;
}

}  // namespace ksched_host
