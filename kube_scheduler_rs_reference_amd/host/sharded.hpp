// sharded.hpp -- one host process, several MI355X: the pod batch row-shards over the devices of the node, the node snapshot is
// replicated, and ONE RCCL all-gather of the int32 (pod -> node) bindings over xGMI gives every device -- and, through one copy
// from device 0, the host -- the whole table (north_star; include/ksched.h "one host thread, several
// devices").
//
// The reference is one process (src/main.rs:127-152) whose reconciles each test a handful of candidates; this is what lets the
// drop-in host span the 8 GPUs of a node WITHOUT becoming 8 processes: `Context` keeps one `Snapshot`, the snapshot keeps one
// `DeviceEvaluator` per device of $KSCHED_DEVICES, and `check_node_validity_batch` hands a batch to the ShardedContext instead
// of to a single ksched_eval:
//
//     rows [lo_r, hi_r) = ksched_shard_bounds(p, n, r)                                  (contiguous, ceil(p / n) per device)
//     for every device r : ksched_eval_begin(ctx_r, rows of r ...)                      copies in + kernels enqueued, no host wait
//     ksched_allgather_bindings_local(comms, n, local[], gathered[], ceil(p / n), streams[])
//     ksched_eval_end(ctx_0, gathered_0, n * ceil(p / n), table)                        one copy, one wait
//     binding[lo_r + i] = table[r * ceil(p / n) + i]                                    (merge_gathered)
//
// Masks are not exchanged (6.3 GB at configs[4]); when the caller asks for them each device copies ITS rows
// straight into rows [lo_r, hi_r) of the caller's arrays.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ksched.h"
#include "encoder.hpp"

namespace ksched_host {

// The row split and its inverse -- pure arithmetic (tests/cpp/host_tests.cpp "cpu" checks them against every (p, n) up to a bound).
struct ShardBounds {
    uint32_t lo = 0, hi = 0, count_per_rank = 0;
};
ShardBounds shard_bounds(uint32_t p, uint32_t nranks, uint32_t rank);  // == ksched_shard_bounds
// table = the all-gathered bindings, [nranks][count_per_rank] (rows past a shard's end are padding); out[p] = pod order
void merge_gathered(const int32_t *table, uint32_t p, uint32_t nranks, int32_t *out);

// $KSCHED_DEVICES: "0,1,2,3" = those HIP devices, "all" = every visible one, unset / empty = {fallback}.  Throws EncodeError on
// anything else (a device listed twice, a device the process does not see, not a number).
std::vector<int> devices_from_env(const char *value, int fallback);

class ShardedContext {
public:
    // How the devices' bindings meet.  Rccl (the product path, always the default): ncclAllGather through
    // ksched_allgather_bindings_local, then ONE copy of the table from device 0.  HostCopies: every device copies its own rows
    // to the host table -- no collective; only for tests that put several evaluators on ONE physical GPU (RCCL refuses a
    // communicator with a device listed twice), never chosen implicitly.
    enum class Exchange { Rccl, HostCopies };

    // `devs`: one evaluator per shard (normally one per device).  Throws EncodeError when the RCCL communicator cannot be built:
    // there is no silent single-device or host-side stand-in.
    explicit ShardedContext(std::vector<std::shared_ptr<DeviceEvaluator>> devs, Exchange exchange = Exchange::Rccl);
    ~ShardedContext();
    ShardedContext(const ShardedContext &) = delete;
    ShardedContext &operator=(const ShardedContext &) = delete;

    uint32_t size() const { return (uint32_t)devs_.size(); }
    Exchange exchange() const { return exchange_; }

    // One batch.  `pc` = the whole batch's columns (n_keys rows of pc.p selector ids); `samples` = [pc.p][attempts] or nullptr;
    // flags as for ksched_eval.  out_feasible / out_fit: [pc.p][W] or nullptr; out_binding: [pc.p] or nullptr (required with a pick).
    void eval(const PodColumns &pc, const uint32_t *samples, uint32_t attempts, uint32_t flags, uint32_t W, uint64_t *out_feasible,
              uint64_t *out_fit, int32_t *out_binding);

    uint64_t batches() const { return batches_; }  // observability: evaluations that went through the exchange
    bool broken() const { return broken_; }        // a failed exchange aborted the communicator: every further eval throws

private:
    std::vector<std::shared_ptr<DeviceEvaluator>> devs_;
    std::vector<ksched_comm *> comms_;
    Exchange exchange_;
    std::vector<int32_t> table_;  // [n][count_per_rank], host copy of the gathered bindings
    uint64_t batches_ = 0;
    bool broken_ = false;  // a failed exchange aborted the communicator clique
};

}  // namespace ksched_host
