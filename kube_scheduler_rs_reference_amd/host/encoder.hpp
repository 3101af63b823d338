// encoder.hpp -- corev1 objects -> the integer SoA columns of include/ksched.h, and the
// device-resident snapshot built from them.
//
// This is the wire-format step on either side of the kernel: quantity strings become exact
// int64 milli-cores / bytes, label strings become dictionary ids (exact interning, never a
// hash), taints become bit positions.  Nodes are put in canonical order (ascending name).
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <string_view>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/ksched.h"
#include "corev1.hpp"
#include "util.hpp"

namespace ksched_host {

struct EncodeError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
// A POD of a batch cannot be encoded (unparsable requests, a value outside the exact integer domain -- where the reference's
// .expect("invalid pod spec") panics, src/util.rs:65,68), or a single pod names more than KSCHED_MAX_KEYS nodeSelector keys.  Thrown by
// Snapshot::encode_pods and by check_node_validity_batch's key-budget walk only, i.e. BEFORE anything of the batch has been evaluated
// or POSTed: the one failure a batching caller may answer by retrying the batch's pods one at a time.
struct PodEncodeError : EncodeError {
    using EncodeError::EncodeError;
};

// RAII over the C ABI handle.
class DeviceEvaluator {
public:
    explicit DeviceEvaluator(int device);
    ~DeviceEvaluator();
    DeviceEvaluator(const DeviceEvaluator &) = delete;
    DeviceEvaluator &operator=(const DeviceEvaluator &) = delete;
    ksched_ctx *handle() const { return h_; }
    void check(int rc, const char *where) const;  // throws EncodeError with ksched_strerror text

private:
    ksched_ctx *h_ = nullptr;
};

// Encoded pod batch (the arguments of ksched_eval).
struct PodColumns {
    uint32_t p = 0, n_keys = 0;
    std::vector<int64_t> req_cpu_milli, req_mem_bytes;
    std::vector<uint32_t> sel_val_ids;  // [n_keys][p]
    std::vector<uint64_t> tolerations;  // [p]
    // the exact sums behind the two request columns, in nano-units (total_pod_resources, src/util.rs:54-75): what a snapshot update of the
    // batch's bindings subtracts from `available` -- kept so that the pods' quantity strings are parsed once per batch, not twice
    std::vector<__int128> req_cpu_nanos, req_mem_nanos;  // [p]
};

// Encoded node snapshot (the arguments of ksched_set_nodes), host copy.
struct NodeColumns {
    uint32_t n = 0, n_keys = 0;
    std::vector<std::string> names;            // canonical order
    std::vector<int64_t> avail_cpu_milli, avail_mem_bytes;
    std::vector<uint32_t> label_val_ids;       // [n_keys][n]
    std::vector<uint64_t> taints;              // [n]
    std::vector<std::string> keys;             // column k <-> label key
};

using TaintId = std::tuple<std::string, std::string, std::string>;  // key, value, effect

class ShardedContext;  // sharded.hpp: the batch over several devices

class Snapshot {
public:
    // device = HIP device index; kEncodeOnly builds the columns on the host and uploads nothing
    // (used to test the wire-format step where there is no GPU; such a snapshot cannot evaluate).
    static constexpr int kEncodeOnly = -1;
    explicit Snapshot(int device);
    // Several devices (one process, n MI355X: sharded.hpp): the snapshot is REPLICATED -- every ksched_set_nodes /
    // ksched_update_nodes goes to every device -- and batches are row-sharded over them by sharded().  `devices` must not be
    // empty; one device and force_sharded = false is exactly Snapshot(int).  force_sharded = true builds the ShardedContext (and
    // its one-rank RCCL communicator) for a single device too: the same code path on the one GPU a test box has.
    explicit Snapshot(const std::vector<int> &devices, bool force_sharded = false);
    ~Snapshot();

    // Encode `nodes` (any order) against the pods `client` LISTs per node and upload.
    // available[n] = allocatable[n] - sum(total_pod_resources(p) for p in LIST(n))
    // (src/predicates.rs:27-38).  A node whose allocatable map lacks cpu or memory, or whose
    // quantities do not parse, throws EncodeError (the reference panics there, :29-31).
    // with_resources = false skips allocatable and the LISTs (columns stay 0): the selector predicate
    // alone never reads them (src/predicates.rs:45-61).
    void rebuild(const std::vector<corev1::Node> &nodes, PodLister *client, bool with_resources = true);

    // Keep `available` current from pod watch events instead of one LIST per evaluation.
    // The reference subtracts total_pod_resources of every pod the LIST for the node returns
    // (src/predicates.rs:34-38); a pod that appears on / disappears from a node changes exactly that sum:
    //   apply_bound_pod   : pod.spec.nodeName now names a node of this snapshot -> available -= requests
    //   apply_deleted_pod : such a pod is gone from the API server              -> available += requests
    // Both patch the host columns and push the changed rows with ksched_update_nodes (only the node's 1024-node
    // tile is re-indexed on the device).  Return false (and change nothing) when the pod has no nodeName or the
    // node is not in this snapshot.  Arithmetic leaving int64 throws EncodeError.
    bool apply_bound_pod(const corev1::Pod &pod);
    bool apply_deleted_pod(const corev1::Pod &pod);
    // Many events, one device update.  second = true: bound, false: deleted.  Returns how many were applied.
    size_t apply_pod_events(const std::vector<std::pair<const corev1::Pod *, bool>> &events);

    // The same bookkeeping for callers that forward a pod WATCH STREAM as it comes ("incrementally maintained from
    // watch events"): the snapshot remembers which pods it counts against which node -- seeded by rebuild() from the LISTs -- so the
    // calls are idempotent.  Applied = the watch's Added / Modified (the pod as it is now), Deleted = its Deleted.
    //   Applied, spec.nodeName names a node of this snapshot : counted there with its current requests; if it was already counted with
    //                                                         the same node and requests (a repeated or unrelated MODIFIED event, or the
    //                                                         echo of a binding this process POSTed itself) nothing changes;
    //   Applied, no nodeName / a node outside the snapshot    : no longer counted anywhere;
    //   Deleted                                               : no longer counted (with the amounts that WERE counted, whatever the
    //                                                         event's object says).
    // One device update per call.  Returns how many events changed `available`.  Strong guarantee: on EncodeError (unparsable
    // requests, a change that is not an integer number of milli-cores / bytes, int64 overflow) nothing has changed.
    // apply_bound_pod / apply_deleted_pod / apply_pod_events above are the raw, untracked form (every call is applied).
    enum class PodEvent { Applied, Deleted };
    size_t observe_pods(const std::vector<std::pair<PodEvent, const corev1::Pod *>> &events);
    bool observe_pod(PodEvent kind, const corev1::Pod &pod) { return observe_pods({{kind, &pod}}) == 1; }
    // The bindings a batch has just created: pod i now runs on node_names[i] (the pod objects themselves still carry no nodeName --
    // no copies are made).  Same bookkeeping and guarantees as observe_pods with Applied events of those pods bound to those nodes.
    size_t observe_bound(const std::vector<std::pair<const corev1::Pod *, const std::string *>> &bound);
    // The same for a batch this snapshot has just evaluated: the node as its CANONICAL INDEX (what the device returned) and the pod's exact
    // requests as encode_pods summed them (PodColumns::req_*_nanos) -- no name lookup, no second parse of the quantity strings.
    struct Bound {
        const corev1::Pod *pod;
        uint32_t node;  // canonical index
        __int128 cpu_nanos, mem_nanos;
    };
    size_t observe_bound(const std::vector<Bound> &bound);
    size_t counted_pods() const { return counted_.size(); }

    // Make sure every label key in `keys` (the selector keys of ONE batch) has a column; re-uploads the label columns when the
    // column set changes.  Columns are a per-batch working set, not a lifetime dictionary: when adding the batch's keys would
    // exceed KSCHED_MAX_KEYS, the columns no pod of this batch uses are evicted.  Throws EncodeError only when one batch alone
    // uses more than KSCHED_MAX_KEYS distinct keys (check_node_validity_batch splits such a batch before it gets here).
    void ensure_keys(const std::set<std::string> &keys);
    // The selector keys of a pod (for callers that split batches by key budget).
    static void selector_keys(const corev1::Pod &pod, std::set<std::string> &into);

    // Extension E2 is opt-in: taints are interned to bit positions only when a caller asks for the taint predicate.  The
    // reference has no taint predicate, so a cluster with any number of distinct taints must schedule normally on the parity
    // path (FIT | SEL); only enable_taints() can fail with "more than 64 distinct taints".
    void enable_taints();
    bool taints_enabled() const { return taints_enabled_; }

    // Encode pods against this snapshot's dictionaries.  Adds columns for selector keys that
    // have none yet (ensure_keys).  A selector value no node carries becomes KSCHED_SEL_NEVER.
    // `known_keys`: the distinct selector keys of `pods` when the caller has collected them already (check_node_validity_batch's plan).
    PodColumns encode_pods(const std::vector<const corev1::Pod *> &pods, const std::set<std::string> *known_keys = nullptr);
    // The distinct nodeSelector keys of a batch, collected by the worker threads; *any_wide = some pod names more than KSCHED_MAX_KEYS keys.
    static std::set<std::string> batch_selector_keys(const std::vector<const corev1::Pod *> &pods, bool *any_wide = nullptr);

    const NodeColumns &columns() const { return cols_; }
    // The unit of the resource columns, in nano-units per column unit: 1 000 000 (milli-cores) and 1 000 000 000 (bytes) unless
    // some node's `available` is finer than that.  The reference compares decimals and accepts ANY quantity (src/util.rs:64-69,
    // src/predicates.rs:29-31: "100u" of CPU, "100m" of memory); the device compares exact int64 -- so the snapshot picks, per
    // resource, the COARSEST unit of {milli, micro, nano}-cores / {1, milli, micro, nano}-bytes in which every node's
    // `available` is a whole number that fits int64 (nano-cores reach 9.2e9 cores, milli-bytes 9.2 PB), re-picks it when a pod
    // event makes a value finer, and encodes a pod's request as ceil(request / unit): `available` being a whole number of
    // units, request <= available  <=>  ceil(request / unit) <= available / unit, exactly.  Refused (EncodeError): only what no
    // unit can hold -- finer than a nano-unit, or too large for int64 in the unit its fineness needs.
    __int128 cpu_unit_nanos() const { return cpu_unit_; }
    __int128 mem_unit_nanos() const { return mem_unit_; }
    int index_of(const std::string &node_name) const;  // canonical index or -1
    // canonical index <-> position in the vector given to rebuild() (the node store's own order)
    uint32_t store_index(uint32_t canonical) const { return store_of_canonical_[canonical]; }
    uint32_t canonical_index(uint32_t store) const { return canonical_of_store_[store]; }
    uint32_t n() const { return cols_.n; }
    uint32_t mask_words() const { return ksched_mask_words(cols_.n); }
    bool has_taints() const { return any_counted_taint_; }  // some node carries a NoSchedule / NoExecute taint
    DeviceEvaluator &device();  // the first device (per-pair calls, ksched_explain)
    size_t device_count() const { return devs_.size(); }
    // the batch path over every device of this snapshot, or nullptr (one device, not forced): callers then use device()
    ShardedContext *sharded();
    const std::map<TaintId, uint32_t> &taint_ids() const { return taint_ids_; }
    uint64_t generation() const { return generation_; }
    // A device call failed after the host columns had been committed (ksched_set_nodes / ksched_update_nodes returned an error):
    // the device may hold an older snapshot.  The next device() -- every evaluation goes through it -- uploads everything again.
    bool device_stale() const { return device_stale_; }

private:
    void encode_labels();
    void upload();

    std::shared_ptr<DeviceEvaluator> dev_;                 // == devs_[0] (nullptr: encode-only)
    std::vector<std::shared_ptr<DeviceEvaluator>> devs_;  // every device the snapshot is replicated to
    std::unique_ptr<ShardedContext> sharded_;
    NodeColumns cols_;
    std::vector<uint32_t> store_of_canonical_, canonical_of_store_;
    std::vector<corev1::StringMap> node_labels_;            // canonical order; empty map when labels is None
    std::vector<bool> node_has_labels_;
    std::vector<std::map<std::string, uint32_t>> value_ids_;  // per column: value string -> id (1..)
    std::map<TaintId, uint32_t> taint_ids_;                 // NoSchedule / NoExecute taints -> bit (filled by enable_taints)
    std::vector<std::vector<TaintId>> node_taints_raw_;     // canonical order: the node's counted taints, un-interned
    bool any_counted_taint_ = false, taints_enabled_ = false, device_stale_ = false;
    void intern_taints();
    static void intern_taints_into(const std::vector<std::vector<TaintId>> &raw, std::map<TaintId, uint32_t> &ids, std::vector<uint64_t> &column);
    uint64_t generation_ = 0;
    std::vector<__int128> avail_cpu_nanos_, avail_mem_nanos_;  // canonical order: the exact values behind the two resource columns
    __int128 cpu_unit_ = 1000000, mem_unit_ = 1000000000;      // nano-units per column unit (see cpu_unit_nanos())
    // new exact `available` values of some nodes -> columns (+ a new unit when one of them needs it) -> device.  Validates before it
    // changes anything (strong guarantee); `commit` runs between validation and the device call (the callers' own bookkeeping).
    void store_available(const std::vector<uint32_t> &nodes, const std::vector<std::pair<__int128, __int128>> &fresh, const std::function<void()> &commit);
    struct Counted {
        uint32_t node;  // canonical index
        __int128 cpu_nanos, mem_nanos;
    };
    // namespace/name -> what `available` currently holds against that pod.  Cut into shards by the key's hash so that a batch's worth of
    // events (50 000 bindings of a C3-size batch) is merged and committed by the worker threads, one shard each, instead of one serial walk.
    struct CountedTable {
        static constexpr size_t kShards = 64;
        std::array<std::unordered_map<std::string, Counted>, kShards> shard;
        static size_t hash_of(std::string_view key) { return std::hash<std::string_view>{}(key); }
        static size_t shard_of(size_t hash) { return (hash >> 7) % kShards; }  // (not the bits the maps' own buckets use)
        size_t size() const {
            size_t n = 0;
            for (const auto &m : shard) n += m.size();
            return n;
        }
        Counted &operator[](const std::string &key) { return shard[shard_of(hash_of(key))][key]; }
    };
    CountedTable counted_;
    void push_rows(const std::vector<uint32_t> &touched);
    // one event of observe_pods / observe_bound: the pod, and the node it runs on now (nullptr = none: deleted / unbound)
    struct Observed {
        const corev1::Pod *pod;
        const std::string *node;
        int node_index = -2;  // >= -1: the canonical index is known (-1 = none) and `node` is not looked up
        bool have_requests = false;
        __int128 cpu_nanos = 0, mem_nanos = 0;
    };
    size_t observe_impl(const std::vector<Observed> &events);

public:
    // observe_bound in two halves, so that a caller can do the first -- everything that only READS the snapshot: keys, lookups, the
    // per-node change, validation inputs -- while the batch's binding POSTs are still in flight, and the second, the commit, when it knows
    // they all landed.  stage_bound never changes the snapshot; commit_staged applies exactly what was staged (returns the number of events
    // that changed `available`) and throws EncodeError -- leaving the snapshot as it was -- when the snapshot has changed in between or
    // the change leaves the exact integer domain.  A caller whose POSTs did not all land drops the staged update and calls observe_bound
    // with the ones that did.
    struct StagedUpdate;
    std::shared_ptr<StagedUpdate> stage_bound(const std::vector<Bound> &bound);
    size_t commit_staged(StagedUpdate &staged);
    bool staged_is_current(const StagedUpdate &staged) const;  // the snapshot has not changed since `staged` was made (commit_staged would not refuse it for that)

private:
    std::shared_ptr<StagedUpdate> stage_impl(const std::vector<Observed> &events);
};

// K8s ToleratesTaint (extension E2): does toleration `t` tolerate taint `x`?
bool toleration_matches(const corev1::Toleration &t, const TaintId &x);

}  // namespace ksched_host
