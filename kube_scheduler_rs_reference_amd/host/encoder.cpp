#include "encoder.hpp"
#include "pool.hpp"

#include "sharded.hpp"

#include <cstdlib>
#include <optional>
#include <string_view>

#include <algorithm>
#include <numeric>
#include <thread>

namespace ksched_host {

// ---- the unit of a resource column (encoder.hpp: cpu_unit_nanos) ------------------------------------------------------------------
namespace {
constexpr __int128 kCpuUnits[] = {1000000, 1000, 1};            // milli-, micro-, nano-cores
constexpr __int128 kMemUnits[] = {1000000000, 1000000, 1000, 1};  // bytes, milli-, micro-, nano-bytes
bool fits_i64(__int128 v) { return v >= (__int128)INT64_MIN && v <= (__int128)INT64_MAX; }
// the coarsest unit in which every value is a whole number that fits int64; 0 = there is none
template <size_t K>
__int128 choose_unit(const std::vector<__int128> &values, const __int128 (&units)[K]) {
    for (const __int128 u : units) {
        bool ok = true;
        for (const __int128 v : values)
            if (v % u != 0 || !fits_i64(v / u)) {
                ok = false;
                break;
            }
        if (ok) return u;
    }
    return 0;
}
// ceil(a / b) for b > 0, any sign of a
__int128 ceil_div(__int128 a, __int128 b) {
    __int128 q = a / b;
    if (a % b > 0) ++q;
    return q;
}
}  // namespace

DeviceEvaluator::DeviceEvaluator(int device) {
    int rc = ksched_create(&h_, device);
    if (rc != KSCHED_OK) throw EncodeError(std::string("ksched_create: ") + ksched_strerror(rc));
}
DeviceEvaluator::~DeviceEvaluator() { ksched_destroy(h_); }
void DeviceEvaluator::check(int rc, const char *where) const {
    if (rc != KSCHED_OK)
        throw EncodeError(std::string(where) + ": " + ksched_strerror(rc) + " (" + ksched_last_error(h_) + ")");
}

Snapshot::Snapshot(int device) {
    if (device != kEncodeOnly) {
        dev_ = std::make_shared<DeviceEvaluator>(device);
        devs_.push_back(dev_);
    }
}

Snapshot::Snapshot(const std::vector<int> &devices, bool force_sharded) {
    if (devices.empty()) throw EncodeError("Snapshot: no device given");
    for (int d : devices) devs_.push_back(std::make_shared<DeviceEvaluator>(d));
    dev_ = devs_[0];
    if (devs_.size() > 1 || force_sharded) sharded_ = std::make_unique<ShardedContext>(devs_);  // (throws when the RCCL communicator cannot be built)
}

Snapshot::~Snapshot() = default;  // (here, where ShardedContext is complete)

ShardedContext *Snapshot::sharded() {
    if (!sharded_) return nullptr;
    if (device_stale_) upload();
    return sharded_.get();
}

DeviceEvaluator &Snapshot::device() {
    if (!dev_) throw EncodeError("this Snapshot was created encode-only (no device): evaluation is not possible");
    if (device_stale_) upload();  // an earlier device call failed after the host state had been committed: bring the device back in line
    return *dev_;
}

bool toleration_matches(const corev1::Toleration &t, const TaintId &x) {
    const auto &[key, value, effect] = x;
    if (t.effect && !t.effect->empty() && *t.effect != effect) return false;
    const std::string op = (t.operator_ && !t.operator_->empty()) ? *t.operator_ : "Equal";
    if (!t.key || t.key->empty()) return op == "Exists";  // empty key + Exists tolerates everything
    if (*t.key != key) return false;
    if (op == "Exists") return true;
    return op == "Equal" && t.value.value_or("") == value;
}

void Snapshot::rebuild(const std::vector<corev1::Node> &nodes, PodLister *client, bool with_resources) {
    // Everything is staged in locals and committed at the end: an EncodeError (a node or a LISTed pod that cannot be encoded, more
    // than 64 distinct taints with the extension on) leaves the snapshot -- host bookkeeping AND device -- exactly as it was.
    const uint32_t n = (uint32_t)nodes.size();
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return corev1::name_any(nodes[a].metadata) < corev1::name_any(nodes[b].metadata);
    });
    std::vector<uint32_t> canonical_of_store(n, 0u);
    for (uint32_t i = 0; i < n; ++i) canonical_of_store[order[i]] = i;
    NodeColumns c;
    CountedTable counted;
    std::vector<__int128> cpu_nanos(n, 0), mem_nanos(n, 0);
    c.n = n;
    c.names.resize(n);
    c.avail_cpu_milli.resize(n);
    c.avail_mem_bytes.resize(n);
    c.taints.assign(n, 0);
    std::vector<corev1::StringMap> labels(n);
    std::vector<bool> has_labels(n, false);
    std::vector<std::vector<TaintId>> taints_raw(n);
    bool any_taint = false;
    for (uint32_t i = 0; i < n; ++i) {
        const corev1::Node &node = nodes[order[i]];
        c.names[i] = corev1::name_any(node.metadata);
        // src/predicates.rs:27-32
        PodResources avail;
        if (with_resources && node.status && node.status->allocatable) {
            const auto &al = *node.status->allocatable;
            auto cpu = al.find("cpu"), mem = al.find("memory");
            if (cpu == al.end() || mem == al.end())
                throw EncodeError("node " + c.names[i] + ": allocatable lacks cpu or memory (reference panics, src/predicates.rs:29-31)");
            try {
                avail.cpu = ParsedQuantity::try_from(cpu->second);
                avail.memory = ParsedQuantity::try_from(mem->second);
            } catch (const QuantityError &e) {
                throw EncodeError("node " + c.names[i] + ": invalid node spec: " + e.what());
            }
        }
        // src/predicates.rs:34-38: every pod the LIST returns is subtracted, any phase
        if (with_resources && client) {
            ++client->list_calls;
            for (const auto &p : client->list_pods_on_node(c.names[i])) {
                try {
                    const PodResources r = total_pod_resources(p);
                    avail -= r;
                    counted[full_name(p.metadata)] = Counted{i, r.cpu.nanos(), r.memory.nanos()};  // observe_pods() starts from the LISTs
                } catch (const QuantityError &e) {
                    throw EncodeError("pod " + full_name(p.metadata) + ": invalid pod spec: " + e.what());
                }
            }
        }
        cpu_nanos[i] = avail.cpu.nanos();
        mem_nanos[i] = avail.memory.nanos();
        if (node.metadata.labels) {
            labels[i] = *node.metadata.labels;
            has_labels[i] = true;
        }
        if (node.spec && node.spec->taints) {
            for (const auto &t : *node.spec->taints) {
                if (t.effect != "NoSchedule" && t.effect != "NoExecute") continue;  // PreferNoSchedule never filters
                taints_raw[i].emplace_back(t.key, t.value.value_or(""), t.effect);
                any_taint = true;
            }
        }
    }
    // the unit of each resource column: the coarsest one in which every node's `available` is a whole int64 number (encoder.hpp)
    const __int128 cpu_unit = choose_unit(cpu_nanos, kCpuUnits), mem_unit = choose_unit(mem_nanos, kMemUnits);
    if (!cpu_unit || !mem_unit) {
        for (uint32_t i = 0; i < n; ++i) {
            const std::vector<__int128> one_c{cpu_nanos[i]}, one_m{mem_nanos[i]};
            if (!choose_unit(one_c, kCpuUnits) || !choose_unit(one_m, kMemUnits))
                throw EncodeError("node " + c.names[i] + ": outside the exact integer domain: available does not fit int64 in any unit fine enough to hold it");
        }
        throw EncodeError("the nodes' available values do not fit int64 in one common unit (one node needs nano-units, another is too large for them)");
    }
    for (uint32_t i = 0; i < n; ++i) {
        c.avail_cpu_milli[i] = (int64_t)(cpu_nanos[i] / cpu_unit);
        c.avail_mem_bytes[i] = (int64_t)(mem_nanos[i] / mem_unit);
    }
    std::map<TaintId, uint32_t> ids;
    if (taints_enabled_) intern_taints_into(taints_raw, ids, c.taints);  // throws BEFORE anything is committed; the extension stays on across rebuilds
    // ---- commit (nothing below throws an EncodeError about the INPUT; a failing device call is remembered, see upload()) ----
    c.keys = cols_.keys;  // keep the label columns that were in use
    cols_ = std::move(c);
    avail_cpu_nanos_ = std::move(cpu_nanos);
    avail_mem_nanos_ = std::move(mem_nanos);
    cpu_unit_ = cpu_unit;
    mem_unit_ = mem_unit;
    counted_ = std::move(counted);
    store_of_canonical_ = std::move(order);
    canonical_of_store_ = std::move(canonical_of_store);
    node_labels_ = std::move(labels);
    node_has_labels_ = std::move(has_labels);
    node_taints_raw_ = std::move(taints_raw);
    any_counted_taint_ = any_taint;
    if (taints_enabled_) taint_ids_ = std::move(ids);
    encode_labels();
    upload();
}

// Extension E2: (key, value, effect) triples -> bit positions, at most 64 per snapshot.
void Snapshot::intern_taints_into(const std::vector<std::vector<TaintId>> &raw, std::map<TaintId, uint32_t> &ids, std::vector<uint64_t> &column) {
    ids.clear();
    column.assign(raw.size(), 0ull);
    for (size_t i = 0; i < raw.size(); ++i)
        for (const TaintId &id : raw[i]) {
            auto it = ids.find(id);
            if (it == ids.end()) {
                if (ids.size() >= 64) throw EncodeError("taint extension: more than 64 distinct NoSchedule/NoExecute taints in one snapshot");
                it = ids.emplace(id, (uint32_t)ids.size()).first;
            }
            column[i] |= 1ull << it->second;
        }
}

void Snapshot::intern_taints() {
    std::map<TaintId, uint32_t> ids;
    std::vector<uint64_t> column;
    intern_taints_into(node_taints_raw_, ids, column);  // (throws before anything changes)
    taint_ids_ = std::move(ids);
    cols_.taints = std::move(column);
}

void Snapshot::enable_taints() {
    if (taints_enabled_) return;
    intern_taints();  // "more than 64 distinct taints" leaves the extension off and the snapshot unchanged
    taints_enabled_ = true;
    if (!taint_ids_.empty()) upload();
}

void Snapshot::encode_labels() {
    const uint32_t n = cols_.n;
    cols_.n_keys = (uint32_t)cols_.keys.size();
    cols_.label_val_ids.assign((size_t)cols_.n_keys * n, 0u);
    value_ids_.assign(cols_.n_keys, {});
    for (uint32_t k = 0; k < cols_.n_keys; ++k) {
        auto &dict = value_ids_[k];
        for (uint32_t i = 0; i < n; ++i) {
            auto it = node_labels_[i].find(cols_.keys[k]);
            if (it == node_labels_[i].end()) continue;  // 0 = key absent on this node
            auto [d, fresh] = dict.emplace(it->second, (uint32_t)dict.size() + 1u);
            (void)fresh;
            cols_.label_val_ids[(size_t)k * n + i] = d->second;  // "" is a value like any other: non-zero id
        }
    }
}

void Snapshot::upload() {
    ++generation_;
    if (!dev_) return;  // encode-only snapshot (host tests of the wire-format step)
    device_stale_ = true;  // until the calls below have succeeded: the host columns are ahead of the device(s)
    for (const auto &d : devs_)  // replicated: every device holds the whole snapshot (<= 2.8 MB at configs[4]); the calls do not wait for the devices
        d->check(ksched_set_nodes(d->handle(), cols_.n, cols_.avail_cpu_milli.data(), cols_.avail_mem_bytes.data(),
                                  cols_.n_keys ? cols_.label_val_ids.data() : nullptr, cols_.n_keys,
                                  (taints_enabled_ && !taint_ids_.empty()) ? cols_.taints.data() : nullptr),
                 "ksched_set_nodes");
    device_stale_ = false;
}

size_t Snapshot::apply_pod_events(const std::vector<std::pair<const corev1::Pod *, bool>> &events) {
    std::map<uint32_t, std::pair<__int128, __int128>> fresh_of;  // node -> its new exact `available`
    size_t applied = 0;
    for (const auto &[pod, bound] : events) {
        if (!pod->spec || !pod->spec->node_name) continue;
        const int idx = index_of(*pod->spec->node_name);
        if (idx < 0) continue;
        __int128 dc, dm;
        try {
            const PodResources r = total_pod_resources(*pod);  // the same sum the LIST loop subtracts (src/predicates.rs:37)
            dc = r.cpu.nanos();
            dm = r.memory.nanos();
        } catch (const QuantityError &e) {
            throw EncodeError("pod " + full_name(pod->metadata) + ": invalid pod spec: " + e.what());
        }
        auto it = fresh_of.find((uint32_t)idx);
        if (it == fresh_of.end()) it = fresh_of.emplace((uint32_t)idx, std::make_pair(avail_cpu_nanos_[(size_t)idx], avail_mem_nanos_[(size_t)idx])).first;
        it->second.first += bound ? -dc : dc;
        it->second.second += bound ? -dm : dm;
        ++applied;
    }
    if (fresh_of.empty()) return 0;
    std::vector<uint32_t> touched;
    std::vector<std::pair<__int128, __int128>> fresh;
    for (const auto &[node, v] : fresh_of) {
        touched.push_back(node);
        fresh.push_back(v);
    }
    store_available(touched, fresh, [] {});
    return applied;
}

// New exact `available` values of `nodes` (ascending, distinct).  When every one of them is a whole int64 number of the current
// units the columns are patched and the rows pushed (ksched_update_nodes); otherwise the units are picked again over ALL nodes, every
// row re-encoded and the whole snapshot uploaded (a pod with a "100u" request has landed on a node: rare, and a rebuild would do no
// less).  Throws EncodeError -- with nothing changed -- when no unit holds the new values.
void Snapshot::store_available(const std::vector<uint32_t> &nodes, const std::vector<std::pair<__int128, __int128>> &fresh, const std::function<void()> &commit) {
    bool same_units = true;
    for (const auto &[cpu, mem] : fresh)
        if (cpu % cpu_unit_ != 0 || mem % mem_unit_ != 0 || !fits_i64(cpu / cpu_unit_) || !fits_i64(mem / mem_unit_)) same_units = false;
    if (same_units) {
        // (a coarser unit may have become possible again; it is only looked for by the next rebuild -- the comparison is exact in any unit)
        commit();
        for (size_t i = 0; i < nodes.size(); ++i) {
            avail_cpu_nanos_[nodes[i]] = fresh[i].first;
            avail_mem_nanos_[nodes[i]] = fresh[i].second;
            cols_.avail_cpu_milli[nodes[i]] = (int64_t)(fresh[i].first / cpu_unit_);
            cols_.avail_mem_bytes[nodes[i]] = (int64_t)(fresh[i].second / mem_unit_);
        }
        push_rows(nodes);
        return;
    }
    std::vector<__int128> cpu_all = avail_cpu_nanos_, mem_all = avail_mem_nanos_;
    for (size_t i = 0; i < nodes.size(); ++i) {
        cpu_all[nodes[i]] = fresh[i].first;
        mem_all[nodes[i]] = fresh[i].second;
    }
    const __int128 cpu_unit = choose_unit(cpu_all, kCpuUnits), mem_unit = choose_unit(mem_all, kMemUnits);
    if (!cpu_unit || !mem_unit) {
        for (size_t i = 0; i < nodes.size(); ++i) {
            const std::vector<__int128> one_c{fresh[i].first}, one_m{fresh[i].second};
            if (!choose_unit(one_c, kCpuUnits) || !choose_unit(one_m, kMemUnits))
                throw EncodeError("node " + cols_.names[nodes[i]] + ": available leaves the int64 domain (in the unit its fineness needs)");
        }
        throw EncodeError("the nodes' available values do not fit int64 in one common unit after this change");
    }
    commit();
    avail_cpu_nanos_ = std::move(cpu_all);
    avail_mem_nanos_ = std::move(mem_all);
    cpu_unit_ = cpu_unit;
    mem_unit_ = mem_unit;
    for (uint32_t i = 0; i < cols_.n; ++i) {
        cols_.avail_cpu_milli[i] = (int64_t)(avail_cpu_nanos_[i] / cpu_unit_);
        cols_.avail_mem_bytes[i] = (int64_t)(avail_mem_nanos_[i] / mem_unit_);
    }
    upload();  // every row changed its scale: the whole snapshot goes up (pods are encoded against the new unit from now on)
}

// the changed rows of `available` go to the device in one ksched_update_nodes (only the touched 1024-node tiles are re-indexed)
void Snapshot::push_rows(const std::vector<uint32_t> &touched) {
    ++generation_;
    if (!dev_) return;  // encode-only snapshot
    if (device_stale_) {  // the device missed an earlier change: the whole snapshot goes up, these rows with it
        upload();
        return;
    }
    // The callers have committed their bookkeeping (counted_, cols_) by now.  If the device call fails the rows never arrived
    // (and the C ABI refuses evaluations until the next ksched_set_nodes): remember it, so that the next evaluation -- or the
    // next change -- uploads everything instead of trusting a device that is behind the host for good.
    device_stale_ = true;
    std::vector<int64_t> cpu(touched.size()), mem(touched.size());
    for (size_t i = 0; i < touched.size(); ++i) {
        cpu[i] = cols_.avail_cpu_milli[touched[i]];
        mem[i] = cols_.avail_mem_bytes[touched[i]];
    }
    for (const auto &d : devs_) d->check(ksched_update_nodes(d->handle(), (uint32_t)touched.size(), touched.data(), cpu.data(), mem.data()), "ksched_update_nodes");
    device_stale_ = false;
}

size_t Snapshot::observe_pods(const std::vector<std::pair<PodEvent, const corev1::Pod *>> &events) {
    std::vector<Observed> ev;
    ev.reserve(events.size());
    for (const auto &[kind, pod] : events)
        if (pod) ev.push_back({pod, (kind == PodEvent::Applied && pod->spec && pod->spec->node_name) ? &*pod->spec->node_name : nullptr});
    return observe_impl(ev);
}

size_t Snapshot::observe_bound(const std::vector<std::pair<const corev1::Pod *, const std::string *>> &bound) {
    std::vector<Observed> ev;
    ev.reserve(bound.size());
    for (const auto &[pod, node] : bound)
        if (pod) ev.push_back({pod, node});
    return observe_impl(ev);
}

size_t Snapshot::observe_bound(const std::vector<Bound> &bound) {
    const std::shared_ptr<StagedUpdate> staged = stage_bound(bound);
    return commit_staged(*staged);
}

// What observe_impl has worked out before it touches the snapshot (encoder.hpp: stage_bound / commit_staged).
struct Snapshot::StagedUpdate {
    struct Pre {
        std::string key;
        size_t hash = 0;
        int idx = -1;
        __int128 cpu = 0, mem = 0;
        std::string error;
    };
    struct Entry {
        std::optional<Counted> entry;  // nullopt = not counted any more
        size_t last = 0;               // the latest event of this key: its string is MOVED into the bookkeeping at the commit
    };
    std::vector<Pre> pre;  // one per event; never resized once filled (the maps below hold views into its strings)
    std::array<std::unordered_map<std::string_view, Entry>, CountedTable::kShards> staged;  // per shard: key -> its entry after these events
    std::vector<uint32_t> touched;                      // nodes whose `available` changes, ascending
    std::vector<std::pair<__int128, __int128>> fresh;   // their new exact values
    size_t changed = 0;
    uint64_t generation = 0;  // of the snapshot the update was staged against
    bool committed = false;
};

std::shared_ptr<Snapshot::StagedUpdate> Snapshot::stage_impl(const std::vector<Observed> &events) {
    // Stage everything first (the new bookkeeping entries, the exact per-node change in nano-units), validate, then commit: an
    // EncodeError leaves the snapshot as it was.  Three passes, the first two on the worker threads for a batch's worth of events:
    //   1. per event: the pod's key + its hash, the node's index, the sum of its requests (quantity parsing unless the caller has it);
    //   2. per SHARD of the bookkeeping (key hash): the shard's events in event order against the shard's table -- what is counted
    //      now, what will be, the change per node;
    //   3. serial: the shards' per-node changes summed into the new `available` values.
    using Pre = StagedUpdate::Pre;
    auto out = std::make_shared<StagedUpdate>();
    StagedUpdate &S = *out;
    S.generation = generation_;
    S.pre.resize(events.size());
    PhaseClock clock("observe");
    const uint32_t parts = events.size() >= 4096 ? WorkerPool::parts_for(events.size()) : 1u;
    WorkerPool::instance().run(events.size(), parts, [&](size_t lo, size_t hi, uint32_t) {
        for (size_t i = lo; i < hi; ++i) {
            const Observed &e = events[i];
            Pre &p = S.pre[i];
            p.key = full_name(e.pod->metadata);
            p.hash = CountedTable::hash_of(p.key);
            p.idx = e.node_index >= -1 ? e.node_index : (e.node ? index_of(*e.node) : -1);
            if (p.idx < 0) continue;
            if (e.have_requests) {  // (summed by encode_pods a moment ago)
                p.cpu = e.cpu_nanos;
                p.mem = e.mem_nanos;
                continue;
            }
            try {
                const PodResources r = total_pod_resources(*e.pod);  // the sum the LIST loop subtracts (src/predicates.rs:37)
                p.cpu = r.cpu.nanos();
                p.mem = r.memory.nanos();
            } catch (const QuantityError &x) {
                p.error = "pod " + p.key + ": invalid pod spec: " + x.what();
            }
        }
    });
    clock.lap("keys + requests of the events (threads)");
    // the events of each shard, in event order (a counting sort)
    constexpr size_t K = CountedTable::kShards;
    std::array<uint32_t, K + 1> first{};
    for (const Pre &p : S.pre) ++first[CountedTable::shard_of(p.hash) + 1];
    for (size_t k = 0; k < K; ++k) first[k + 1] += first[k];
    std::vector<uint32_t> by_shard(events.size());
    {
        std::array<uint32_t, K> fill{};
        for (size_t i = 0; i < S.pre.size(); ++i) {
            const size_t k = CountedTable::shard_of(S.pre[i].hash);
            by_shard[first[k] + fill[k]++] = (uint32_t)i;
        }
    }
    struct Change {
        uint32_t node;
        __int128 cpu, mem;
    };
    struct ShardOut {
        std::vector<Change> changes;
        size_t changed = 0;
        size_t error_at = SIZE_MAX;  // the first event of the shard whose requests do not parse
    };
    std::array<ShardOut, K> so;
    WorkerPool::instance().run(K, events.size() >= 4096 ? std::min<uint32_t>((uint32_t)K, WorkerPool::parts_for(events.size())) : 1u, [&](size_t klo, size_t khi, uint32_t) {
        for (size_t k = klo; k < khi; ++k) {
            auto &staged = S.staged[k];
            const auto &table = counted_.shard[k];
            ShardOut &o = so[k];
            staged.reserve(first[k + 1] - first[k]);
            for (uint32_t j = first[k]; j < first[k + 1]; ++j) {
                const size_t i = by_shard[j];
                const Pre &p = S.pre[i];
                std::optional<Counted> was;
                if (auto st = staged.find(std::string_view(p.key)); st != staged.end()) {
                    was = st->second.entry;
                } else if (auto it = table.find(p.key); it != table.end()) {
                    was = it->second;
                }
                if (p.idx < 0) {  // deleted, not bound, or bound to a node this snapshot does not hold: counted nowhere from now on
                    if (!was) continue;
                    o.changes.push_back({was->node, was->cpu_nanos, was->mem_nanos});
                    staged[std::string_view(p.key)] = StagedUpdate::Entry{std::nullopt, i};
                    ++o.changed;
                    continue;
                }
                if (!p.error.empty()) {
                    o.error_at = i;
                    break;  // (nothing of this update will be committed)
                }
                const Counted now{(uint32_t)p.idx, p.cpu, p.mem};
                if (was && was->node == now.node && was->cpu_nanos == now.cpu_nanos && was->mem_nanos == now.mem_nanos) continue;  // already counted
                if (was) o.changes.push_back({was->node, was->cpu_nanos, was->mem_nanos});
                o.changes.push_back({now.node, -now.cpu_nanos, -now.mem_nanos});
                staged[std::string_view(p.key)] = StagedUpdate::Entry{now, i};
                ++o.changed;
            }
        }
    });
    size_t error_at = SIZE_MAX;
    for (const ShardOut &o : so) error_at = std::min(error_at, o.error_at);
    if (error_at != SIZE_MAX) throw EncodeError(S.pre[error_at].error);  // (the first one in event order, like a sequential walk)
    clock.lap("events merged per shard of the bookkeeping (threads)");
    // node -> change of available (cpu, mem) in nano-units, ascending node order
    std::map<uint32_t, std::pair<__int128, __int128>> sparse;
    std::vector<std::pair<__int128, __int128>> tab;
    std::vector<uint32_t> nodes;
    const bool dense = events.size() >= 64 && cols_.n <= (1u << 22);
    if (dense) tab.assign(cols_.n, {0, 0});
    std::vector<uint8_t> seen(dense ? cols_.n : 0, 0);
    for (const ShardOut &o : so) {
        S.changed += o.changed;
        for (const Change &c : o.changes) {
            if (!dense) {
                auto &d = sparse[c.node];
                d.first += c.cpu;
                d.second += c.mem;
                continue;
            }
            if (!seen[c.node]) {
                seen[c.node] = 1;
                nodes.push_back(c.node);
            }
            tab[c.node].first += c.cpu;
            tab[c.node].second += c.mem;
        }
    }
    auto add = [&](uint32_t node, const std::pair<__int128, __int128> &d) {
        if (d.first == 0 && d.second == 0) return;
        S.touched.push_back(node);
        S.fresh.emplace_back(avail_cpu_nanos_[node] + d.first, avail_mem_nanos_[node] + d.second);
    };
    if (dense) {
        std::sort(nodes.begin(), nodes.end());
        for (uint32_t node : nodes) add(node, tab[node]);
    } else {
        for (const auto &[node, d] : sparse) add(node, d);
    }
    clock.lap("per-node change summed");
    return out;
}

bool Snapshot::staged_is_current(const StagedUpdate &S) const { return !S.committed && S.generation == generation_; }

size_t Snapshot::commit_staged(StagedUpdate &S) {
    if (S.committed) throw EncodeError("commit_staged: this update has been committed already");
    if (S.generation != generation_) throw EncodeError("commit_staged: the snapshot has changed since the update was staged");
    PhaseClock clock("observe");
    auto commit = [&] {  // (no lookup in `staged` after this: its keys are views into the strings moved here)
        size_t n = 0;
        for (const auto &m : S.staged) n += m.size();
        WorkerPool::instance().run(CountedTable::kShards, n >= 4096 ? std::min<uint32_t>((uint32_t)CountedTable::kShards, WorkerPool::parts_for(n)) : 1u, [&](size_t klo, size_t khi, uint32_t) {
            for (size_t k = klo; k < khi; ++k) {
                auto &table = counted_.shard[k];
                table.reserve(table.size() + S.staged[k].size());
                for (auto &[key, st] : S.staged[k]) {
                    if (st.entry) table.insert_or_assign(std::move(S.pre[st.last].key), *st.entry);
                    else table.erase(S.pre[st.last].key);
                }
            }
        });
        S.committed = true;
    };
    if (S.touched.empty()) {
        commit();  // (changes that cancel out: the table still moves)
        ++generation_;
    } else {
        store_available(S.touched, S.fresh, commit);  // validates first: on EncodeError nothing -- table, columns, device -- has changed
    }
    clock.lap("commit + columns + ksched_update_nodes");
    return S.changed;
}

std::shared_ptr<Snapshot::StagedUpdate> Snapshot::stage_bound(const std::vector<Bound> &bound) {
    std::vector<Observed> ev;
    ev.reserve(bound.size());
    for (const Bound &b : bound)
        if (b.pod) {
            Observed o{b.pod, nullptr};
            o.node_index = b.node < cols_.n ? (int)b.node : -1;
            o.have_requests = true;
            o.cpu_nanos = b.cpu_nanos;
            o.mem_nanos = b.mem_nanos;
            ev.push_back(o);
        }
    return stage_impl(ev);
}

size_t Snapshot::observe_impl(const std::vector<Observed> &events) {
    const std::shared_ptr<StagedUpdate> staged = stage_impl(events);
    return commit_staged(*staged);
}

bool Snapshot::apply_bound_pod(const corev1::Pod &pod) { return apply_pod_events({{&pod, true}}) == 1; }
bool Snapshot::apply_deleted_pod(const corev1::Pod &pod) { return apply_pod_events({{&pod, false}}) == 1; }

void Snapshot::selector_keys(const corev1::Pod &pod, std::set<std::string> &into) {
    if (pod.spec && pod.spec->node_selector)
        for (const auto &kv : *pod.spec->node_selector) into.insert(kv.first);
}

void Snapshot::ensure_keys(const std::set<std::string> &keys) {
    if (keys.size() > KSCHED_MAX_KEYS)
        throw EncodeError("one batch uses more than KSCHED_MAX_KEYS distinct nodeSelector keys (check_node_validity_batch splits such batches)");
    std::vector<std::string> missing;
    for (const auto &k : keys)
        if (std::find(cols_.keys.begin(), cols_.keys.end(), k) == cols_.keys.end()) missing.push_back(k);
    if (missing.empty()) return;
    if (cols_.keys.size() + missing.size() > KSCHED_MAX_KEYS) {
        // evict: keep only what this batch uses (columns are a working set; the dictionaries are rebuilt from the node labels)
        cols_.keys.assign(keys.begin(), keys.end());
    } else {
        cols_.keys.insert(cols_.keys.end(), missing.begin(), missing.end());
    }
    encode_labels();
    upload();
}

int Snapshot::index_of(const std::string &node_name) const {
    auto it = std::lower_bound(cols_.names.begin(), cols_.names.end(), node_name);
    if (it == cols_.names.end() || *it != node_name) return -1;
    return (int)(it - cols_.names.begin());
}

std::set<std::string> Snapshot::batch_selector_keys(const std::vector<const corev1::Pod *> &pods, bool *any_wide) {
    // every worker collects its range's keys (a handful of distinct strings), merged afterwards -- one serial walk over 100 k pods
    // inserting into one set was a third of encode_pods' time
    const uint32_t kthreads = pods.size() >= 4096 ? WorkerPool::parts_for(pods.size()) : 1u;
    std::vector<std::set<std::string>> part(std::max(1u, kthreads));
    std::vector<uint8_t> wide(part.size(), 0);
    WorkerPool::instance().run(pods.size(), kthreads, [&](size_t lo, size_t hi, uint32_t t) {
        std::set<std::string> &mine = part[t];
        const std::string *last = nullptr;  // (consecutive pods mostly name keys already seen: one comparison instead of a tree walk)
        for (size_t i = lo; i < hi; ++i) {
            const corev1::Pod &pod = *pods[i];
            if (!pod.spec || !pod.spec->node_selector) continue;
            if (pod.spec->node_selector->size() > KSCHED_MAX_KEYS) wide[t] = 1;
            for (const auto &kv : *pod.spec->node_selector) {
                if (last && *last == kv.first) continue;
                last = &*mine.insert(kv.first).first;
            }
        }
    });
    std::set<std::string> keys;
    for (auto &s : part) keys.insert(s.begin(), s.end());
    if (any_wide) *any_wide = std::find(wide.begin(), wide.end(), (uint8_t)1) != wide.end();
    return keys;
}

PodColumns Snapshot::encode_pods(const std::vector<const corev1::Pod *> &pods, const std::set<std::string> *known_keys) {
    PhaseClock clock("encode_pods");
    std::set<std::string> collected;
    if (!known_keys) collected = batch_selector_keys(pods);
    const std::set<std::string> &keys = known_keys ? *known_keys : collected;
    clock.lap("selector keys of the batch (threads)");
    ensure_keys(keys);

    PodColumns pc;
    pc.p = (uint32_t)pods.size();
    pc.n_keys = cols_.n_keys;
    pc.req_cpu_milli.resize(pc.p);
    pc.req_mem_bytes.resize(pc.p);
    pc.sel_val_ids.assign((size_t)pc.n_keys * pc.p, 0u);
    pc.tolerations.assign(pc.p, 0ull);
    pc.req_cpu_nanos.resize(pc.p);
    pc.req_mem_nanos.resize(pc.p);
    // column index of every key once (the batch's keys all have a column now)
    std::map<std::string, uint32_t> col_of;
    for (uint32_t k = 0; k < cols_.n_keys; ++k) col_of.emplace(cols_.keys[k], k);
    auto encode_range = [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; ++i) {
            const corev1::Pod &pod = *pods[i];
            try {
                const PodResources r = total_pod_resources(pod);  // src/predicates.rs:40
                // ceil(request / unit): `available` is a whole number of units, so request <= available <=> ceil(request / unit) <= available / unit
                const __int128 qc = ceil_div(r.cpu.nanos(), cpu_unit_), qm = ceil_div(r.memory.nanos(), mem_unit_);
                if (!fits_i64(qc) || !fits_i64(qm)) throw QuantityError("the request does not fit int64 in the snapshot's unit");
                pc.req_cpu_milli[i] = (int64_t)qc;
                pc.req_mem_bytes[i] = (int64_t)qm;
                pc.req_cpu_nanos[i] = r.cpu.nanos();
                pc.req_mem_nanos[i] = r.memory.nanos();
            } catch (const QuantityError &e) {
                throw PodEncodeError("pod " + full_name(pod.metadata) + ": invalid pod spec: " + e.what());
            }
            if (pod.spec && pod.spec->node_selector) {
                for (const auto &[k, v] : *pod.spec->node_selector) {  // src/predicates.rs:48-53
                    const uint32_t col = col_of.at(k);
                    auto it = value_ids_[col].find(v);
                    pc.sel_val_ids[(size_t)col * pc.p + i] = (it == value_ids_[col].end()) ? KSCHED_SEL_NEVER : it->second;
                }
            }
            if (taints_enabled_ && pod.spec && pod.spec->tolerations) {
                for (const auto &[id, bit] : taint_ids_)
                    for (const auto &t : *pod.spec->tolerations)
                        if (toleration_matches(t, id)) {
                            pc.tolerations[i] |= 1ull << bit;
                            break;
                        }
            }
        }
    };
    // The wire-format step is per-pod string work (quantity parsing, dictionary lookups): for a large batch it is what the host
    // spends its time on, and the pods are independent -- fan it out over threads (each writes only its own rows).
    clock.lap("ensure_keys + columns allocated");
    // (a PodEncodeError of the lowest pod range is the one rethrown, like the sequential walk would have raised first)
    WorkerPool::instance().run(pc.p, pc.p >= 4096u ? WorkerPool::parts_for(pc.p) : 1u,
                               [&](size_t lo, size_t hi, uint32_t) { encode_range((uint32_t)lo, (uint32_t)hi); });
    clock.lap("requests + selector ids (threads)");
    return pc;
}

}  // namespace ksched_host
