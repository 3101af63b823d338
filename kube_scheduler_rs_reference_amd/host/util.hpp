// util.hpp -- host mirror of the reference's `util` module (src/util.rs:1-75).
//
// Same names and meaning as the Rust items they stand for:
//   Context              src/util.rs:12-15   {client, node_store}; gains the device snapshot
//   PodResources         src/util.rs:17-36   {cpu, memory} of ParsedQuantity, new() seeds "0", SubAssign
//   is_pod_bound         src/util.rs:38-45
//   full_name            src/util.rs:47-52
//   total_pod_resources  src/util.rs:54-75   containers only; requests only
// `kube::Client` is reduced to the one call the predicate path makes on it: the LIST of pods with
// field selector spec.nodeName=<node> (src/predicates.rs:21-25,34) -> PodLister.
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "corev1.hpp"
#include "quantity.hpp"

namespace ksched_host {

class Snapshot;

// The slice of kube::Api<Pod> the path uses (src/predicates.rs:21-34).
struct PodLister {
    virtual ~PodLister() = default;
    // every pod whose spec.nodeName == node_name, any phase (the reference applies no phase filter)
    virtual std::vector<corev1::Pod> list_pods_on_node(const std::string &node_name) = 0;
    uint64_t list_calls = 0;  // observability: how many LISTs were issued
};

// In-memory cluster state: what a fake API server would answer.
struct StaticPodLister : PodLister {
    std::vector<corev1::Pod> pods;
    std::vector<corev1::Pod> list_pods_on_node(const std::string &node_name) override;
};

struct PodResources {
    ParsedQuantity cpu, memory;
    PodResources();  // PodResources::new(): both "0" (src/util.rs:22-29)
    PodResources &operator-=(const PodResources &o) {  // impl SubAssign (src/util.rs:31-36)
        cpu -= o.cpu;
        memory -= o.memory;
        return *this;
    }
};

struct Context {
    std::shared_ptr<PodLister> client;     // pub client: Client
    std::vector<corev1::Node> node_store;  // pub node_store: reflector::Store<Node> (state() = this vector, any order)
    int device = 0;                        // HIP device the evaluator lives on ...
    // ... or several: the batch row-shards over them and the bindings are all-gathered (sharded.hpp).  Empty = $KSCHED_DEVICES
    // ("0,1,2,3" / "all"), and without that variable {device}.  $KSCHED_SHARDED=1 takes the sharded path with one device too.
    std::vector<int> devices;
    // Device-resident snapshot of node_store + LISTs (added field; SURVEY.md section 8b allows it).
    std::shared_ptr<Snapshot> snapshot;
    // one-node scratch snapshot used by the per-pair predicates::* entry points
    std::shared_ptr<Snapshot> pair_snapshot;

    // tracing's WARN level: where the reference's warn!() lines go -- one per rejected candidate (src/main.rs:62: "Node {} failed validity
    // check for pod {}: {:?}") and one per failed reconcile (:123).  The reference's subscriber prints everything up to INFO (:128), so the
    // default writes them to stderr; $KSCHED_LOG=off (or =error) or an empty function switches the level off, and then NOTHING is computed
    // for it: the batched path asks the device for the rejected draws' reasons (ksched_explain) only when somebody listens.
    std::function<void(const std::string &)> warn = default_warn_sink();

    // (Re)build `snapshot` from node_store and one LIST per node.
    void refresh_snapshot();

    static std::function<void(const std::string &)> default_warn_sink();
};

bool is_pod_bound(const corev1::Pod &pod);
std::string full_name(const corev1::ObjectMeta &meta);
PodResources total_pod_resources(const corev1::Pod &pod);

}  // namespace ksched_host
