// scheduler.hpp -- host mirror of the pick and reconcile path of the reference's main.rs.
//
//   Rust (src/main.rs)                                           here
//   -----------------------------------------------------------  ------------------------------------------
//   const ATTEMPTS: u32 = 5                                  :49  ATTEMPTS
//   async fn select_node_for_pod(&Pod,&Context)->Option<Node> :51  select_node_for_pod(pod, ctx, chooser)
//   async fn reconcile(Arc<Pod>, Arc<Context>)              :73  reconcile(pod, ctx, chooser, sink)
//   fn error_policy(..) -> Action::requeue(5 min)          :122  error_policy(...)
//   enum ReconcileError (src/error.rs:5-15)                      ReconcileError (same variants, same #[error] text)
//
// The reference draws candidates with rand::thread_rng() (src/main.rs:56), so its pick cannot be
// reproduced; here the draw is an injected NodeChooser (SliceRandom::choose over the node store's
// current state).  The batched entry point select_nodes_for_pods does the same thing for a whole
// batch in one device call (KSCHED_PICK_SAMPLED): every pod gets ATTEMPTS draws, the first
// feasible one wins, none -> no node (NoNodeFound).
#pragma once
#include <cstdint>
#include <functional>
#include <optional>
#include <string>
#include <vector>

#include "corev1.hpp"
#include "predicates.hpp"
#include "util.hpp"

namespace ksched_host {

constexpr uint32_t ATTEMPTS = 5;  // src/main.rs:49

// rand::seq::SliceRandom::choose on a slice of length n: an index in [0, n), or nothing when n == 0.
struct NodeChooser {
    virtual ~NodeChooser() = default;
    virtual std::optional<size_t> choose(size_t n) = 0;
};

// Deterministic chooser for tests and benchmarks: SplitMix64 stream.
struct SplitMixChooser : NodeChooser {
    uint64_t state;
    explicit SplitMixChooser(uint64_t seed) : state(seed) {}
    std::optional<size_t> choose(size_t n) override;
};

// Replays a fixed list of indices (golden tests: samples [3,3,7,1,0]).
struct ScriptedChooser : NodeChooser {
    std::vector<size_t> script;
    size_t next = 0;
    std::optional<size_t> choose(size_t n) override;
};

// what the reference logs at WARN for every rejected candidate (src/main.rs:62)
struct RejectedCandidate {
    std::string node_name;
    predicates::InvalidNodeReason reason;
};

// src/main.rs:51-71, one pod, per-pair predicate calls (reference cost model: one LIST per probe).
std::optional<corev1::Node> select_node_for_pod(const corev1::Pod &pod, Context &ctx, NodeChooser &chooser,
                                                std::vector<RejectedCandidate> *rejected = nullptr);

// The same pick for a batch against ctx.snapshot, one device call.  Draws are made up front, ATTEMPTS per
// pod in pod order (the reference stops drawing after the first success; with an injected chooser the
// results are identical draw for draw because unused draws do not influence the outcome).
// Returns for every pod the chosen node's index into ctx.node_store, or -1.
struct BatchSelection {
    std::vector<int32_t> node_store_index;             // [p] index into ctx.node_store or -1
    std::vector<std::vector<RejectedCandidate>> rejected;  // [p] candidates tried and refused, in order (filled on request)
    predicates::BatchValidity validity;                // the bindings; with want_rejected also both masks (canonical node order): without it no mask is computed or copied
    std::vector<uint32_t> samples;                     // [p][ATTEMPTS] the draws as canonical node indices (n = "no draw": empty store)
};
BatchSelection select_nodes_for_pods(const std::vector<const corev1::Pod *> &pods, Context &ctx, NodeChooser &chooser,
                                     bool want_rejected = false);

// The candidates a batch's pods tried and were refused, in draw order, WITHOUT the two masks: the draws ahead of each pod's winning one
// (all of them when none won) go to the device as (pod, node) pairs and come back with check_node_validity's reason (ksched_explain).  What
// select_node_for_pod logs at WARN (src/main.rs:62); costs one small device call per batch instead of two P x N masks.
std::vector<std::vector<RejectedCandidate>> explain_rejected(const std::vector<const corev1::Pod *> &pods, Context &ctx, const BatchSelection &sel);
// the reference's line for one of them: "Node {} failed validity check for pod {}: {:?}" (src/main.rs:62)
std::string rejected_line(const corev1::Pod &pod, const RejectedCandidate &r);
// emits them through ctx.warn, pod by pod, draw by draw (nothing happens -- and nothing is asked of the device -- when the level is off)
void warn_rejected(const std::vector<const corev1::Pod *> &pods, Context &ctx, const BatchSelection &sel);

// ---- reconcile ------------------------------------------------------------------------------------

enum class ReconcileError { CreateBindingFailed, CreateBindingObjectFailed, NoNodeFound };  // src/error.rs:5-15
const char *error_text(ReconcileError e);  // the #[error("...")] strings
const char *debug_name(ReconcileError e);  // #[derive(Debug)]: the variant's name, as error_policy prints it ("reconcile failed on pod {}: {:?}", src/main.rs:123)

// corev1::Binding as reconcile builds it (src/main.rs:83-91): the pod's metadata, target = node name
struct Binding {
    corev1::ObjectMeta metadata;
    std::string target_name;
};

// ctx.client.send(Binding::create_pod(..)) (src/main.rs:94-103): false = the POST failed
struct BindingSink {
    virtual ~BindingSink() = default;
    virtual bool create_pod_binding(const std::string &pod_name, const std::string &pod_namespace, const Binding &b) = 0;
};

enum class Action { AwaitChange, RequeueAfter5Min };  // Action::await_change() / Action::requeue(5 * 60 s)

struct ReconcileOutcome {
    bool ok = true;
    ReconcileError error = ReconcileError::NoNodeFound;  // valid when !ok
    Action action = Action::AwaitChange;                 // error_policy applied when !ok
    std::optional<std::string> bound_to;                 // node name when a binding was created
};

// src/main.rs:73-120 for one pod.
ReconcileOutcome reconcile(const corev1::Pod &pod, Context &ctx, NodeChooser &chooser, BindingSink &sink);

// The binding POSTs of a batch, up to `post_concurrency` in flight at once.  In the reference every
// reconcile is its own tokio task under the kube-rs Controller (src/main.rs:141-144), so the POSTs of pending pods
// (src/main.rs:94-103) overlap; a batch that POSTed one by one would serialise P network round trips behind a device step of
// tens of microseconds.  chosen[i] = the node picked for pods[i] or nullptr (-> NoNodeFound, src/main.rs:116-118).  Outcomes are
// per pod and do not depend on the order the POSTs complete in (the reference's have no order either).
// post_concurrency <= 1: the calling thread POSTs in batch order.  > 1: that many worker threads share the batch, and
// `sink.create_pod_binding` is called from several threads at once -- the sink must allow that.  A sink that throws is
// reported as CreateBindingFailed for that pod (nothing unwinds through the workers).
std::vector<ReconcileOutcome> post_bindings(const std::vector<const corev1::Pod *> &pods, const std::vector<const corev1::Node *> &chosen,
                                            BindingSink &sink, unsigned post_concurrency = 1);

// The batching reconciler: bound pods are skipped (src/main.rs:74-76), the rest go
// through ONE batched evaluation and pick, then each gets its own binding POST (post_bindings) and outcome.
std::vector<ReconcileOutcome> reconcile_batch(const std::vector<const corev1::Pod *> &pods, Context &ctx, NodeChooser &chooser,
                                              BindingSink &sink, unsigned post_concurrency = 1);

// In-batch capacity accounting -- OPT-IN and OUTSIDE the parity claim: the reference has no
// assume/reserve step (src/main.rs:78-119), so reconciles racing on one API-server state may over-commit a node,
// and reconcile_batch above reproduces exactly that.  This variant never over-commits:
//   round r: every still-pending pod is evaluated and picked ON THE DEVICE against the current snapshot (fresh
//   ATTEMPTS draws per round, as a requeued reconcile would make); per node only the FIRST pod (batch order) bound
//   to it in this round is accepted -- the device just said it fits what is available now -- its binding is
//   POSTed and Snapshot::apply_bound_pod shrinks the node's `available` (ksched_update_nodes); the other pods
//   that drew the same node go to the next round and are re-evaluated against the shrunk snapshot.
// A pod with no feasible draw in its round gets NoNodeFound (the reference's outcome, src/main.rs:117); pods still
// colliding after `max_rounds` get NoNodeFound too.  Every feasibility decision is the device's; the host only
// groups bindings by node.  The accepted bindings are a legal serial execution of the reference (round order,
// then batch order) in which each pod saw all earlier bindings.
struct SequentialStats {
    uint32_t rounds = 0;
    uint32_t conflicts = 0;  // (pod, round) pairs deferred because an earlier pod of the round took the node
};
std::vector<ReconcileOutcome> reconcile_batch_sequential(const std::vector<const corev1::Pod *> &pods, Context &ctx, NodeChooser &chooser,
                                                         BindingSink &sink, uint32_t max_rounds = 64, SequentialStats *stats = nullptr);

// src/main.rs:122-125: the requeue, and -- with a Context whose WARN level is on -- the reference's line "reconcile failed on pod {}: {:?}" (:123)
Action error_policy(const corev1::Pod &pod, ReconcileError error);
Action error_policy(const corev1::Pod &pod, ReconcileError error, Context &ctx);

}  // namespace ksched_host
