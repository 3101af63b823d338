// batcher.hpp -- the collecting half of the batching reconciler.
//
// The reference's kube-rs Controller hands pending pods to `reconcile` one at a time, each as its own task
// (src/main.rs:141-144: Controller::new(pods, ..).run(reconcile, error_policy, ctx)).  The device path wants them in batches:
// PodBatcher sits where the Controller's scheduler sits and gives `ready_chunks(N)` semantics --
//   * push(pod): a pending pod arrived (watch event, or a requeue fired); while a pod of the same namespace/name is still
//     queued the new object REPLACES it in place (the Controller's scheduler keeps one pending request per object too:
//     a pod updated five times before it is reconciled is reconciled once, with its latest state);
//   * next_batch(): blocks until at least one pod is queued (or the batcher is closed), then returns everything that is
//     ready right now, at most max_pods, in arrival order -- it never waits to fill a batch: a lone pod is evaluated
//     alone, a burst of 100 000 goes out in ceil(100 000 / max_pods) calls;
//   * close(): no more pods will come; next_batch() drains what is queued and then returns an empty batch.
// run_batches() is the loop around it: batch -> reconcile function (reconcile_batch in the product) -> one outcome per pod
// handed to `done`.  What happens to a failed pod afterwards is the caller's error_policy (src/main.rs:122-125: requeue in
// five minutes = push() again later); nothing is re-queued here.
// Thread-safe: any number of producers, one or more consumers.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "corev1.hpp"
#include "scheduler.hpp"
#include "util.hpp"

namespace ksched_host {

class PodBatcher {
public:
    using PodPtr = std::shared_ptr<const corev1::Pod>;
    explicit PodBatcher(size_t max_pods);

    // false when the batcher is closed (the pod is dropped)
    bool push(PodPtr pod);
    void close();
    bool closed() const;
    size_t pending() const;
    uint64_t coalesced() const;  // pushes that replaced a queued pod instead of adding one

    // empty result <=> closed and drained
    std::vector<PodPtr> next_batch();
    // what is ready now (possibly nothing), never blocks
    std::vector<PodPtr> try_next_batch();

private:
    std::vector<PodPtr> take_locked();
    const size_t max_pods_;
    mutable std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::pair<uint64_t, PodPtr>> queue_;        // (ticket, pod) in arrival order; tickets are consecutive from head_ticket_
    std::unordered_map<std::string, uint64_t> ticket_of_;  // namespace/name -> ticket of its queued entry
    uint64_t next_ticket_ = 0, head_ticket_ = 0, coalesced_ = 0;
    bool closed_ = false;
};

struct BatchLoopStats {
    uint64_t batches = 0, pods = 0, largest = 0;
    uint64_t isolated_batches = 0, failed_pods = 0;  // batches that had to be reconciled pod by pod; pods handed to `failed`
};

// Pull batches until the batcher is closed and drained.  `reconcile` maps a batch to one outcome per pod (the product passes a
// lambda around reconcile_batch(pods, ctx, chooser, sink, post_concurrency)); `done` receives every (pod, outcome) pair.
//
// A pod the host cannot encode (an unparsable quantity in its requests: the reference's `.expect("invalid pod spec")` panics on it,
// src/util.rs:65,68) makes reconcile_batch throw for the batch it sits in, before anything is evaluated or POSTed.  With a `failed`
// callback the loop does not give up the whole batch: it reconciles that batch's pods one at a time, hands the offender(s) to
// `failed(pod, what)` and everybody else's outcome to `done` -- one bad object costs one batch its batching, not the other pods
// their scheduling.  Without the callback the exception propagates to the caller (the batch's pods are then not reconciled).
// Only that exception (PodEncodeError, thrown by Snapshot::encode_pods) is answered this way: any other one may come from AFTER
// the batch's POSTs, where reconciling the pods again would create their bindings a second time -- it propagates.
BatchLoopStats run_batches(PodBatcher &batcher,
                           const std::function<std::vector<ReconcileOutcome>(const std::vector<const corev1::Pod *> &)> &reconcile,
                           const std::function<void(const PodBatcher::PodPtr &, const ReconcileOutcome &)> &done,
                           const std::function<void(const PodBatcher::PodPtr &, const std::string &)> &failed = nullptr);

}  // namespace ksched_host
