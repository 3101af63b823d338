// predicates.hpp -- host mirror of the reference's `predicates` module (src/predicates.rs:1-80).
//
// Same names, argument meaning and results as the Rust items; every evaluation runs on the
// MI355X through the C ABI (include/ksched.h) -- there is no CPU implementation of a predicate in
// this file, and without a HIP device every function here throws EncodeError.
//
//   Rust                                                        here
//   ----------------------------------------------------------  -------------------------------------------
//   enum InvalidNodeReason {NotEnoughResources,                 enum class InvalidNodeReason (same order;
//        NodeSelectorMismatch}            src/predicates.rs:14-18   debug_name() == the {:?} text)
//   async fn can_pod_fit(&Pod,&Node,&Context) -> bool   :20-43  can_pod_fit(pod, node, ctx)
//   fn does_node_selector_match(&Pod,&Node) -> bool     :45-61  does_node_selector_match(pod, node)
//   pub async fn check_node_validity(..)                :63-77  check_node_validity(pod, node, ctx)
//        -> Result<(), InvalidNodeReason>                         -> Validity (nullopt == Ok(()))
//
// The per-pair entry points keep the reference's cost model on purpose (one LIST per call, :34):
// they exist so the reference's call sites and unit tests read the same.  The batched entry point
// check_node_validity_batch is the one the accelerated reconciler uses: all pods x all nodes of
// ctx.snapshot in one ksched_eval.
#pragma once
#include <cstdint>
#include <functional>
#include <optional>
#include <vector>

#include "corev1.hpp"
#include "encoder.hpp"
#include "util.hpp"

namespace ksched_host {
namespace predicates {

enum class InvalidNodeReason {
    NotEnoughResources,    // src/predicates.rs:16
    NodeSelectorMismatch,  // src/predicates.rs:17
    TaintNotTolerated,     // extension E2 only (BASELINE.json configs[4]); never produced by the reference's path
};
// #[derive(Debug)] text, as printed at src/main.rs:62
const char *debug_name(InvalidNodeReason r);

using Validity = std::optional<InvalidNodeReason>;  // Result<(), InvalidNodeReason>: nullopt == Ok(())

bool can_pod_fit(const corev1::Pod &pod, const corev1::Node &node, Context &ctx);
bool does_node_selector_match(const corev1::Pod &pod, const corev1::Node &node);
Validity check_node_validity(const corev1::Pod &pod, const corev1::Node &node, Context &ctx);

// Both masks of one batch against ctx.snapshot, pod-major (include/ksched.h conventions).
struct BatchValidity {
    uint32_t p = 0, n = 0, W = 0;
    uint32_t flags = 0;
    std::vector<uint64_t> feasible, fit;  // [p][W]
    std::vector<int32_t> binding;         // [p] when a pick was requested, canonical node index or -1
    // [p] the pods' exact request sums in nano-units as the encoder computed them (empty for rows a wide pod's key groups produced: those
    // hold 0 and `exact_requests` is false): lets the caller update the snapshot with the batch's bindings without parsing the pods again
    std::vector<__int128> req_cpu_nanos, req_mem_nanos;
    bool exact_requests = false;

    bool is_valid(uint32_t pod, uint32_t node) const { return (feasible[(size_t)pod * W + (node >> 6)] >> (node & 63u)) & 1ull; }
    // check_node_validity's result for the pair, rebuilt in the reference's order (fit first)
    Validity validity(uint32_t pod, uint32_t node) const;
    uint64_t feasible_count(uint32_t pod) const;
};

// check_node_validity for every (pod, node) pair.  `pick_flags` may add KSCHED_PICK_SAMPLED (with
// `samples`, [p][attempts] canonical node indices) or KSCHED_PICK_BESTFIT; `taints` adds extension E2.
// want_masks = false (with a pick): bindings only -- `feasible` / `fit` stay empty, no mask kernel runs and nothing but the bindings
// comes back from the device (the reference's reconcile needs the chosen node, not the matrix: src/main.rs:53-66).
// `samples_ready`: called (once or more) right before the first device call that reads `samples` -- a caller that is still filling the
// draws on another thread while the batch is planned and encoded waits for that thread there (scheduler.cpp).
BatchValidity check_node_validity_batch(const std::vector<const corev1::Pod *> &pods, Context &ctx, bool taints = false,
                                        uint32_t pick_flags = 0, const std::vector<uint32_t> *samples = nullptr,
                                        uint32_t attempts = 0, bool want_masks = true, const std::function<void()> *samples_ready = nullptr);

// How check_node_validity_batch cuts a batch into device evaluations (no device involved: the host-side plan alone, what the CPU tests
// check).  The device takes KSCHED_MAX_KEYS label columns per call and the reference has no limit on selector keys
// (src/predicates.rs:48-53): pods [lo, hi) are one call whose distinct selector keys fit the budget; a pod with more keys than one call
// takes is a call of its own (hi == lo + 1) with `groups` = that pod once per group of at most KSCHED_MAX_KEYS keys, masks ANDed.
struct DeviceCall {
    size_t lo = 0, hi = 0;
    std::vector<corev1::Pod> groups;
};
std::vector<DeviceCall> device_calls(const std::vector<const corev1::Pod *> &pods);

// check_node_validity's result for listed (pod, node) pairs, decided pair by pair on the device (ksched_explain): what the
// reference logs at WARN for every rejected candidate (src/main.rs:62).  Unlike BatchValidity::validity (two masks) this tells a
// selector failure from a taint failure when the taint extension is on.  pairs[i] = {index into `pods`, CANONICAL node index}.
std::vector<Validity> explain_pairs(const std::vector<const corev1::Pod *> &pods, Context &ctx,
                                    const std::vector<std::pair<uint32_t, uint32_t>> &pairs, bool taints = false);

}  // namespace predicates
}  // namespace ksched_host
