/*
 * oracle.h -- CPU restatement of the reference's predicate filter-and-pick path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kube_scheduler_rs_reference_amd/ (the product) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the reported CPU baseline.
 *
 * Provenance and pinning
 *   - Follows acrlabs/kube-scheduler-rs-reference: src/predicates.rs:14-77, src/util.rs:17-36,
 *     54-75, src/main.rs:49-71 (each function below cites its lines).
 *   - does_node_selector_match is PINNED by the reference's own tests
 *     (src/predicates/test.rs:42-58, KAT-S1..S3), replayed in tests/test_oracle_kat.py.
 *   - The resource-fit arithmetic lives in the un-vendored crate kube_quantity 0.6.1
 *     (Cargo.lock:787-797) over rust_decimal 1.30.0; the reference has no test on it and no Rust
 *     toolchain exists here, so resource-fit parity is UNPINNED ("parity unpinned").  This file
 *     restates the published Kubernetes quantity grammar with exact integer arithmetic
 *     (nano-units in a 128-bit integer); on the canonical domain D of SURVEY.md section 8c
 *     (CPU "<n>" / "<n>m", memory plain integer bytes, optionally Ki/Mi) every reading of the
 *     crate agrees with it.  Neither the build container nor the GPU box has cargo/rustc (profiles/
 *     r02_a_toolchain_probe_gpu_box.txt), so there is no oracle/_ref.  Pinning is one command for anyone who has cargo:
 *     rust/pin_parity.sh runs the reference's OWN fits() / does_node_selector_match on tests/golden/<name>_objects.json and
 *     tests/test_reference_fixtures.py compares the result with the fixtures this oracle reproduces.
 *     Outside D the unpinned half is BRACKETED: oracle_ref.py carries a second reading (kube_quantity 0.6.1 as the surveyor recalls
 *     it: f32 scale factors kept to 7 digits) and tests/test_quantity_readings.py states, per spelling, where the two agree (D, Ki,
 *     Mi, every decimal suffix) and where they do not (Gi and above, exponent forms); both fit masks are committed for the two
 *     fixtures that hold such spellings.  This C file implements the exact reading only.
 *   - Taints/tolerations and best-fit are extensions (BASELINE.json config 5) with no
 *     reference code: semantics are defined in DESIGN.md and restated here independently.
 */
#ifndef KSCHED_ORACLE_H
#define KSCHED_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exact quantity: value * 1e9 (nano-units) */
typedef __int128 ora_q;

/* error codes: the reference panics (expect/unwrap/index) in these situations */
#define ORA_OK 0
#define ORA_E_PARSE (-1)      /* src/util.rs:65,68 / src/predicates.rs:29,31: try_into().expect(...) */
#define ORA_E_MISSING_KEY (-2) /* src/predicates.rs:29-31: allocatable["cpu"] on a map without the key */
#define ORA_E_RANGE (-3)      /* value outside this oracle's 128-bit nano-unit range / not exactly representable */

/* InvalidNodeReason, src/predicates.rs:14-18 */
#define ORA_REASON_OK 0
#define ORA_REASON_NOT_ENOUGH_RESOURCES 1
#define ORA_REASON_NODE_SELECTOR_MISMATCH 2
#define ORA_REASON_TAINT_NOT_TOLERATED 3 /* extension */

/* predicate flags, same values as include/ksched.h */
#define ORA_FIT 0x01u
#define ORA_SEL 0x02u
#define ORA_TAINT 0x04u
#define ORA_PICK_SAMPLED 0x08u
#define ORA_PICK_BESTFIT 0x10u
#define ORA_WANT_FIT_MASK 0x20u
#define ORA_SEL_NEVER 0xFFFFFFFFu

/* ---- object model: the subset of corev1::Pod / corev1::Node the path reads -------------------- */
typedef struct { const char *key, *val; } ora_kv; /* maps are arrays sorted by key (BTreeMap order) */

typedef struct {
    int has_resources;  /* Container.resources is Some */
    int has_requests;   /* ResourceRequirements.requests is Some */
    const char *cpu;    /* requests.get("cpu"), NULL = absent */
    const char *memory; /* requests.get("memory"), NULL = absent */
} ora_container;

typedef struct { const char *key, *op, *value, *effect; } ora_toleration; /* NULL = field absent */
typedef struct { const char *key, *value, *effect; } ora_taint;

typedef struct {
    const char *ns, *name;
    int has_spec;
    uint32_t n_containers;
    const ora_container *containers;
    int has_node_selector;
    uint32_t n_sel;
    const ora_kv *sel;
    const char *node_name; /* spec.nodeName, NULL = unbound */
    uint32_t n_tol;
    const ora_toleration *tol;
} ora_pod;

typedef struct {
    const char *name;
    int has_labels;
    uint32_t n_labels;
    const ora_kv *labels;
    int has_status, has_allocatable;
    const char *alloc_cpu, *alloc_memory; /* NULL = key missing from the allocatable map */
    uint32_t n_taints;
    const ora_taint *taints;
} ora_node;

typedef struct { ora_q cpu, memory; } ora_resources; /* PodResources, src/util.rs:17-20 */

/* ---- object-level restatement ------------------------------------------------------------------- */
/* Kubernetes resource.Quantity text -> exact nano-units (the role of kube_quantity's TryFrom) */
int ora_parse_quantity(const char *s, ora_q *out);
/* i64 views used to compare with the encoded columns; ORA_E_RANGE when not exactly representable */
int ora_q_to_milli(ora_q q, int64_t *out);
int ora_q_to_units(ora_q q, int64_t *out);

/* src/util.rs:54-75 */
int ora_total_pod_resources(const ora_pod *pod, ora_resources *out);
/* src/predicates.rs:20-43 with the LIST of :34 injected: `pods_on_node` is what
 * Api::list(field_selector spec.nodeName=<node>) returned.  *fit receives the boolean of :42. */
int ora_can_pod_fit(const ora_pod *pod, const ora_node *node, const ora_pod *const *pods_on_node, uint32_t n_on_node,
                    int *fit);
/* the LIST itself, src/predicates.rs:21-25,34: every pod of `all` whose spec.nodeName equals the
 * node's name, any phase.  Writes up to cap pointers, returns the count. */
uint32_t ora_list_pods_on_node(const ora_pod *all, uint32_t n_all, const char *node_name, const ora_pod **out,
                               uint32_t cap);
/* src/predicates.rs:45-61 */
int ora_does_node_selector_match(const ora_pod *pod, const ora_node *node);
/* src/predicates.rs:63-77: returns ORA_REASON_* (>= 0) or a negative ORA_E_* where the reference panics */
int ora_check_node_validity(const ora_pod *pod, const ora_node *node, const ora_pod *const *pods_on_node,
                            uint32_t n_on_node);
/* src/main.rs:51-71 with the draws injected: samples[i] is the index `choose` returned on attempt
 * i.  Returns the chosen node index or -1 (None); negative ORA_E_* - 16 on a reference panic. */
int ora_select_node_for_pod(const ora_pod *pod, const ora_node *nodes, uint32_t n_nodes, const ora_pod *all_pods,
                            uint32_t n_all, const uint32_t *samples, uint32_t attempts);

/* extension E2: every NoSchedule/NoExecute taint of the node is tolerated by the pod */
int ora_tolerates_node_taints(const ora_pod *pod, const ora_node *node);

/* ---- batch driver on objects (CPU baseline "port") ---------------------------------------------
 * Evaluates every (pod, node) pair one at a time with string/map lookups, exactly as the
 * per-pair functions above do, against ONE snapshot: quantities are parsed once up front and the
 * per-node LIST is done once per node (both stated in DESIGN.md).  `bound` are all already-bound
 * pods of the cluster.  Masks are pod-major uint64 rows of ceil(n_nodes/64) words.
 * threads <= 0 -> all cores (OpenMP).  Returns 0 or a negative ORA_E_*.
 */
int ora_eval_objects(const ora_pod *pods, uint32_t n_pods, const ora_node *nodes, uint32_t n_nodes,
                     const ora_pod *bound, uint32_t n_bound, uint32_t flags, uint64_t *out_feasible,
                     uint64_t *out_fit, int threads);

/* ---- encoded-level restatement (same contract as include/ksched.h ksched_eval) ------------------ */
int ora_eval_encoded(uint32_t n, const int64_t *avail_cpu, const int64_t *avail_mem, const uint32_t *label_ids,
                     uint32_t n_keys, const uint64_t *taints, uint32_t p, const int64_t *req_cpu,
                     const int64_t *req_mem, const uint32_t *sel_ids, const uint64_t *tolerations,
                     const uint32_t *samples, uint32_t attempts, uint32_t flags, uint64_t *out_feasible,
                     uint64_t *out_fit, int32_t *out_binding, int threads);

int ora_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
