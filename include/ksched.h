/*
 * ksched.h -- C ABI of the MI355X-native batched pods x nodes predicate evaluator.
 *
 * This is the drop-in boundary for ONE path of acrlabs/kube-scheduler-rs-reference: the
 * per-pod predicate filter-and-pick.  The reference has no FFI of its own (it is a pure Rust
 * binary), so each entry point below names the reference code whose work it replaces.  All
 * file:line citations are into the reference checkout.
 *
 *   reference                                            replaced by
 *   ---------------------------------------------------  ------------------------------------------
 *   node_store snapshot + per-evaluation LIST            ksched_set_nodes   (columns of `available`)
 *     src/main.rs:56, src/predicates.rs:21-38            ksched_update_nodes (sparse rows, from watch events)
 *   can_pod_fit            src/predicates.rs:20-43       KSCHED_FIT   bit of ksched_eval*
 *   does_node_selector_match  src/predicates.rs:45-61    KSCHED_SEL   bit of ksched_eval*
 *   check_node_validity    src/predicates.rs:63-77       feasible = fit AND sel (+ fit mask for the reason)
 *   select_node_for_pod    src/main.rs:49-71             KSCHED_PICK_SAMPLED (injected sample indices), ksched_pick_device
 *   (extension E1, BASELINE.json config 5)               KSCHED_PICK_BESTFIT
 *   (extension E2, BASELINE.json config 5)               KSCHED_TAINT
 *
 * Conventions
 *   - plain C, no C++ or torch types; nothing ever unwinds across this boundary: every failure
 *     is a negative return code (ksched_strerror gives the text).
 *   - all quantities are exact signed 64-bit integers on the canonical domain: CPU in
 *     milli-cores, memory in bytes.  `available` may be negative (src/predicates.rs:37).
 *   - nodes are in canonical order (ascending node name); node index == column index.
 *   - label columns are structure-of-arrays: label_val_ids[k*n + node] is the dictionary id of
 *     the value of key k on that node, 0 = key absent.  Ids are exact (interned), never hashes.
 *     sel_val_ids[k*p + pod]: 0 = pod does not constrain key k, KSCHED_SEL_NEVER = the pod asks
 *     for a value no node carries, otherwise the required id.
 *   - masks are pod-major: row `pod` has ksched_mask_words(n) uint64 words, bit (node % 64) of
 *     word (node / 64); padding bits of the last word are always zero.
 *   - the *_device entry points take device pointers (HBM resident, e.g. torch tensors'
 *     data_ptr()) and a hipStream_t passed as void*; they enqueue and return without syncing.
 *     The host-pointer entry points copy in, run, copy out and synchronise.
 *   - a ksched_ctx is internally serialised by a mutex (host side); use one ctx per device.  Evaluations may be enqueued on
 *     any number of the caller's streams: the library orders them against snapshot changes and against each other's use
 *     of ctx-owned scratch memory with events.  Tell it before a stream is destroyed (ksched_forget_stream).
 *   - the *_device entry points are not meant to be captured into a hipGraph and replayed: an evaluation may allocate scratch on its
 *     first use, and some of its scratch is chosen per CALL on the host (the two-stage best-fit pick rotates over three sets of
 *     hand-over counters, each call zeroing the next call's set instead of paying a memset launch).  What a graph would save -- the
 *     launch gap between consecutive steps -- is what ksched_pipe's "alternate" mode addresses (KSCHED_OPT_PIPE_MODE).
 */
#ifndef KSCHED_H
#define KSCHED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSCHED_ABI_VERSION 6u

/* at most this many label-key columns per batch */
#define KSCHED_MAX_KEYS 32u
/* reference: const ATTEMPTS: u32 = 5 (src/main.rs:49); the ABI accepts any value up to this */
#define KSCHED_MAX_ATTEMPTS 64u
/* selector id meaning "a value that no node carries" -> never matches */
#define KSCHED_SEL_NEVER 0xFFFFFFFFu

/* return codes */
#define KSCHED_OK 0
#define KSCHED_E_INVAL (-1)       /* bad argument (null pointer, inconsistent sizes, bad flags) */
#define KSCHED_E_NODEVICE (-2)    /* no usable MI355X / HIP device; there is NO CPU fallback */
#define KSCHED_E_HIP (-3)         /* a HIP runtime call failed; see ksched_last_error */
#define KSCHED_E_NOMEM (-4)       /* host or device allocation failed */
#define KSCHED_E_STATE (-5)       /* ksched_set_nodes has not been called */
#define KSCHED_E_UNSUPPORTED (-6) /* request outside what this build supports */
#define KSCHED_E_RCCL (-7)        /* RCCL unavailable or a collective call failed; see ksched_comm_last_error */

/* predicate / output selection flags for ksched_eval* */
#define KSCHED_FIT 0x01u           /* resource fit: req <= available on cpu AND memory */
#define KSCHED_SEL 0x02u           /* nodeSelector exact label match */
#define KSCHED_TAINT 0x04u         /* (taints[n] & ~tolerations[p]) == 0   (extension E2) */
#define KSCHED_PICK_SAMPLED 0x08u  /* first feasible of the injected samples, else -1 */
#define KSCHED_PICK_BESTFIT 0x10u  /* lexicographic min (mem residual, cpu residual, node) (extension E1) */
#define KSCHED_WANT_FIT_MASK 0x20u /* also write the fit-only mask (to rebuild InvalidNodeReason) */

/* InvalidNodeReason, src/predicates.rs:14-18 (variant order kept); 0 = Ok(()) */
#define KSCHED_REASON_OK 0
#define KSCHED_REASON_NOT_ENOUGH_RESOURCES 1
#define KSCHED_REASON_NODE_SELECTOR_MISMATCH 2
#define KSCHED_REASON_TAINT_NOT_TOLERATED 3 /* extension E2 only; never produced without KSCHED_TAINT */

/* kernel selection for ksched_set_option(ctx, KSCHED_OPT_KERNEL, v) */
#define KSCHED_OPT_KERNEL 1
#define KSCHED_KERNEL_AUTO 0
#define KSCHED_KERNEL_DIRECT 1  /* lanes = nodes, compare-is-ballot; always applicable */
#define KSCHED_KERNEL_FUSED 3   /* LDS-resident per-tile bitmap index, one launch: rank search + row AND + store (default when applicable) */
/* KSCHED_OPT_TIMING: N > 0 = every N-th mask kernel launch carries hipEvents on its dispatch (1 = every launch; the events
 * cost ~1.5 us of launch gap each, so a throughput measurement samples with N > 1); 0 = off */
#define KSCHED_OPT_TIMING 2
/* KSCHED_OPT_DEBUG: ablation bits for kernel timing experiments (tools/); any non-zero value makes results invalid */
#define KSCHED_OPT_DEBUG 3

/* KSCHED_OPT_TRACE: 1 = the fused kernel records per-block phase timestamps (diagnostics; see ksched_trace_read) */
#define KSCHED_OPT_TRACE 4
#define KSCHED_TRACE_WORDS 8u /* uint64 words per block: t_entry, t_staged_issue, t_barrier, t_phase1, t_group0, t_loop_end, t_drained, xcc_id */

/* KSCHED_OPT_PICK_FROM_MASK: 1 = KSCHED_PICK_SAMPLED reads the candidates' bits back from the feasibility mask (after
 * the mask kernel); 0 (default) = it tests the drawn candidates directly from the pod and node columns, as the reference
 * does (src/main.rs:53-66: draw, check_node_validity(pod, candidate)), before and independently of the mask kernel.  Same
 * results either way; a bindings-only request (no output mask) then launches no mask kernel at all.
 * KSCHED_PICK_BESTFIT likewise: 0 (default) = candidates are tested in best-fit order from node columns kept in that
 * order, the mask row is only scanned for pods whose best node sits deep in the order; 1 = every candidate's bit is
 * looked up in the mask. */
#define KSCHED_OPT_PICK_FROM_MASK 5
/* KSCHED_OPT_INDEX_BUILD: how ksched_set_nodes fills the per-tile bitmap index: 0 (default) = HIP kernels from the columns in
 * HBM; 1 = the host code that specifies it (csrc/tile_index.hpp), uploaded.  Same tables bit for bit (ksched_index_checksum). */
#define KSCHED_OPT_INDEX_BUILD 6
/* KSCHED_OPT_BESTFIT_STAGES: the best-fit pick from the bitmaps in best-fit order runs in one stage (a wave per pod) or in two
 * (a lane per pod decides from the first 512 - 1024 candidates, a wave per pod finishes the rest): 0 (default) = two stages from
 * 24576 pods per call on (two dependent launches cost ~45 us whatever the batch; one stage ~20 us + 1 us per 1000 pods), 1 = always
 * one, 2 = always two.  Same bindings either way. */
#define KSCHED_OPT_BESTFIT_STAGES 7
/* KSCHED_OPT_SNAPSHOT_STREAM: where ksched_set_nodes / ksched_update_nodes (and the lazy best-fit rebuild) enqueue their device
 * work.  0 (default) = when evaluations have been enqueued on exactly ONE caller stream so far, onto that stream: the change is
 * ordered behind the evaluations already there and ahead of the next ones by the stream itself, with no event and no
 * cross-stream wait (a scheduler loop "pod events -> update -> evaluate" on one stream then never leaves it); with no or several
 * caller streams, onto the ctx's own stream, ordered against the callers' streams by events.  1 = always the ctx's own stream.
 * Same results either way: every evaluation sees the snapshot that was current when it was enqueued. */
#define KSCHED_OPT_SNAPSHOT_STREAM 8
/* KSCHED_OPT_FUSED_PICK: 1 (default) = when an evaluation asks for the feasibility mask AND the sampled pick and the fused mask
 * kernel runs, the pick rides in that launch (select_node_for_pod, src/main.rs:51-71): ONE kernel per step -- as long as the launch
 * is short enough for the pick to hide in its fill (a wave of the kernel has at most five rounds of 64 pods: up to 261 000 pods
 * against 5 000 nodes, 128 000 against 10 000); longer launches keep the pick as its own launch, which is cheaper there.  Two forms,
 * chosen by the request: tile tests -- every block tests the drawn candidates that lie in its own tile against the bitmap rows it
 * holds in LDS, the blocks of a pod combine through one atomic per eight pods (ATTEMPTS = 5 draws, no taint predicate, at most eight
 * label keys) -- or, otherwise, a few waves of every block test the pods' candidates from the node records while the block's tile is
 * staged.  0 = the pick is its own launch ahead of the mask kernel (k_select_sampled); 2 = rides, always as waves of the fill;
 * 3 = rides as tile tests or the call fails with KSCHED_E_UNSUPPORTED (2 and 3: whatever the launch's length).  Same bindings in every form. */
#define KSCHED_OPT_FUSED_PICK 9
/* KSCHED_OPT_FAULT: test hook for the "nothing unwinds across this boundary" rule.  value = kind | (skip << 8): after `skip`
 * further fault points (the places where a call enters the library's C++: snapshot calls, evaluations, explain, checksum) the
 * next one throws inside the call -- kind 1: std::bad_alloc, kind 2: std::runtime_error -- once.  That call must come back with
 * KSCHED_E_NOMEM / KSCHED_E_INVAL and ksched_last_error set; the process must not terminate.  0 = off (default). */
#define KSCHED_OPT_FAULT 10
/* KSCHED_OPT_PIPE_MODE: how ksched_pipe_submit spreads a batch over the pipe's streams.  0 (default) = split: the mask kernel
 * on the mask stream, the pick (and whatever the caller enqueues behind it, e.g. the all-gather of the bindings) on the pick
 * stream.  m >= 1 = alternate: the WHOLE evaluation of a slot -- one launch when the pick rides in the mask kernel -- goes onto
 * stream (slot mod k), k = max(2, m) <= KSCHED_PIPE_MAX_STREAMS: consecutive batches overlap (the next launch's blocks fill while
 * the previous one's blocks still store -- as far as the chip has room for them: KSCHED_OPT_GRID_CUS).  Same results either way;
 * ksched_pipe_wait / ksched_pipe_wait_mask order a consumer behind the slot's work, ksched_pipe_slot_stream names its stream. */
#define KSCHED_OPT_PIPE_MODE 11
#define KSCHED_PIPE_MAX_STREAMS 8u
/* KSCHED_OPT_GRID_CUS: how many of the chip's 256 compute units ONE fused mask launch may occupy (0, the default, = all of them;
 * else 8 .. 256).  A block of the fused kernel owns a compute unit (its tile index fills the LDS), and every block of a launch
 * spends the first microseconds filling it while nothing can be stored; a launch that takes the whole chip leaves no room for
 * the next batch's launch to do that in the meantime.  With the launches of k batches in flight on k streams (the pipe's
 * alternate mode) and 256 / k compute units each, one batch's fill hides under the others' stores.  A throughput setting:
 * a single batch's own latency grows.  Results do not depend on it. */
#define KSCHED_OPT_GRID_CUS 12
/* KSCHED_OPT_MASK_PROBE: candidates ksched_mask_alloc's probe-and-keep path allocates and times (1 .. 16, default 6; 1 = no probing). */
#define KSCHED_OPT_MASK_PROBE 13
/* KSCHED_OPT_ROUND_ORDER: which pods the fused mask kernel's waves take in which order.  A launch cuts the batch into rounds of 64 pods and has
 * chunks * 16 waves per tile working at once.  0 (default) = interleaved, wave-major: round g goes to stream g mod (chunks * 16), neighbouring
 * rounds to different blocks -- the chip writes one moving window of the mask; 2 = interleaved, chunk-major (a block's 16 waves take 16
 * neighbouring rounds); 1 = blocked: every wave owns one contiguous pod range (rounds 1 .. 5's order: chunks * 16 write streams megabytes apart).
 * Results do not depend on it (tests/test_gpu_fullsize.py runs all three); rates do: profiles/r06_round_order.md. */
#define KSCHED_OPT_ROUND_ORDER 14

typedef struct ksched_ctx ksched_ctx;

/* ---- lifetime -------------------------------------------------------------------------- */

/* Create an evaluator bound to HIP device `device_id`.  Fails with KSCHED_E_NODEVICE when the
 * process sees no GPU: the product path has no CPU backend by design. */
int ksched_create(ksched_ctx **out, int device_id);
void ksched_destroy(ksched_ctx *ctx);

uint32_t ksched_abi_version(void);
/* HIP devices the process sees (0 when there is none or the runtime cannot be initialised): what a host checks $KSCHED_DEVICES against */
int ksched_device_count(void);
const char *ksched_strerror(int code);
/* text of the last HIP failure on this ctx (empty string if none); valid until the next call */
const char *ksched_last_error(const ksched_ctx *ctx);
/* words per mask row for n nodes = ceil(n / 64) */
uint32_t ksched_mask_words(uint32_t n_nodes);
int ksched_set_option(ksched_ctx *ctx, int option, int64_t value);

/* ---- node snapshot ----------------------------------------------------------------------
 * Replaces, for a whole batch, what can_pod_fit recomputes per evaluation
 * (src/predicates.rs:27-38): available[n] = allocatable[n] - sum(requests of pods bound to n).
 * Host pointers; the library copies them (they may be reused when the call returns) and builds its device-side indexes with
 * HIP kernels; the call does not wait for the device.  Evaluations already enqueued keep reading the previous snapshot;
 * evaluations enqueued afterwards (on any stream) are ordered behind the build -- by the stream itself when the build rides the
 * one caller stream this ctx has seen, by events otherwise (KSCHED_OPT_SNAPSHOT_STREAM).
 *   avail_cpu_milli, avail_mem_bytes : [n] signed
 *   label_val_ids : [n_keys][n] or NULL when n_keys == 0; ids must be < KSCHED_SEL_NEVER
 *   taints        : [n] bit set of (interned) taints, or NULL = no taints
 */
int ksched_set_nodes(ksched_ctx *ctx, uint32_t n, const int64_t *avail_cpu_milli, const int64_t *avail_mem_bytes,
                     const uint32_t *label_val_ids, uint32_t n_keys, const uint64_t *taints);

/* Incremental form of the same step: `available` of `count` nodes changed -- a pod was
 * bound to or removed from them (src/predicates.rs:36-38 subtracts every pod the LIST returns; here the caller
 * keeps that sum current from watch events instead of re-LISTing).  node_index[i] is a canonical node index
 * (< ksched_num_nodes); the two value arrays hold the node's NEW available cpu / memory.  Labels and taints are
 * not touched (use ksched_set_nodes when the node set or its labels change).  A node listed twice takes its last values.
 * Cost: the new values are scattered into the columns and the fit part (rows, search trees, cnt tables) of the touched
 * 1024-node tiles is rebuilt by one kernel; updates of up to 16 nodes travel in kernel arguments (no copy).  The best-fit
 * order is only marked stale: the next KSCHED_PICK_BESTFIT request rebuilds it.  The host does not wait: evaluations already
 * enqueued on any stream this ctx has seen read the snapshot as it was, later ones the new one.  With ONE caller stream the
 * two kernels are enqueued on that stream (no event, no cross-stream wait: an [update + bindings-only pick] loop runs 32 us per
 * iteration instead of 49); with several, on the ctx's own stream, ordered by events (KSCHED_OPT_SNAPSHOT_STREAM).
 * If a HIP call fails midway the snapshot is invalidated (KSCHED_E_STATE until the next ksched_set_nodes).
 */
int ksched_update_nodes(ksched_ctx *ctx, uint32_t count, const uint32_t *node_index, const int64_t *avail_cpu_milli,
                        const int64_t *avail_mem_bytes);

/* The ctx remembers every stream a *_device evaluation was enqueued on (that is how snapshot changes are ordered against
 * them without a device-wide wait).  A caller that DESTROYS such a stream while the ctx lives must say so first; streams that
 * outlive the ctx (e.g. a framework's pooled streams) need nothing.  ksched_pipe_destroy does this for the pipe's streams. */
int ksched_forget_stream(ksched_ctx *ctx, void *hip_stream);

/* number of nodes / keys of the current snapshot (0 before ksched_set_nodes) */
uint32_t ksched_num_nodes(const ksched_ctx *ctx);
uint32_t ksched_num_keys(const ksched_ctx *ctx);

/* ---- evaluation, host buffers -------------------------------------------------------------
 * One call = check_node_validity (src/predicates.rs:63-77) for every (pod, node) pair of the
 * batch against the snapshot, plus optionally the pick of select_node_for_pod (src/main.rs:51-71).
 *   req_cpu_milli, req_mem_bytes : [p]   total_pod_resources (src/util.rs:54-75), encoded
 *   sel_val_ids   : [n_keys][p] or NULL (= no pod has a selector)
 *   tolerations   : [p] or NULL (= tolerate nothing)        -- only read with KSCHED_TAINT
 *   samples       : [p][attempts] node indices or NULL       -- only read with KSCHED_PICK_SAMPLED
 *                   an index >= n is treated as an infeasible draw
 *   out_feasible  : [p][W] or NULL
 *   out_fit       : [p][W] or NULL; requires KSCHED_WANT_FIT_MASK
 *   out_binding   : [p] or NULL; requires one of the KSCHED_PICK_* flags; -1 = no node
 * Predicates not selected in `flags` are treated as true.
 * The output arrays may be ordinary pageable memory (the copies run at the link's rate into it: 63 MB of mask in 1.2 ms on an MI355X box); keep them from
 * call to call -- a freshly allocated array is first touched BY the copy, which then takes four times as long (bench.py end_to_end.host_arrays_to_mask).
 */
int ksched_eval(ksched_ctx *ctx, uint32_t p, const int64_t *req_cpu_milli, const int64_t *req_mem_bytes,
                const uint32_t *sel_val_ids, const uint64_t *tolerations, const uint32_t *samples,
                uint32_t attempts, uint32_t flags, uint64_t *out_feasible, uint64_t *out_fit,
                int32_t *out_binding);

/* ---- one host thread, several devices: ksched_eval in two halves ---------------------------------
 * north_star: "the pod batch row-shards across the 8 GPUs of one node with an RCCL allgather of the resulting (pod -> node)
 * bindings".  The reference is ONE process (src/main.rs:127-152); a drop-in host therefore drives n devices from one thread: one
 * ksched_ctx per device, the node snapshot replicated to all of them (ksched_set_nodes / ksched_update_nodes on each), the batch's
 * pod rows cut with ksched_shard_bounds, and per batch
 *     for every device r : ksched_eval_begin(ctx[r], rows [lo_r, hi_r) ...)      copies in + evaluation enqueued, NO host wait
 *     ksched_allgather_bindings_local(comms, n, local[], gathered[], count_per_rank, streams[])
 *     ksched_eval_end(ctx[0], gathered[0], n * count_per_rank, host_table)       one copy of the whole table, one host wait
 *     ksched_eval_end(ctx[r], NULL, 0, NULL) for the others                      (their masks, if asked for, have landed)
 * host_table[r * count_per_rank + i] is the binding of pod row lo_r + i (rows past a shard's end are -1).
 * (kube_scheduler_rs_reference_amd/host/sharded.cpp and rust/src/ksched.rs `ShardedEvaluator` are this loop.)
 *
 * ksched_shard_bounds   the one definition of the row split: rank r owns rows [lo, hi) = [r * c, min(p, (r + 1) * c)),
 *                       c = count_per_rank = ceil(p / nranks); the last ranks may own fewer rows, or none.  Pure arithmetic.
 * ksched_eval_begin     ksched_eval without its last two steps (the copy of the bindings to the host, the host wait):
 *                       host pointers as in ksched_eval, for THIS ctx's rows only; `sel_val_ids` may point INTO the whole batch's
 *                       [n_keys][sel_stride] array (sel_val_ids = all + lo, sel_stride = P: column k of the shard starts
 *                       sel_stride entries after column k - 1's; sel_stride = p for a packed array).  Masks, when asked for, are
 *                       copied to out_feasible / out_fit ([p][W], packed) behind the evaluation on the same stream: they are
 *                       complete after ksched_eval_end.  With a pick the bindings stay ON THE DEVICE in a ctx-owned buffer of
 *                       max(p, binding_capacity) int32 -- entries [p, binding_capacity) are -1, the padding the all-gather of
 *                       unequal shards needs -- returned through *binding_dev; *hip_stream is the ctx's own stream, where all of
 *                       this was enqueued (pass both to ksched_allgather_bindings*).  p = 0 is allowed (an empty shard still
 *                       takes part in the exchange).  The input arrays and the two host mask arrays must stay alive and untouched
 *                       until ksched_eval_end has returned (the copies are asynchronous).  The device buffers are valid until the
 *                       next ksched_eval / ksched_eval_begin on the ctx.
 * ksched_gather_buffer  a ctx-owned device buffer of `count` int32 for the gathered table (grown on demand, reused).
 * ksched_eval_end       enqueue the copy of `count` int32 from `bindings_dev` (any device pointer: the gathered table, or
 *                       *binding_dev itself) to `out_host` on the ctx's stream -- skipped when out_host is NULL -- and wait for
 *                       the stream. */
void ksched_shard_bounds(uint32_t p, uint32_t nranks, uint32_t rank, uint32_t *lo, uint32_t *hi, uint32_t *count_per_rank);
int ksched_eval_begin(ksched_ctx *ctx, uint32_t p, const int64_t *req_cpu_milli, const int64_t *req_mem_bytes,
                      const uint32_t *sel_val_ids, uint32_t sel_stride, const uint64_t *tolerations, const uint32_t *samples,
                      uint32_t attempts, uint32_t flags, uint64_t *out_feasible, uint64_t *out_fit, uint32_t binding_capacity,
                      int32_t **binding_dev, void **hip_stream);
int ksched_gather_buffer(ksched_ctx *ctx, uint32_t count, int32_t **dev);
int ksched_eval_end(ksched_ctx *ctx, const int32_t *bindings_dev, uint32_t count, int32_t *out_host);

/* ---- evaluation, device buffers -----------------------------------------------------------
 * Same contract with every array already resident in HBM.  out_feasible may be NULL only when
 * no pick is requested and out_fit is given; when a pick is requested without out_feasible the
 * library uses an internal scratch mask.  `hip_stream` is a hipStream_t (NULL = default stream).
 */
int ksched_eval_device(ksched_ctx *ctx, uint32_t p, const int64_t *req_cpu_milli, const int64_t *req_mem_bytes,
                       const uint32_t *sel_val_ids, const uint64_t *tolerations, const uint32_t *samples,
                       uint32_t attempts, uint32_t flags, uint64_t *out_feasible, uint64_t *out_fit,
                       int32_t *out_binding, void *hip_stream);

/* Same, with the output masks pitched: row `pod` of out_feasible / out_fit starts at word
 * pod * mask_pitch_words (mask_pitch_words >= ksched_mask_words(n)); words [W, pitch) of a row are padding
 * and are written as zero or left untouched.  ksched_mask_pitch(n) rounds W up to a multiple of 16 words
 * (128 bytes), which keeps every row on cache-line boundaries: on MI355X the mask stream then runs at the
 * HBM write ceiling instead of ~70 % of it ("row pitch").  ksched_eval_device == this with
 * mask_pitch_words = W. */
int ksched_eval_device_pitched(ksched_ctx *ctx, uint32_t p, const int64_t *req_cpu_milli, const int64_t *req_mem_bytes,
                               const uint32_t *sel_val_ids, const uint64_t *tolerations, const uint32_t *samples,
                               uint32_t attempts, uint32_t flags, uint64_t *out_feasible, uint64_t *out_fit,
                               int32_t *out_binding, uint32_t mask_pitch_words, void *hip_stream);
uint32_t ksched_mask_pitch(uint32_t n_nodes);

/* ---- mask buffers allocated by the library (ABI 6) ---------------------------------------------
 * The reference never materialises a mask (check_node_validity answers one pair at a time, src/predicates.rs:63-77); the batched
 * path's mask is its own artefact and 92 % of its HBM traffic, and on MI355X the rate at which the mask kernel runs depends on the
 * PHYSICAL placement of the buffer it writes (two rates, 4 % apart at 100k x 5k and 20 % apart at 125k x 50k:
 * profiles/r05_bimodal_by_allocation.md, profiles/r06_mask_alloc.md).  A caller may keep allocating its own masks (any device pointer works);
 * ksched_mask_alloc hands out [p] rows at the pitch ksched_mask_pitch(n) gives, placed the way the measurements found fastest.
 *   how            : KSCHED_MASK_ALLOC_AUTO, or one specific path (for re-measuring the choice: tools/alloc_probe.py)
 *   out_pitch_words: receives the row pitch in 64-bit words (pass it to ksched_eval_device_pitched / ksched_pipe_submit); may be NULL
 * KSCHED_E_STATE before ksched_set_nodes (the pitch follows the node count); a snapshot with a different node count needs new masks.
 * ksched_mask_free waits for the device before it unmaps; ksched_destroy frees what the caller left. */
#define KSCHED_MASK_ALLOC_AUTO 0u
#define KSCHED_MASK_ALLOC_PLAIN 1u      /* hipMalloc */
#define KSCHED_MASK_ALLOC_PROBE 11u     /* probe-and-keep: KSCHED_OPT_MASK_PROBE hipMalloc candidates alive at once; the fused mask kernel is timed into each
                                         * (fit only, zero requests, the current snapshot); the fastest is kept, the others are freed */
/* Measurement paths: implemented in the TEST build of the library only (tests/cpp/hooks/libksched_hip.so = the shipped object code +
 * tests/cpp/test_hooks.cpp); the shipped library answers KSCHED_E_UNSUPPORTED.  tools/alloc_probe.py re-measures the choice with them; none
 * selects the fast placement, and HIP's virtual-memory API showed STALE READS on a mapping's first use right after a contiguous allocation was
 * freed in the same process (profiles/r06_mask_alloc.md section 3): AUTO and PROBE never use them. */
#define KSCHED_MASK_ALLOC_VMM 2u        /* hipMemCreate in one piece at the recommended granularity + hipMemAddressReserve + hipMemMap */
#define KSCHED_MASK_ALLOC_VMM_MIN 4u    /* the same at the minimum granularity */
#define KSCHED_MASK_ALLOC_CONTIGUOUS 5u /* hipExtMallocWithFlags(hipDeviceMallocContiguous): one physical range */
#define KSCHED_MASK_ALLOC_SCATTER_2M 8u  /* physical pieces of 2 MiB created one by one (a quarter more than needed), shuffled, mapped at consecutive addresses */
#define KSCHED_MASK_ALLOC_SCATTER_16M 9u /* the same with pieces of 16 MiB */
#define KSCHED_MASK_ALLOC_LAST 11u      /* (3, 6, 7, 10 -- 1 GiB-aligned VMM, uncached, a memory pool, 64 KiB pieces -- were measured in round 6 and removed) */
/* AUTO = PROBE for masks of at least KSCHED_MASK_PROBE_MIN_BYTES when the snapshot has its bitmap index, PLAIN otherwise: no allocation path selects
 * the fast placement, and below that size a buffer's own rate does not stand out of the run-to-run noise */
#define KSCHED_MASK_PROBE_MIN_BYTES 0x8000000u /* 128 MiB */
int ksched_mask_alloc(ksched_ctx *ctx, uint32_t p, uint32_t how, uint64_t **out_mask, uint32_t *out_pitch_words);
int ksched_mask_free(ksched_ctx *ctx, uint64_t *mask);
/* What the latest probe-and-keep allocation of this ctx measured: microseconds per mask kernel launch into each candidate, in candidate order
 * (the kept one is the smallest).  Returns the number of candidates written (<= cap), 0 when the latest ksched_mask_alloc did not probe. */
int ksched_mask_probe_report(ksched_ctx *ctx, double *out_us, uint32_t cap);

/* The pick alone, from a feasibility mask already on the device (a previous ksched_eval_device* call): lets a caller
 * run the mask kernel of batch i + 1 and the pick of batch i on different HIP streams (the two do not depend on each
 * other; the mask of batch i must stay untouched until its pick has run).
 *   flags        : exactly one of KSCHED_PICK_SAMPLED (select_node_for_pod, src/main.rs:51-71: `samples`, `attempts`)
 *                  and KSCHED_PICK_BESTFIT (extension E1); KSCHED_FIT tells the best-fit pick that the mask includes
 *                  the resource fit (then `req_mem_bytes` [p] is read to skip candidates that cannot fit)
 *   feasible     : [p] rows, mask_pitch_words apart, as written by ksched_eval_device_pitched
 * Results are identical to requesting the pick in the ksched_eval_device* call that produced the mask. */
int ksched_pick_device(ksched_ctx *ctx, uint32_t p, const uint64_t *feasible, uint32_t mask_pitch_words,
                       const int64_t *req_mem_bytes, const uint32_t *samples, uint32_t attempts, uint32_t flags,
                       int32_t *out_binding, void *hip_stream);
/* The same from HOST masks (rows packed at ksched_mask_words(n) words): copies in, picks on the ctx's own stream, copies the bindings out and
 * waits.  For hosts that combine masks themselves -- the mirror ANDs the masks of a pod whose selector has more keys than one call takes
 * (KSCHED_MAX_KEYS; the reference has no limit, src/predicates.rs:48-53) and lets the device pick from the result. */
int ksched_pick(ksched_ctx *ctx, uint32_t p, const uint64_t *feasible, const int64_t *req_mem_bytes, const uint32_t *samples,
                uint32_t attempts, uint32_t flags, int32_t *out_binding);

/* ---- pipelined evaluation (throughput form) ------------------------------------------------
 * Consecutive batches do not depend on each other, and within a batch the pick only needs the finished mask.  A
 * ksched_pipe runs them software-pipelined on two internal HIP streams: the mask kernel of batch i + 1 on the
 * "mask" stream while the pick of batch i (and whatever the caller enqueues behind it, e.g. the RCCL all-gather of
 * the bindings) is still running on the "pick" stream.  `depth` slots (1 .. 64); the caller owns the per-slot output buffers
 * (device memory) and passes them to every submit, so the library retains nothing but streams and events.
 *
 *   ksched_pipe_submit(slot, ...)  enqueue one batch into `slot` (round-robin 0 .. depth-1):
 *        mask stream : mask kernel into `mask`
 *        pick stream : pick into `binding`
 *      By default the picks do not read the mask (KSCHED_OPT_PICK_FROM_MASK), so the two streams are not ordered
 *      against each other at all: each is in order by itself.  With a mask-reading pick the pick waits for its mask
 *      kernel and the slot's next mask kernel for that pick (events).
 *      `flags` must contain one KSCHED_PICK_* flag; predicates and pick mean what they mean in ksched_eval_device.
 *      All input arrays are device pointers that must be complete before the call and stay untouched until the
 *      slot's pick has run.  A caller that enqueues its own work on the pick stream behind the pick (reading
 *      `binding`) thereby also delays the slot's next use correctly (the next pick of the slot is ordered after it).
 *   ksched_pipe_wait(slot, stream)  make `hip_stream` wait for the slot's pick; stream == NULL blocks the host.
 *   ksched_pipe_wait_mask(slot, stream)  the same for the slot's MASK (the two streams are not ordered against each other, so
 *        a finished pick says nothing about the mask kernel).  Waits for everything submitted to the mask stream so far; the
 *        event is recorded by this call, so consumers that only read bindings pay nothing for it.
 *   ksched_pipe_stream(which)       the internal hipStream_t: 0 = mask stream, 1 = pick stream, 2 .. = the further streams of the
 *        alternate mode over more than two (NULL until a submit has used them).
 *   ksched_pipe_slot_stream(slot)   the stream that carried the slot's latest pick (alternate mode: its whole evaluation): work
 *        enqueued THERE after the submit -- the all-gather of the slot's bindings -- is ordered behind it by the stream itself,
 *        whatever mode the submit ran in (NULL before the slot's first submit).
 * Results are identical to ksched_eval_device_pitched with the same arguments (tests/test_gpu_parity.py).
 */
typedef struct ksched_pipe ksched_pipe;
int ksched_pipe_create(ksched_ctx *ctx, uint32_t depth, ksched_pipe **out);
void ksched_pipe_destroy(ksched_pipe *pipe);
int ksched_pipe_submit(ksched_pipe *pipe, uint32_t slot, uint32_t p, const int64_t *req_cpu_milli, const int64_t *req_mem_bytes,
                       const uint32_t *sel_val_ids, const uint64_t *tolerations, const uint32_t *samples, uint32_t attempts,
                       uint32_t flags, uint64_t *mask, uint32_t mask_pitch_words, int32_t *binding);
int ksched_pipe_wait(ksched_pipe *pipe, uint32_t slot, void *hip_stream);
int ksched_pipe_wait_mask(ksched_pipe *pipe, uint32_t slot, void *hip_stream);
void *ksched_pipe_stream(ksched_pipe *pipe, int which);
void *ksched_pipe_slot_stream(ksched_pipe *pipe, uint32_t slot);

/* ---- reasons ------------------------------------------------------------------------------
 * Host helper: rebuild check_node_validity's result for one pair from the two masks, in the
 * reference's order (fit first: src/predicates.rs:68-70, then selector: :72-74).
 * feasible_row / fit_row point at the pod's row of each mask.
 */
int ksched_reason(const uint64_t *feasible_row, const uint64_t *fit_row, uint32_t node, uint32_t flags);

/* Per-pair reasons, decided on the device: check_node_validity's result (src/predicates.rs:63-77) for `count` listed
 * (pod, node) pairs of an encoded batch -- what the reference logs at WARN for every rejected candidate
 * (src/main.rs:62).  Order: resources (:68-70), then selector (:72-74), then the taint extension.  Two masks cannot
 * tell a selector failure from a taint failure when both predicates are active (ksched_reason then answers
 * NODE_SELECTOR_MISMATCH); this entry point re-checks the listed pairs from the columns and can.
 *   pod columns as in ksched_eval (host pointers, [p]); pair_pod[i] < p, pair_node[i] < ksched_num_nodes
 *   flags : subset of KSCHED_FIT | KSCHED_SEL | KSCHED_TAINT;  out_reason[i] = KSCHED_REASON_* */
int ksched_explain(ksched_ctx *ctx, uint32_t p, const int64_t *req_cpu_milli, const int64_t *req_mem_bytes,
                   const uint32_t *sel_val_ids, const uint64_t *tolerations, uint32_t count, const uint32_t *pair_pod,
                   const uint32_t *pair_node, uint32_t flags, int32_t *out_reason);

/* ---- multi-GPU: all-gather of the (pod -> node) bindings over RCCL / xGMI --------------------
 * The pod batch row-shards over the GPUs of one node (north_star): rank r evaluates pod rows
 * [r * shard, (r + 1) * shard) with its own ksched_ctx against the replicated node snapshot, and ONE all-gather of the
 * int32 bindings (4 B per pod) gives every rank the whole table.  Masks stay on the GPU that produced them.  The
 * reference has no exchange step at all (one process, src/main.rs:127-152); this replaces nothing, it is what makes
 * the batched path span 8 GPUs.
 *
 * Two ways to build the communicator:
 *   one process per GPU : rank 0 calls ksched_comm_unique_id and hands the 128 bytes to the other ranks out of band
 *                         (env, file, TCP store); every rank calls ksched_comm_create(ctx, id, rank, nranks)
 *                         (ncclCommInitRank on the ctx's device; collective: all ranks must call it).
 *   one process, n GPUs : ksched_comm_create_local(ctxs, n, comms) (ncclCommInitAll): comms[i] belongs to ctxs[i].
 * ksched_allgather_bindings enqueues ncclAllGather(int32) on `hip_stream` and returns without syncing:
 *   gathered[r * count_per_rank + i] = rank r's local[i].  Device pointers; every rank passes the same count (pad the
 *   last shard with -1).  Enqueue it on the stream the pick was launched on and no host sync or event is needed.
 * With one process driving n devices the n calls of one collective must be issued together:
 * ksched_allgather_bindings_local wraps them in ncclGroupStart/End (local[i], gathered[i], hip_streams[i] belong to comms[i]).
 * If any of the n calls (or the group's end) fails, the collective is at best half issued: the library then aborts the whole clique
 * (ncclCommAbort), the handles stay valid only for ksched_comm_destroy, and every later call with them returns KSCHED_E_INVAL --
 * create a new clique.  The streams carry no stuck collective afterwards: ksched_eval_end on them returns.
 * RCCL is loaded on first use (dlopen "librccl.so.1"); a process that never calls these never loads it.
 * Test hook (never in production): with $KSCHED_TEST_HOOKS=1, $KSCHED_RCCL_LIB names the library to load instead
 * (tests/cpp/fake_rccl.cpp: n ranks on one GPU); $KSCHED_RCCL_LIB without the switch makes every ksched_comm_* call fail.
 */
#define KSCHED_COMM_ID_BYTES 128u
typedef struct ksched_comm ksched_comm;
int ksched_comm_unique_id(uint8_t *id /* [KSCHED_COMM_ID_BYTES] */);
int ksched_comm_create(ksched_ctx *ctx, const uint8_t *id, int rank, int nranks, ksched_comm **out);
int ksched_comm_create_local(ksched_ctx *const *ctxs, int n, ksched_comm **out /* [n] */);
void ksched_comm_destroy(ksched_comm *comm);
int ksched_comm_rank(const ksched_comm *comm);
int ksched_comm_size(const ksched_comm *comm);
int ksched_allgather_bindings(ksched_comm *comm, const int32_t *local, int32_t *gathered, uint32_t count_per_rank, void *hip_stream);
int ksched_allgather_bindings_local(ksched_comm *const *comms, int n, const int32_t *const *local, int32_t *const *gathered,
                                    uint32_t count_per_rank, void *const *hip_streams);
/* text of the calling thread's last RCCL failure (empty string if none) */
const char *ksched_comm_last_error(void);

/* ---- measurement --------------------------------------------------------------------------
 * With KSCHED_OPT_TIMING = 1 every ksched_eval* brackets its mask kernel with hipEvents on the
 * launch stream.  ksched_kernel_time_ms synchronises those events and returns the accumulated
 * kernel milliseconds and launch count since the last reset (both reset by the call).
 */
int ksched_kernel_time_ms(ksched_ctx *ctx, double *total_ms, uint64_t *launches);
/* The same measurement launch by launch: writes the duration (ms) of up to `cap` timed mask kernel launches since the
 * last reset, in launch order, returns how many were written (<0 on error) and resets like ksched_kernel_time_ms. */
int ksched_kernel_time_samples(ksched_ctx *ctx, double *out_ms, uint32_t cap);
/* Diagnostics: copy out the trace of the last fused launch (KSCHED_OPT_TRACE = 1): up to max_blocks records of
 * KSCHED_TRACE_WORDS uint64 (100 MHz timestamps); returns the number of blocks of that launch, <0 on error. */
int ksched_trace_read(ksched_ctx *ctx, uint64_t *out, uint32_t max_blocks);
/* Diagnostics: 64-bit checksums of the per-tile bitmap index as it is on the device right now (out[0]: bitmap rows, out[1]:
 * search trees + cnt tables; both 0 when the snapshot has no index).  Waits for pending snapshot work.  Lets tests show that
 * the device-built index, the host-specified one (KSCHED_OPT_INDEX_BUILD) and an incrementally updated one are the same bits. */
int ksched_index_checksum(ksched_ctx *ctx, uint64_t *out /* [2] */);
/* name of the mask kernel variant the last ksched_eval* used ("direct", "indexed", ...) */
const char *ksched_last_kernel(const ksched_ctx *ctx);
/* how the pick of the last ksched_eval* ran: "fused-tile" / "fused" (it rode in the fused mask launch as tile tests / as waves of
 * the fill, KSCHED_OPT_FUSED_PICK), "select" (its own launch testing the drawn candidates), "bestfit-rows", "from-mask"
 * (KSCHED_OPT_PICK_FROM_MASK / no bitmap index), "none" */
const char *ksched_last_pick(const ksched_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* KSCHED_H */
