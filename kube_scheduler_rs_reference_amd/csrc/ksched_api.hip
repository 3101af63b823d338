// ksched_api.hip -- implementation of include/ksched.h (C ABI) on HIP for gfx950.
//
// Host side of the evaluator: owns the device-resident node snapshot and its indexes, launches
// the mask kernels and the pick kernels.  There is deliberately NO CPU implementation of the
// predicates in this library: without a HIP device ksched_create fails with KSCHED_E_NODEVICE.
#include "../../include/ksched.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>


#include "comm_rccl.hpp"
#include "kernels_build.hpp"
#include "kernels_direct.hpp"
#include "kernels_fused.hpp"
#include "mask_alloc.hpp"
#include "tile_index.hpp"

using namespace ksched;

static_assert(kListBytes == 6144u && kTileNodes == 1024, "k_pick_bestfit_listed (kernels_direct.hpp) addresses the tile lists with these sizes");

namespace {

template <class T>
struct DevBuf {
    T *ptr = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipMalloc((void **)&ptr, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

}  // namespace

struct ksched_ctx {
    int device = 0;
    std::mutex mu;
    std::string last_error;
    hipStream_t stream = nullptr;  // used by the host-pointer entry points

    // node snapshot
    bool have_nodes = false;
    uint32_t n = 0, nkeys = 0, W = 0;
    bool have_taints = false;
    DevBuf<int64_t> ncpu, nmem;
    DevBuf<int64_t> nrec;  // 64-byte node records (kernels_direct.hpp): one cache line per candidate for the candidate-testing picks
    DevBuf<uint32_t> nlab;
    DevBuf<uint64_t> ntaint;
    DevBuf<uint32_t> bf_order, bf_rank;
    DevBuf<int64_t> bf_mem, bf_cpu;  // node columns once more, in best-fit order
    DevBuf<int64_t> cpu_sorted;      // ascending avail_cpu (rank of a cpu request)
    DevBuf<int64_t> bf_samples;      // sample arrays of bf_mem and cpu_sorted: [mem s1][mem s2][cpu s1][cpu s2]
    uint32_t bf_n1 = 0, bf_n2 = 0;
    DevBuf<int64_t> bf_levels;       // 8-ary level arrays of bf_mem / cpu_sorted (k_pick_bestfit_lanes)
    uint32_t bf_nlev = 0, bf_lvl_half = 0, bf_lvl_off[6] = {};
    DevBuf<uint32_t> bf_fallback;    // hand-over of the two-stage best-fit pick: 3 sets of sub-list counters in rotation, the listed mask, the sub-lists' 64-byte records
    uint32_t bf_slot = 0;            // which of the three counter sets the next two-stage pick uses (the call before zeroed it)
    size_t bf_fallback_zeroed_cap = 0;
    DevBuf<uint64_t> bf_trace;       // KSCHED_OPT_DEBUG bit 20: per-wave time stamps of the two best-fit stages (tools/bestfit_ab.py --trace)
    DevBuf<uint64_t> bf_rows;        // [rows][Wbf] bitmaps over best-fit positions (k_pick_bestfit_rows); built with the tile index
    bool bf_rows_built = false;
    uint32_t bf_row_cpu0 = 0, bf_q = 1, bf_W = 0;
    // the best-fit structures are built lazily, by the first PICK_BESTFIT request after the snapshot changed
    bool bf_dirty = true;
    DevBuf<uint32_t> by_cpu, cpurank;
    DevBuf<int64_t> srt_k0[2], srt_k1[2];  // ping-pong sets of the best-fit orders' merge sort (kernels_build.hpp)
    DevBuf<uint32_t> srt_idx[2];
    IndexedSnapshot idx;  // per-tile bitmap index (tile_index.hpp), built on the device (kernels_build.hpp)
    std::string index_reason;  // why the snapshot has no bitmap index (the fused kernel is then not applicable)
    // host -> device staging for the snapshot calls: pinned, so the copies are asynchronous on the ctx's stream
    uint8_t *h_stage = nullptr;
    size_t h_stage_cap = 0;
    hipEvent_t ev_stage = nullptr;  // the last copy out of h_stage
    DevBuf<uint8_t> d_stage;
    // Ordering between the snapshot (built / patched on `stream`) and evaluations on the caller's streams, without a device
    // synchronize: every stream an evaluation was enqueued on is remembered with an event; a snapshot change first makes
    // `stream` wait for what those streams hold (evaluations already enqueued read the snapshot as it was), and an evaluation
    // enqueued afterwards makes its stream wait for the change (ev_build).
    struct UserStream {
        hipStream_t s;
        hipEvent_t ev;
        uint64_t gen;  // snapshot generation this stream has waited for
    };
    std::vector<UserStream> user_streams;
    hipEvent_t ev_build = nullptr;
    uint64_t build_gen = 0;
    // Where a snapshot change is enqueued (snapshot_begin decides, the build / patch code uses it, snapshot_end records ev_build
    // there).  With exactly ONE caller stream known -- one scheduler loop on one stream, the common case -- the change goes onto
    // that stream itself: ordered behind the evaluations already there and ahead of the next ones by the stream, with no event
    // and no cross-stream wait at all (each costs ~5 us; on some hardware queues ~180 us, tools/stream_probe.py).  Otherwise:
    // the ctx's own stream and the events described above.  KSCHED_OPT_SNAPSHOT_STREAM = 1 keeps every change on the ctx's stream.
    hipStream_t change_stream = nullptr;
    uint64_t own_gen = 0;  // snapshot generation the ctx's own stream is ordered behind
    bool opt_own_stream = false;
    // The ctx-owned device scratch that evaluations on the caller's streams use (bf_fallback, scratch_mask, trace) belongs to one
    // stream at a time: when another stream is about to use it, that stream first waits for what the previous one holds
    // (scratch_enter; an event recorded at that moment, nothing on the common one-stream path).
    hipStream_t scratch_stream = nullptr;
    bool scratch_owned = false, scratch_ev_pending = false;
    hipEvent_t ev_scratch = nullptr;

    // scratch for the host-pointer path
    DevBuf<int64_t> pcpu, pmem;
    DevBuf<uint32_t> psel, psamples;
    DevBuf<uint64_t> ptol, feas, fit;
    DevBuf<int32_t> binding;
    DevBuf<int32_t> gathered;  // ksched_gather_buffer: the all-gathered binding table of a host that drives several devices
    DevBuf<uint32_t> xpairs;  // ksched_explain: [pair_pod][pair_node]
    DevBuf<int32_t> xreason;
    // scratch mask when a pick is requested without an output mask
    DevBuf<uint64_t> scratch_mask;

    // options
    int opt_kernel = KSCHED_KERNEL_AUTO;
    uint32_t opt_timing = 0;       // 0 off, N: every N-th mask launch carries events
    uint64_t timing_seq = 0;
    uint32_t opt_debug = 0;
    bool opt_trace = false;
    bool opt_pick_from_mask = false;
    int opt_fused_pick = 1;       // KSCHED_OPT_FUSED_PICK: 0 = the pick is its own launch, 1 = it rides in the fused mask launch (the form is chosen
                                  // by the request), 2 = ... only as waves of the fill, 3 = ... only as tile tests
    // per-pod accumulators of the tile-test pick (kernels_fused.hpp PICK == 2): all zero between launches (the kernel zeroes what it
    // used); one buffer per stream evaluations are enqueued on, so that launches on different streams may overlap
    struct PickAcc {
        hipStream_t s;
        DevBuf<uint64_t> buf;
    };
    std::vector<PickAcc> pick_acc;
    int opt_pipe_mode = 0;        // KSCHED_OPT_PIPE_MODE: 0 split (mask stream / pick stream), m >= 1 alternate (whole steps, stream = slot % max(2, m))
    int opt_round_order = 0;      // KSCHED_OPT_ROUND_ORDER: 0 interleaved wave-major (default), 1 blocked, 2 interleaved chunk-major
    uint32_t opt_mask_probe = 6;  // KSCHED_OPT_MASK_PROBE: candidates of ksched_mask_alloc's probe-and-keep path
    double mask_probe_us[16] = {};  // what the latest probe measured per candidate (diagnostics: ksched_last_error carries them as text)
    uint32_t opt_grid_cus = 0;    // KSCHED_OPT_GRID_CUS: compute units ONE fused mask launch may occupy (0 = the whole chip)
    uint32_t fault_kind = 0, fault_skip = 0;  // KSCHED_OPT_FAULT (test hook of the no-unwind rule)
    int opt_bestfit_stages = 0;  // KSCHED_OPT_BESTFIT_STAGES: 0 auto, 1 one stage, 2 two stages
    int opt_index_build = 0;  // KSCHED_OPT_INDEX_BUILD: 0 = device kernels (default), 1 = host spec (tile_index.hpp)
    DevBuf<uint64_t> trace;
    uint32_t trace_blocks_last = 0;
    const char *last_kernel = "none";
    const char *last_pick = "none";  // how the latest evaluation's pick ran (ksched_last_pick)

    // timing
    struct EvPair {
        hipEvent_t a, b;
    };
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
};

namespace {

// ---- nothing unwinds across the C ABI (include/ksched.h "Conventions") -----------------------------------------------------
// Every extern "C" body below is a function-try-block ending in KSCHED_ABI_CATCH: the library's own C++ (std::vector,
// std::string, std::mutex) may throw -- std::bad_alloc above all -- and an exception leaving an extern "C" function
// called from Rust or C is an abort.  bad_alloc -> KSCHED_E_NOMEM, anything else -> KSCHED_E_INVAL, text in ksched_last_error.
int abi_caught(ksched_ctx *c, int code, const char *what) noexcept {
    if (c) {
        try {  // (the body's lock_guard was released by the unwinding)
            std::lock_guard<std::mutex> lk(c->mu);
            c->last_error.assign("exception inside the library: ");
            c->last_error.append(what ? what : "?");
        } catch (...) {  // no memory for the text either: the code alone has to do
        }
    }
    return code;
}
#define KSCHED_ABI_CATCH(CTX)                                                                          \
    catch (const std::bad_alloc &) { return abi_caught((CTX), KSCHED_E_NOMEM, "std::bad_alloc"); }      \
    catch (const std::exception &e_) { return abi_caught((CTX), KSCHED_E_INVAL, e_.what()); }           \
    catch (...) { return abi_caught((CTX), KSCHED_E_INVAL, "unknown exception"); }

// KSCHED_OPT_FAULT: the test hook.  Called (with the ctx's mutex held) at the points where the library's C++ is entered.
void fault_point(ksched_ctx *c) {
    if (!c->fault_kind) return;
    if (c->fault_skip) {
        --c->fault_skip;
        return;
    }
    const uint32_t kind = c->fault_kind;
    c->fault_kind = 0;  // one shot
    if (kind == 1u) throw std::bad_alloc();
    throw std::runtime_error("injected fault (KSCHED_OPT_FAULT)");
}

int fail_hip(ksched_ctx *c, hipError_t e, const char *what) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    c->last_error = buf;
    return e == hipErrorOutOfMemory ? KSCHED_E_NOMEM : KSCHED_E_HIP;
}

#define HIPCHK(ctx, call)                                   \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #call); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int timing_slot(ksched_ctx *c, size_t *slot) {
    if (c->ev_used == c->ev_pool.size()) {
        // Timing-only events: without the system-scope fence a default event performs when it is recorded (a cache
        // write-back / invalidate around the very kernel being measured: +2 us at C3 and a cold L2 for its tables).
        // KSCHED_TIMING_EVENT_FLAGS overrides the flags (A/B of that effect, tools/).
        unsigned flags = hipEventDisableSystemFence;
        if (const char *e = getenv("KSCHED_TIMING_EVENT_FLAGS")) flags = (unsigned)strtoul(e, nullptr, 0);
        ksched_ctx::EvPair ep;
        HIPCHK(c, hipEventCreateWithFlags(&ep.a, flags));
        HIPCHK(c, hipEventCreateWithFlags(&ep.b, flags));
        c->ev_pool.push_back(ep);
    }
    *slot = c->ev_used++;
    return KSCHED_OK;
}

// ---- stream ordering (see ksched_ctx::user_streams) ---------------------------------------------------------------

// remember `s` as a stream evaluations are enqueued on; make it wait for the latest snapshot change if it has not yet
int stream_enter(ksched_ctx *c, hipStream_t s) {
    if (s == c->stream) {  // the ctx's own stream: in order by itself unless the latest change went onto a caller's stream
        if (c->own_gen != c->build_gen) {
            HIPCHK(c, hipStreamWaitEvent(s, c->ev_build, 0));
            c->own_gen = c->build_gen;
        }
        return KSCHED_OK;
    }
    ksched_ctx::UserStream *u = nullptr;
    for (auto &x : c->user_streams)
        if (x.s == s) u = &x;
    if (!u) {
        ksched_ctx::UserStream n{s, nullptr, 0};
        HIPCHK(c, hipEventCreateWithFlags(&n.ev, hipEventDisableTiming));
        c->user_streams.push_back(n);
        u = &c->user_streams.back();
    }
    if (u->gen != c->build_gen) {
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_build, 0));
        u->gen = c->build_gen;
    }
    return KSCHED_OK;
}

// `s` is about to enqueue work that uses the ctx-owned scratch buffers
int scratch_enter(ksched_ctx *c, hipStream_t s) {
    if (c->scratch_owned && c->scratch_stream != s) {
        HIPCHK(c, hipEventRecord(c->ev_scratch, c->scratch_stream));
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_scratch, 0));
    } else if (!c->scratch_owned && c->scratch_ev_pending) {
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_scratch, 0));
    }
    c->scratch_stream = s;
    c->scratch_owned = true;
    c->scratch_ev_pending = false;
    return KSCHED_OK;
}

void stream_forget(ksched_ctx *c, hipStream_t s) {
    for (size_t i = 0; i < c->pick_acc.size(); ++i)
        if (c->pick_acc[i].s == s) {  // (its launches are ordered before whatever the caller does to the stream next: freeing is safe after a sync there;
                                      // hipFree synchronises the device itself)
            c->pick_acc[i].buf.release();
            c->pick_acc.erase(c->pick_acc.begin() + (std::ptrdiff_t)i);
            break;
        }
    if (c->scratch_owned && c->scratch_stream == s) {  // its last use of the scratch buffers stays ordered: as an event
        c->scratch_ev_pending = hipEventRecord(c->ev_scratch, s) == hipSuccess;
        if (!c->scratch_ev_pending) (void)hipGetLastError();
        c->scratch_owned = false;
        c->scratch_stream = nullptr;
    }
    for (size_t i = 0; i < c->user_streams.size(); ++i)
        if (c->user_streams[i].s == s) {
            (void)hipEventDestroy(c->user_streams[i].ev);
            c->user_streams.erase(c->user_streams.begin() + (std::ptrdiff_t)i);
            return;
        }
}

// Before the snapshot changes: choose the stream that carries the change (ksched_ctx::change_stream).  One caller stream known: that
// stream (it is behind its own evaluations by itself).  Otherwise the ctx's stream, which first waits for everything already enqueued
// on the remembered streams.
int snapshot_begin(ksched_ctx *c) {
    c->change_stream = c->stream;
    if (c->user_streams.size() == 1 && !c->opt_own_stream) {
        auto &u = c->user_streams[0];
        const hipError_t q = hipStreamQuery(u.s);  // host-side probe: a stream the caller destroyed without telling is not used
        if (q == hipSuccess || q == hipErrorNotReady) {
            if (u.gen != c->build_gen) {  // (an earlier change went onto the ctx's stream and this stream has not met it yet)
                HIPCHK(c, hipStreamWaitEvent(u.s, c->ev_build, 0));
                u.gen = c->build_gen;
            }
            c->change_stream = u.s;
            return KSCHED_OK;
        }
        (void)hipGetLastError();  // fall through: the loop below forgets the stream
    }
    if (c->own_gen != c->build_gen) {  // an earlier change went onto a caller's stream: this one is ordered behind it
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_build, 0));
        c->own_gen = c->build_gen;
    }
    for (size_t i = 0; i < c->user_streams.size();) {
        auto &u = c->user_streams[i];
        if (hipEventRecord(u.ev, u.s) != hipSuccess) {  // the caller destroyed that stream: forget it
            (void)hipGetLastError();
            (void)hipEventDestroy(u.ev);
            c->user_streams.erase(c->user_streams.begin() + (std::ptrdiff_t)i);
            continue;
        }
        HIPCHK(c, hipStreamWaitEvent(c->stream, u.ev, 0));
        ++i;
    }
    return KSCHED_OK;
}

// after the change has been enqueued on change_stream: ev_build marks it for every other stream
int snapshot_end(ksched_ctx *c) {
    const hipStream_t s = c->change_stream;
    HIPCHK(c, hipEventRecord(c->ev_build, s));
    ++c->build_gen;
    if (s == c->stream) c->own_gen = c->build_gen;
    for (auto &u : c->user_streams)
        if (u.s == s) u.gen = c->build_gen;  // the stream that carries the change is behind it by itself
    return KSCHED_OK;
}

// pinned staging: returns a host pointer to `bytes` bytes whose previous use (an async copy) has completed
int stage_reserve(ksched_ctx *c, size_t bytes, uint8_t **out) {
    if (c->h_stage) HIPCHK(c, hipEventSynchronize(c->ev_stage));
    if (bytes > c->h_stage_cap) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_cap = 0;
        const size_t cap = std::max<size_t>(bytes + bytes / 4, 1u << 16);
        HIPCHK(c, hipHostMalloc((void **)&c->h_stage, cap, hipHostMallocDefault));
        c->h_stage_cap = cap;
    }
    *out = c->h_stage;
    return KSCHED_OK;
}

// ---- index and best-fit build (kernels_build.hpp) --------------------------------------------------------------------

// (re)build the fit part of the listed tiles (or of every tile: tiles == nullptr) on the ctx's stream
int launch_build_fit(ksched_ctx *c, const uint32_t *d_tile_list, uint32_t count) {
    const IndexedLayout &l = c->idx.lay;
    BuildFitArgs a{};
    a.ncpu = c->ncpu.ptr;
    a.nmem = c->nmem.ptr;
    a.tables = c->idx.d_tables;
    a.aux = c->idx.d_aux;
    a.tile_list = d_tile_list;
    a.n = l.n;
    a.rows = l.rows;
    a.row_cpu = l.row_cpu;
    hipLaunchKernelGGL(k_build_tile_fit, dim3(d_tile_list ? count : l.tiles, 2), dim3(1024), 0, c->change_stream, a);
    HIPCHK(c, hipGetLastError());
    return KSCHED_OK;
}

int launch_build_lists(ksched_ctx *c) {
    const IndexedLayout &l = c->idx.lay;
    if (l.nlist == 0) return KSCHED_OK;
    BuildListArgs a{};
    a.nlab = c->nlab.ptr;
    a.lists = c->idx.d_list;
    a.n = l.n;
    a.nlist = l.nlist;
    for (uint32_t j = 0; j < l.nlist; ++j) a.list_col[j] = l.list_col[j];
    hipLaunchKernelGGL(k_build_tile_list, dim3(l.tiles, l.nlist), dim3(1024), 0, c->change_stream, a);
    HIPCHK(c, hipGetLastError());
    return KSCHED_OK;
}

int launch_build_named(ksched_ctx *c) {
    const IndexedLayout &l = c->idx.lay;
    BuildNamedArgs a{};
    a.nlab = c->nlab.ptr;
    a.ntaint = c->have_taints ? c->ntaint.ptr : nullptr;
    a.tables = c->idx.d_tables;
    a.lab_meta = c->idx.d_lab_meta;
    a.n = l.n;
    a.rows = l.rows;
    a.nkeys = l.nkeys;
    a.ngroups = l.ngroups;
    a.row_valid = l.row_valid;
    a.row_taint = l.row_taint;
    a.named_rows = l.row_cpu;
    const uint32_t lds = l.row_cpu * 128u;
    HIPCHK(c, hipFuncSetAttribute((const void *)k_build_tile_named, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_build_tile_named, dim3(l.tiles), dim3(1024), lds, c->change_stream, a);
    HIPCHK(c, hipGetLastError());
    return KSCHED_OK;
}

// Best-fit candidate order of the snapshot, ascending (avail_mem, avail_cpu, node), its inverse, the node
// columns in that order, the sorted cpu column with the sample arrays of the two rank searches, and -- when the snapshot has a
// bitmap index -- the named rows once more over best-fit positions plus the 257 cpu threshold rows (k_pick_bestfit_rows).
// Two device merge sorts (kernels_build.hpp: k_sort_runs + k_merge_pass) and two kernels, on the ctx's stream; called lazily by the first PICK_BESTFIT request
// after the snapshot changed (ksched_set_nodes / ksched_update_nodes only mark it dirty).
int build_bestfit(ksched_ctx *c) {
    const uint32_t n = c->n;
    c->bf_rows_built = false;
    if (n == 0) {
        c->bf_dirty = false;
        return KSCHED_OK;
    }
    hipStream_t s = c->change_stream;
    const uint32_t n1 = (n + 63u) / 64u, n2 = (n + 4095u) / 4096u;
    HIPCHK(c, c->by_cpu.reserve(n));
    HIPCHK(c, c->cpurank.reserve(n));
    HIPCHK(c, c->bf_samples.reserve(2 * (size_t)(n1 + n2)));
    const dim3 grid((n + 255u) / 256u), block(256);
    // The two orders, by the merge sort of kernels_build.hpp: runs of 1024 sorted in LDS, then merged by ranking, ping-pong between two
    // sets of (k0, k1, idx) arrays; the last pass of each sort lands in the arrays the pick kernels read.
    for (int b = 0; b < 2; ++b) {
        HIPCHK(c, c->srt_k0[b].reserve(n));
        HIPCHK(c, c->srt_k1[b].reserve(n));
        HIPCHK(c, c->srt_idx[b].reserve(n));
    }
    auto device_sort = [&](const int64_t *k0, const int64_t *k1, int64_t *dst_k0, uint32_t *dst_idx) -> int {
        if (n > (1u << 31)) {  // (the merge passes index records with 32 bits; no snapshot of that size fits a GPU's memory anyway)
            c->last_error = "best-fit structures: more than 2^31 nodes";
            return KSCHED_E_INVAL;
        }
        uint32_t passes = 0;
        for (uint64_t run = 1024; run < n; run <<= 1) ++passes;
        // buffer of pass i's output: the destination arrays for the last one, the ping-pong sets before it
        auto out_of = [&](uint32_t i, SortArgs &q) {  // i = 0: k_sort_runs, i = 1 .. passes: merge passes
            if (i == passes) {
                q.k0_out = dst_k0;
                q.k1_out = k1 ? c->srt_k1[i & 1].ptr : nullptr;  // (nobody reads the second key of the final order)
                q.idx_out = dst_idx;
            } else {
                q.k0_out = c->srt_k0[i & 1].ptr;
                q.k1_out = k1 ? c->srt_k1[i & 1].ptr : nullptr;
                q.idx_out = c->srt_idx[i & 1].ptr;
            }
        };
        SortArgs q{};
        q.n = n;
        q.k0_in = k0;
        q.k1_in = k1;
        q.idx_in = nullptr;
        q.run = 0;
        out_of(0, q);
        hipLaunchKernelGGL(k_sort_runs, dim3((n + 1023u) / 1024u), dim3(1024), 0, s, q);
        HIPCHK(c, hipGetLastError());
        uint32_t i = 0;
        for (uint64_t run = 1024; run < n; run <<= 1) {
            SortArgs m{};
            m.n = n;
            m.run = (uint32_t)run;
            m.k0_in = q.k0_out;
            m.k1_in = q.k1_out;
            m.idx_in = q.idx_out;
            out_of(++i, m);
            hipLaunchKernelGGL(k_merge_pass, grid, block, 0, s, m);
            HIPCHK(c, hipGetLastError());
            q = m;
        }
        return KSCHED_OK;
    };
    // order 1: ascending (cpu, node) -> cpu_sorted (the sorted cpu column), by_cpu (node with cpu rank r)
    if (int rc = device_sort(c->ncpu.ptr, nullptr, c->cpu_sorted.ptr, c->by_cpu.ptr)) return rc;
    // order 2: ascending (mem, cpu, node) -> bf_mem, bf_order
    if (int rc = device_sort(c->nmem.ptr, c->ncpu.ptr, c->bf_mem.ptr, c->bf_order.ptr)) return rc;
    BfGatherArgs g{};
    g.ncpu = c->ncpu.ptr;
    g.nmem = c->nmem.ptr;
    g.bf_order = c->bf_order.ptr;
    g.by_cpu = c->by_cpu.ptr;
    g.bf_rank = c->bf_rank.ptr;
    g.cpurank = c->cpurank.ptr;
    g.bf_mem = c->bf_mem.ptr;
    g.bf_cpu = c->bf_cpu.ptr;
    g.cpu_sorted = c->cpu_sorted.ptr;
    g.samples = c->bf_samples.ptr;
    g.n = n;
    g.n1 = n1;
    g.n2 = n2;
    hipLaunchKernelGGL(k_bf_gather, grid, block, 0, s, g);
    HIPCHK(c, hipGetLastError());
    c->bf_n1 = n1;
    c->bf_n2 = n2;
    {   // 8-ary level arrays for the lane-per-pod searches: level k = last element of every block of 8^k entries
        BfLevelsArgs lv{};
        uint32_t nk = n, off = 0;
        c->bf_nlev = 0;
        while (nk > 8u && c->bf_nlev < 6u) {
            nk = (nk + 7u) / 8u;
            c->bf_lvl_off[c->bf_nlev++] = off;
            off += (nk + 7u) & ~7u;  // every level starts on a 64-byte boundary and may be read in whole blocks of eight
        }
        c->bf_lvl_half = off;
        HIPCHK(c, c->bf_levels.reserve(2 * (size_t)off + 8));
        if (c->bf_nlev) {
            lv.bf_mem = c->bf_mem.ptr;
            lv.cpu_sorted = c->cpu_sorted.ptr;
            lv.lvl = c->bf_levels.ptr;
            lv.n = n;
            lv.nlev = c->bf_nlev;
            lv.lvl_half = off;
            for (uint32_t k = 0; k < 6; ++k) lv.lvl_off[k] = c->bf_lvl_off[k];
            hipLaunchKernelGGL(k_bf_levels, dim3(((n + 7u) / 8u + 255u) / 256u), dim3(256), 0, s, lv);
            HIPCHK(c, hipGetLastError());
        }
    }
    if (c->idx.built) {  // (list keys have no rows: pods that constrain one are picked from the key's sorted lists, k_pick_bestfit_listed)
        const IndexedLayout &l = c->idx.lay;
        const uint32_t Wbf = ((n + 63u) / 64u + 7u) & ~7u, named = l.row_cpu;  // (rows padded to whole 64-byte lines: k_pick_bestfit_lanes reads aligned blocks of 8 words)
        const uint32_t levels = 256u, q = (n + levels - 1u) / levels;
        const uint32_t rows = named + levels + 1u;
        HIPCHK(c, c->bf_rows.reserve((size_t)rows * Wbf));
        HIPCHK(c, hipMemsetAsync(c->bf_rows.ptr, 0, (size_t)rows * Wbf * 8, s));
        BfRowsArgs r{};
        r.nlab = c->nlab.ptr;
        r.ntaint = c->have_taints ? c->ntaint.ptr : nullptr;
        r.bf_order = c->bf_order.ptr;
        r.cpurank = c->cpurank.ptr;
        r.lab_meta = c->idx.d_lab_meta;
        r.rows = c->bf_rows.ptr;
        r.n = n;
        r.Wbf = Wbf;
        r.nkeys = c->nkeys;
        r.ngroups = l.ngroups;
        r.row_valid = l.row_valid;
        r.row_taint = l.row_taint;
        r.row_cpu0 = named;
        r.q = q;
        r.levels = levels;
        hipLaunchKernelGGL(k_bf_rows, dim3((Wbf + 3u) / 4u), dim3(256), 0, s, r);
        HIPCHK(c, hipGetLastError());
        c->bf_row_cpu0 = named;
        c->bf_W = Wbf;
        c->bf_q = q;
        c->bf_rows_built = true;
    }
    c->bf_dirty = false;
    return KSCHED_OK;
}

// will the best-fit rows exist once ensure_bestfit has run?  (they are built with the bitmap index's row numbering)
inline bool bf_rows_expected(const ksched_ctx *c) { return c->idx.built && c->n > 0; }

// a PICK_BESTFIT request is about to be enqueued: make sure the structures match the snapshot
int ensure_bestfit(ksched_ctx *c) {
    if (!c->bf_dirty) return KSCHED_OK;
    int rc = snapshot_begin(c);  // picks already enqueued read the previous order
    if (rc) return rc;
    if ((rc = build_bestfit(c))) return rc;
    return snapshot_end(c);
}

// ---- mask kernel dispatch ---------------------------------------------------------------------

struct DirectPtrs {
    const int64_t *ncpu, *nmem;
    const uint32_t *nlab;
    const uint64_t *ntaint;
    const int64_t *pcpu, *pmem;
    const uint32_t *psel;
    const uint64_t *ptol;
    uint64_t *out_feas, *out_fit;
};

template <bool SEL, bool TAINT>
void launch_direct_t(const DirectPtrs &q, const DirectArgs &a, dim3 grid, bool want_fit, hipStream_t s) {
    if (want_fit)
        hipLaunchKernelGGL((k_eval_direct<SEL, TAINT, true>), grid, dim3(64 * kDirectWaves), 0, s, q.ncpu, q.nmem, q.nlab,
                           q.ntaint, q.pcpu, q.pmem, q.psel, q.ptol, q.out_feas, q.out_fit, a);
    else
        hipLaunchKernelGGL((k_eval_direct<SEL, TAINT, false>), grid, dim3(64 * kDirectWaves), 0, s, q.ncpu, q.nmem, q.nlab,
                           q.ntaint, q.pcpu, q.pmem, q.psel, q.ptol, q.out_feas, q.out_fit, a);
}

int run_direct(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
               const uint64_t *ptol, uint32_t flags, uint64_t *out_feas, uint64_t *out_fit, uint32_t pitch, hipStream_t s) {
    DirectPtrs q{};
    q.ncpu = c->ncpu.ptr;
    q.nmem = c->nmem.ptr;
    q.nlab = c->nlab.ptr;
    q.ntaint = c->have_taints ? c->ntaint.ptr : nullptr;
    q.pcpu = pcpu;
    q.pmem = pmem;
    q.psel = psel;
    q.ptol = ptol;
    q.out_feas = out_feas;
    q.out_fit = out_fit;
    DirectArgs a{};
    a.n = c->n;
    a.p = p;
    a.W = c->W;
    a.pitch = pitch;
    a.do_fit = (flags & KSCHED_FIT) ? 1u : 0u;

    const bool sel = (flags & KSCHED_SEL) && psel && c->nkeys > 0;
    const bool taint = (flags & KSCHED_TAINT) != 0;
    const bool want_fit = (flags & KSCHED_WANT_FIT_MASK) && out_fit;

    const uint32_t words_per_block = kDirectCW * kDirectWaves;
    const uint32_t gx = (c->W + words_per_block - 1) / words_per_block;
    const uint32_t pod_tiles = (p + 63) / 64;
    // aim for >= ~2048 blocks so all 256 CUs (8 XCDs) stay busy, but keep a block on its node
    // columns for as many pods as possible (node registers are loaded once per block)
    uint32_t gy = std::max(1u, std::min(pod_tiles, (2048u + gx - 1) / gx));
    a.pod_tiles_per_block = (pod_tiles + gy - 1) / gy;
    gy = (pod_tiles + a.pod_tiles_per_block - 1) / a.pod_tiles_per_block;
    dim3 grid(gx, gy);

    // first pass: fit + taints + keys [0, 8)
    a.key0 = 0;
    a.nkeys = sel ? std::min(c->nkeys, (uint32_t)kDirectKeys) : 0;
    a.accumulate = 0;
    if (sel) {
        if (taint) launch_direct_t<true, true>(q, a, grid, want_fit, s);
        else launch_direct_t<true, false>(q, a, grid, want_fit, s);
    } else {
        if (taint) launch_direct_t<false, true>(q, a, grid, want_fit, s);
        else launch_direct_t<false, false>(q, a, grid, want_fit, s);
    }
    // further passes: 8 more keys each, ANDed into the feasible mask
    if (sel && out_feas) {
        for (uint32_t k0 = kDirectKeys; k0 < c->nkeys; k0 += kDirectKeys) {
            DirectArgs b = a;
            b.key0 = k0;
            b.nkeys = std::min(c->nkeys - k0, (uint32_t)kDirectKeys);
            b.accumulate = 1;
            b.do_fit = 0;
            DirectPtrs r = q;
            r.out_fit = nullptr;
            r.ptol = nullptr;
            launch_direct_t<true, false>(r, b, grid, false, s);
        }
    }
    HIPCHK(c, hipGetLastError());
    c->last_kernel = "direct";
    return KSCHED_OK;
}

// the pick of select_node_for_pod (src/main.rs:51-71) / the best-fit extension, from a feasibility mask on the device
int launch_pick(ksched_ctx *c, uint32_t p, const uint64_t *feas, uint32_t pitch, const int64_t *pmem, const uint32_t *samples,
                uint32_t attempts, uint32_t flags, int32_t *out_binding, hipStream_t s) {
    if ((flags & KSCHED_PICK_BESTFIT) && c->n > 0)
        if (int rcb = ensure_bestfit(c)) return rcb;
    if (int rce = stream_enter(c, s)) return rce;
    if (flags & KSCHED_PICK_SAMPLED) {
        hipLaunchKernelGGL(k_pick_sampled, dim3((p + 255) / 256), dim3(256), 0, s, feas, samples, out_binding, p, c->n,
                           pitch, attempts);
    } else if (flags & KSCHED_PICK_BESTFIT) {
        hipLaunchKernelGGL(k_pick_bestfit, dim3((p + 3) / 4), dim3(256), 0, s, feas, c->bf_order.ptr, c->bf_rank.ptr,
                           c->bf_mem.ptr, pmem, out_binding, p, c->n, c->W, pitch, (flags & KSCHED_FIT) ? 1u : 0u);
    }
    HIPCHK(c, hipGetLastError());
    return KSCHED_OK;
}

// KSCHED_OPT_DEBUG bit 20: where the waves of the two best-fit stages spend their time (synchronous; stderr; tools only)
void bestfit_trace_report(ksched_ctx *c, uint32_t p, size_t slots2, hipStream_t s) {
    const size_t waves1 = (p + 63) / 64, total = waves1 * 8 + slots2 * 4;
    std::vector<uint64_t> h(total);
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(h.data(), c->bf_trace.ptr, total * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    if (const char *path = getenv("KSCHED_BF_TRACE_FILE"))  // the raw stamps: [waves of stage 1][8] then [hand-over slots][4] uint64
        if (FILE *f = fopen(path, "wb")) {
            fwrite(h.data(), 8, total, f);
            fclose(f);
        }
    auto pct = [](std::vector<double> &v, double f) { return v.empty() ? 0.0 : v[(size_t)(f * (double)(v.size() - 1))]; };
    const double ghz = 0.1;  // s_memrealtime: the constant 100 MHz reference clock
    uint64_t t0 = ~0ull;
    for (size_t w = 0; w < waves1; ++w)
        if (h[w * 8]) t0 = std::min(t0, h[w * 8]);
    const char *names[4] = {"operands", "searches", "word trips", "hand-over"};
    std::vector<double> ph[4], entry, exit_, und;
    for (size_t w = 0; w < waves1; ++w) {
        const uint64_t *r = &h[w * 8];
        if (!r[0] || !r[4]) continue;
        for (int i = 0; i < 4; ++i) ph[i].push_back((double)(r[i + 1] - r[i]) / ghz * 1e-3);
        entry.push_back((double)(r[0] - t0) / ghz * 1e-3);
        exit_.push_back((double)(r[4] - t0) / ghz * 1e-3);
        und.push_back((double)r[5]);
    }
    auto line = [&](const char *what, std::vector<double> &v) {
        std::sort(v.begin(), v.end());
        double sum = 0;
        for (double x : v) sum += x;
        fprintf(stderr, "  %-28s mean %7.2f   p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f   (%zu waves)\n", what, v.empty() ? 0.0 : sum / (double)v.size(), pct(v, 0.1), pct(v, 0.5), pct(v, 0.9),
                pct(v, 1.0), v.size());
    };
    fprintf(stderr, "best-fit stage 1 (k_pick_bestfit_lanes), us:\n");
    line("entry after the first wave", entry);
    for (int i = 0; i < 4; ++i) line(names[i], ph[i]);
    line("exit after the first entry", exit_);
    line("lanes handed over per wave", und);
    const uint64_t *t2 = &h[waves1 * 8];
    uint64_t t20 = ~0ull;
    std::vector<double> e2, d2, x2, words;
    for (size_t w = 0; w < slots2; ++w)
        if (t2[w * 4]) t20 = std::min(t20, t2[w * 4]);
    for (size_t w = 0; w < slots2; ++w) {
        const uint64_t *r = &t2[w * 4];
        if (!r[0] || !r[1]) continue;
        e2.push_back((double)(r[0] - t20) / ghz * 1e-3);
        d2.push_back((double)(r[1] - r[0]) / ghz * 1e-3);
        x2.push_back((double)(r[1] - t20) / ghz * 1e-3);
    }
    fprintf(stderr, "best-fit stage 2 (k_pick_bestfit_handed), us:   first entry %.2f us after stage 1's first entry\n", t20 == ~0ull ? 0.0 : (double)(t20 - t0) / ghz * 1e-3);
    line("entry after the first wave", e2);
    line("scan of one pod", d2);
    line("exit after the first entry", x2);
}

SelectArgs make_select_args(const ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                            const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags, int32_t *out_binding) {
    const bool sel = (flags & KSCHED_SEL) && psel && c->nkeys > 0;
    const bool taint = (flags & KSCHED_TAINT) && c->have_taints;
    SelectArgs q{};
    q.nrec = c->nrec.ptr;
    q.nlab = c->nlab.ptr;
    q.pcpu = pcpu;
    q.pmem = pmem;
    q.psel = sel ? psel : nullptr;
    q.ptol = ptol;
    q.samples = samples;
    q.binding = out_binding;
    q.p = p;
    q.n = c->n;
    q.nkeys = sel ? c->nkeys : 0u;
    q.attempts = attempts;
    q.do_fit = (flags & KSCHED_FIT) ? 1u : 0u;
    q.do_taint = taint ? 1u : 0u;
    return q;
}

// select_node_for_pod the reference's way: only the sampled candidates are tested, from the columns (k_select_sampled)
int launch_select(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel, const uint64_t *ptol,
                  const uint32_t *samples, uint32_t attempts, uint32_t flags, int32_t *out_binding, hipStream_t s) {
    const SelectArgs q = make_select_args(c, p, pcpu, pmem, psel, ptol, samples, attempts, flags, out_binding);
    const dim3 grid((p + 255) / 256), block(256);  // (64- and 128-thread blocks measured slower: 6.6 / 6.5 us against 5.7 at C3)
    if (attempts <= 5) {
        switch ((c->opt_debug >> 8) & 3u) {  // KSCHED_OPT_DEBUG bits 8-9: A/B of the number of eagerly fetched draws (tools/)
            case 1: hipLaunchKernelGGL((k_select_sampled<5, 3>), grid, block, 0, s, q); break;
            case 2: hipLaunchKernelGGL((k_select_sampled<5, 5>), grid, block, 0, s, q); break;
            case 3: hipLaunchKernelGGL((k_select_sampled<5, 2>), grid, block, 0, s, q); break;
            default: hipLaunchKernelGGL((k_select_sampled<5, 1>), grid, block, 0, s, q);  // rocprofv3 at C3, in the step: 1 eager draw 5.79 us, 2: 6.08, 3: 6.78, 5: 7.25
        }
    }
    else hipLaunchKernelGGL((k_select_sampled<8, 2>), grid, block, 0, s, q);
    HIPCHK(c, hipGetLastError());
    return KSCHED_OK;
}

int eval_on_device(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                   const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags,
                   uint64_t *out_feas, uint64_t *out_fit, int32_t *out_binding, uint32_t pitch, hipStream_t s) {
    const bool pick_s = flags & KSCHED_PICK_SAMPLED, pick_b = flags & KSCHED_PICK_BESTFIT;
    if (p == 0) return KSCHED_OK;
    if (pick_b && c->n > 0)
        if (int rcb = ensure_bestfit(c)) return rcb;  // lazily (re)built after a snapshot change, on the ctx's stream
    if (int rce = stream_enter(c, s)) return rce;      // `s` waits for the latest snapshot change; remembered for the next one
    if (c->n == 0) {
        // no nodes: empty mask rows, no binding possible (reference: choose() on an empty store
        // yields None on every attempt, src/main.rs:56,70)
        if ((pick_s || pick_b) && out_binding) HIPCHK(c, hipMemsetAsync(out_binding, 0xFF, (size_t)p * sizeof(int32_t), s));
        return KSCHED_OK;
    }
    fault_point(c);
    // kernel choice: fused (one launch over the bitmap index) when the snapshot has an index that fits LDS, else the
    // always-applicable direct kernel; KSCHED_OPT_KERNEL can force one.
    const bool want_mask = out_feas || out_fit;
    const bool can_fused = fused_applicable(c->idx, flags);
    int kern = c->opt_kernel;
    if (kern == KSCHED_KERNEL_AUTO) kern = can_fused ? KSCHED_KERNEL_FUSED : KSCHED_KERNEL_DIRECT;
    // The sampled pick tests only the drawn candidates, from the node records: it does not need the mask.  When a mask is
    // asked for too and the fused kernel runs, the pick RIDES in that launch (KSCHED_OPT_FUSED_PICK, kernels_fused.hpp "PICK":
    // a step is one kernel); otherwise it is its own launch, first (nothing waits on a mask kernel), and a bindings-only
    // request launches no mask kernel at all.  KSCHED_OPT_PICK_FROM_MASK restores the mask-reading pick (a cross-check).
    const bool select_direct = pick_s && !c->opt_pick_from_mask;
    // Does a riding pick pay?  Two forms (kernels_fused.hpp PICK): TILE TESTS -- every tile-block of a pod range tests the draws that fall into
    // its tile from the rows it holds -- and WAVES OF THE FILL, which run select_one_pod while the tile is staged.
    //  * Waves of the fill only hide in the fill: they ride when a wave has at most five rounds (C3: 2, the C4 shard: 5; beyond that they cost
    //    twice the stand-alone kernel, round 3's measurement, re-measured in round 6: 400 k x 5 k 67.1 us riding against 66.1).
    //  * Tile tests re-read a pod's operands and draws once per tile; the tile-blocks of a pod range sit on one XCD, so tiles - 1 of those reads
    //    come from that XCD's L2 -- as long as the XCD's share of the batch's operands and draws (68 B per pod / 8 XCDs) stays in its 4 MiB.
    //    Round 6 (session r7a, interleaved round order, step us riding / own launch): 100 k x 5 k  17.6 / 22.6;  400 k x 5 k  53.4 / 65.9;
    //    250 k x 10 k  64.8 / 75.0;  500 k x 10 k  123.3 / 142.8;  but 800 k x 5 k  143.4 / 125.3, 1 M x 10 k  322.0 / 288.6, 1.6 M x 5 k  281.8 / 241.6.
    //    (Rounds 3 - 5 had the tile tests stop riding at five rounds per wave too: with every wave on its own contiguous pod range the blocks of
    //    a pod range drifted apart much earlier.)  They ride up to 524 288 pods per call.
    const bool tile_form = c->opt_fused_pick == 3 || (c->opt_fused_pick == 1 && can_fused && c->idx.lay.tiles >= 2u && c->idx.lay.tiles <= 12u &&
                                                      fused_tile_pick_applicable(c->idx, flags, attempts, psel != nullptr));
    bool ride_pays = true;
    if (c->opt_fused_pick == 1 && can_fused) {
        if (tile_form) {
            ride_pays = p <= (1u << 19);
        } else {
            const uint32_t tiles = std::max(1u, c->idx.lay.tiles), cus = c->opt_grid_cus ? c->opt_grid_cus : 256u;
            const uint32_t chunks = std::max(1u, std::min(cus / tiles, (p + 255u) / 256u));
            ride_pays = (uint64_t)p <= (uint64_t)chunks * 5u * 64u * kFusedWaves;
        }
    }
    const bool pick_rides = select_direct && want_mask && c->opt_fused_pick && kern == KSCHED_KERNEL_FUSED && can_fused && ride_pays &&
                            fused_pick_applicable(c->idx, flags, (flags & KSCHED_WANT_FIT_MASK) && out_fit, p);
    SelectArgs ride{};
    int ride_form = 1;
    uint64_t *ride_acc = nullptr;
    if (pick_rides) {
        ride = make_select_args(c, p, pcpu, pmem, psel, ptol, samples, attempts, flags, out_binding);
        // the form: tile tests in phase 1 (no node records fetched, no wave taken off the staging) where the request allows it, else
        // waves of the fill running select_one_pod
        const bool tile_ok = fused_tile_pick_applicable(c->idx, flags, attempts, psel != nullptr);
        // Which form when both apply: the tile tests cost every (pod, tile) pair five draw loads and a handful of LDS reads, the waves
        // of the fill cost every block a longer fill.  Measured (session r3g3, rotated outputs, step): 5 tiles (C3) 19.8 us against 21.6;
        // 10 tiles (the C4 shard) 45.3 us against 42.1; 1 tile (C2) 7.1 us against 6.8.  With one device-scope atomic per unit of eight pods and
        // the draws loaded coalesced (later in round 3) the C4 shard reads 40.9 - 41.5 us against 42.1 - 42.3, C2 6.8 against 6.4: 2 .. 12 tiles.
        const bool tile_pays = c->idx.lay.tiles >= 2u && c->idx.lay.tiles <= 12u;
        if (((c->opt_fused_pick == 1 && tile_pays) || c->opt_fused_pick == 3) && tile_ok) {
            ksched_ctx::PickAcc *pa = nullptr;
            for (auto &x : c->pick_acc)
                if (x.s == s) pa = &x;
            if (!pa) {
                c->pick_acc.emplace_back();
                pa = &c->pick_acc.back();
                pa->s = s;
            }
            const size_t units = ((size_t)p + 7) / 8 + 8;  // one word per unit of eight pods (+ slack: the lanes past a batch's last unit compute an address, never use it)
            if (pa->buf.cap < units) {  // (re)allocated: zero once; from then on the kernel leaves every word it used at zero
                HIPCHK(c, pa->buf.reserve(units));
                HIPCHK(c, hipMemsetAsync(pa->buf.ptr, 0, units * 8, s));
            }
            ride_form = 2;
            ride_acc = pa->buf.ptr;
        } else if (c->opt_fused_pick == 3) {
            c->last_error = "the tile-test pick is not applicable to this request (attempts != 5, taints, more than eight label keys, or no room in LDS)";
            return KSCHED_E_UNSUPPORTED;
        }
    }
    c->last_pick = "none";
    if (select_direct && !pick_rides) {
        int rcs = launch_select(c, p, pcpu, pmem, psel, ptol, samples, attempts, flags, out_binding, s);
        if (rcs) return rcs;
        c->last_pick = "select";
        if (!want_mask) return KSCHED_OK;
    }
    // The best-fit pick likewise: from bitmaps kept in best-fit order (k_pick_bestfit_rows), no mask involved.
    const bool bestfit_rows = pick_b && !c->opt_pick_from_mask && c->bf_rows_built;
    if (bestfit_rows) {
        const IndexedLayout &l = c->idx.lay;
        BestfitRowsArgs q{};
        const bool sel = (flags & KSCHED_SEL) && psel && c->nkeys > 0;
        q.rows = c->bf_rows.ptr;
        q.lab_meta = c->idx.d_lab_meta;
        q.cpu_sorted = c->cpu_sorted.ptr;
        q.bf_mem = c->bf_mem.ptr;
        q.bf_cpu = c->bf_cpu.ptr;
        q.bf_order = c->bf_order.ptr;
        if (c->n <= 64u * 64u * 64u) {  // three rounds of 64 cover the array
            q.mem_s1 = c->bf_samples.ptr;
            q.mem_s2 = q.mem_s1 + c->bf_n1;
            q.cpu_s1 = q.mem_s2 + c->bf_n2;
            q.cpu_s2 = q.cpu_s1 + c->bf_n1;
        }
        q.pcpu = pcpu;
        q.pmem = pmem;
        q.psel = sel ? psel : nullptr;
        q.ptol = ptol;
        q.binding = out_binding;
        q.p = p;
        q.n = c->n;
        q.Wbf = c->bf_W;
        q.nkeys = sel ? c->nkeys : 0u;
        q.ngroups = l.ngroups;
        q.row_valid = l.row_valid;
        q.row_zero = l.row_zero;
        q.row_taint = l.row_taint;
        q.row_cpu0 = c->bf_row_cpu0;
        q.q = c->bf_q;
        q.do_fit = (flags & KSCHED_FIT) ? 1u : 0u;
        q.do_taint = ((flags & KSCHED_TAINT) && c->have_taints) ? 1u : 0u;
        for (int k = 0; k < 8; ++k) {
            q.lab_base8[k] = l.lab_base[k];
            q.lab_max8[k] = l.lab_max[k];
        }
        // one stage (a wave per pod) or two (a lane per pod first): the second launch and the hand-over list pay off from tens of
        // thousands of pods on (20k pods: 33 us against 44; 125k pods: 160 against 120) -- KSCHED_OPT_BESTFIT_STAGES overrides
        const bool lists = sel && l.nlist > 0;  // pods that constrain a list key are split off by the first stage: two stages it is
        const bool two_stage = lists || c->opt_bestfit_stages == 2 || (c->opt_bestfit_stages == 0 && p >= 24576u);  // (measured crossover at the C5 shard's snapshot: ~24 k pods)
        if (!lists && (!two_stage || (c->opt_debug & 0x400u) || c->n > (1u << 21))) {
            // one wave per pod, one wave per block (scans differ tenfold in length: with four waves per block the long ones hold up the
            // placement of whole blocks; 158 against 172 us for 125 k pods at the C5 shard)
            hipLaunchKernelGGL(k_pick_bestfit_rows, dim3(p), dim3(64), 0, s, q);
        } else {
            // two stages: one lane per pod decides from the first two candidate words; the rare rest goes to the wave-per-pod kernel
            if (c->n > (1u << 21)) {
                c->last_error = "best fit over a snapshot with list keys supports at most 2097152 nodes";
                return KSCHED_E_UNSUPPORTED;
            }
            if (int rsc = scratch_enter(c, s)) return rsc;
            // [3 sets of kBfSublists counters, one per 128-byte line, in rotation][listed mask: ceil(p / 64) words][64-byte hand-over records:
            // kBfSublists lists of sub_cap slots]; each call zeroes the NEXT call's counters (no memset launch)
            const size_t waves1 = ((size_t)p + 63) / 64, sub_cap = ((waves1 + kBfSublists - 1) / kBfSublists) * 64;
            const size_t ctr_u32 = 3 * (size_t)kBfSublists * 32, mask_u32 = ((waves1 * 2 + 15) / 16) * 16;  // (records stay 64-byte aligned)
            HIPCHK(c, c->bf_fallback.reserve(ctr_u32 + mask_u32 + 16 * (size_t)kBfSublists * sub_cap));
            if (c->bf_fallback_zeroed_cap != c->bf_fallback.cap) {  // a fresh allocation: all counters once
                HIPCHK(c, hipMemsetAsync(c->bf_fallback.ptr, 0, ctr_u32 * 4, s));
                c->bf_fallback_zeroed_cap = c->bf_fallback.cap;
                c->bf_slot = 0;
            }
            q.handover_count = c->bf_fallback.ptr + (size_t)c->bf_slot * kBfSublists * 32;
            q.zero_next = c->bf_fallback.ptr + (size_t)((c->bf_slot + 1u) % 3u) * kBfSublists * 32;
            q.sub_cap = (uint32_t)sub_cap;
            q.lvl = c->bf_levels.ptr;
            q.nlev = c->bf_nlev;
            q.lvl_half = c->bf_lvl_half;
            for (uint32_t k = 0; k < 6; ++k) q.lvl_off[k] = c->bf_lvl_off[k];
            q.handover_recs = c->bf_fallback.ptr + ctr_u32 + mask_u32;
            q.nlist = lists ? l.nlist : 0u;
            for (uint32_t j = 0; j < q.nlist; ++j) q.list_col[j] = l.list_col[j];
            q.listed_mask = lists ? reinterpret_cast<uint64_t *>(c->bf_fallback.ptr + ctr_u32) : nullptr;
            q.lane_blocks = ((c->opt_debug >> 12) & 15u) ? ((c->opt_debug >> 12) & 15u) : 2u;  // KSCHED_OPT_DEBUG bits 12-15: A/B of the hand-over point (in 64-byte blocks of 8 words)
            const bool tracing = (c->opt_debug & 0x100000u) != 0u;
            if (tracing) {
                HIPCHK(c, c->bf_trace.reserve(waves1 * 8 + (size_t)kBfSublists * sub_cap * 4));
                HIPCHK(c, hipMemsetAsync(c->bf_trace.ptr, 0, (waves1 * 8 + (size_t)kBfSublists * sub_cap * 4) * 8, s));
                q.trace = c->bf_trace.ptr;
                q.trace2 = c->bf_trace.ptr + waves1 * 8;
            }
            hipLaunchKernelGGL(k_pick_bestfit_lanes, dim3((p + 255) / 256), dim3(256), 0, s, q);
            HIPCHK(c, hipGetLastError());
            c->bf_slot = (c->bf_slot + 1u) % 3u;  // (only once the kernel that zeroes the next set is on its way)
            BestfitRowsArgs q2 = q;
            q2.sub_count = q.handover_count;
            q2.pod_recs = q.handover_recs;
            // one wave per block; the grid covers 1 / 2^k of the lists' capacity (KSCHED_OPT_DEBUG bits 21-22: k = 2 by default, A/B 0 / 1 / 3)
            const uint32_t gshift = ((c->opt_debug >> 21) & 3u) == 0u ? 2u : ((c->opt_debug >> 21) & 3u) == 1u ? 0u : ((c->opt_debug >> 21) & 3u) == 2u ? 1u : 3u;
            const uint32_t per_list = std::max<uint32_t>(1u, (uint32_t)sub_cap >> gshift);
            hipLaunchKernelGGL(k_pick_bestfit_handed, dim3(kBfSublists * per_list), dim3(64), 0, s, q2);
            if (tracing) bestfit_trace_report(c, p, (size_t)kBfSublists * sub_cap, s);
            if (lists) {
                BestfitListedArgs la{};
                la.lists = c->idx.d_list;
                la.nrec = c->nrec.ptr;
                la.nlab = c->nlab.ptr;
                la.bf_rank = c->bf_rank.ptr;
                la.bf_order = c->bf_order.ptr;
                la.pcpu = pcpu;
                la.pmem = pmem;
                la.psel = psel;
                la.ptol = ptol;
                la.listed_mask = q.listed_mask;
                la.binding = out_binding;
                la.p = p;
                la.n = c->n;
                la.nkeys = c->nkeys;
                la.tiles = l.tiles;
                la.nlist = l.nlist;
                for (uint32_t j = 0; j < l.nlist; ++j) la.list_col[j] = l.list_col[j];
                la.do_fit = q.do_fit;
                la.do_taint = q.do_taint;
                hipLaunchKernelGGL(k_pick_bestfit_listed, dim3((p + 3) / 4), dim3(256), 0, s, la);
            }
        }
        HIPCHK(c, hipGetLastError());
        c->last_pick = "bestfit-rows";
        if (!out_feas && !out_fit) return KSCHED_OK;
    }
    uint64_t *feas = out_feas;
    if (!feas) {  // the mask kernels always write the feasible mask: a pick that reads it, or a fit-mask-only request, gets a scratch one
        if (int rsc = scratch_enter(c, s)) return rsc;
        HIPCHK(c, c->scratch_mask.reserve((size_t)p * pitch));
        feas = c->scratch_mask.ptr;
    }

    // (the fused kernel carries its timing events on the dispatch packet itself -- hipExtLaunchKernel --, the multi-launch
    // paths are bracketed by stream events)
    int rc;
    if (kern == KSCHED_KERNEL_FUSED && !can_fused) {
        c->last_error = "fused kernel not applicable to this snapshot/request: " + (c->index_reason.empty() ? std::string("the bitmap index does not fit LDS") : c->index_reason);
        return KSCHED_E_UNSUPPORTED;
    }
    size_t slot = 0;
    // (at most 65536 unread samples: a caller that switches timing on and never reads it does not grow the event pool for ever)
    const bool timed = c->opt_timing && c->ev_used < 65536u && (c->timing_seq++ % c->opt_timing) == 0;
    if (timed) {
        int trc = timing_slot(c, &slot);
        if (trc) return trc;
        if (kern != KSCHED_KERNEL_FUSED) HIPCHK(c, hipEventRecord(c->ev_pool[slot].a, s));
    }
    if (kern == KSCHED_KERNEL_FUSED) {
        constexpr uint32_t kTraceBlocks = 8192;
        if (c->opt_trace) {
            DeviceGuard g2(c->device);
            if (int rsc = scratch_enter(c, s)) return rsc;
            HIPCHK(c, c->trace.reserve((size_t)kTraceBlocks * KSCHED_TRACE_WORDS));
            HIPCHK(c, hipMemsetAsync(c->trace.ptr, 0, (size_t)kTraceBlocks * KSCHED_TRACE_WORDS * 8, s));
        }
        hipError_t e = run_fused(c->idx, p, pcpu, pmem, psel, ptol, flags, feas, out_fit, pitch, s, c->opt_debug,
                                 timed ? c->ev_pool[slot].a : nullptr, timed ? c->ev_pool[slot].b : nullptr,
                                 c->opt_trace ? c->trace.ptr : nullptr, kTraceBlocks, pick_rides ? &ride : nullptr, ride_form, ride_acc,
                                 c->opt_grid_cus, c->opt_round_order);
        if (e != hipSuccess) return fail_hip(c, e, "run_fused");
        c->last_kernel = "fused";
        if (pick_rides) c->last_pick = ride_form == 2 ? "fused-tile" : "fused";
        rc = KSCHED_OK;
    } else {
        rc = run_direct(c, p, pcpu, pmem, psel, ptol, flags, feas, out_fit, pitch, s);
    }
    if (rc) return rc;
    if (timed && kern != KSCHED_KERNEL_FUSED) HIPCHK(c, hipEventRecord(c->ev_pool[slot].b, s));

    if (select_direct || bestfit_rows || !(pick_s || pick_b)) return KSCHED_OK;  // (a riding pick is a select_direct one)
    c->last_pick = "from-mask";
    return launch_pick(c, p, feas, pitch, pmem, samples, attempts, flags, out_binding, s);
}

int check_eval_args(const ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *samples,
                    uint32_t attempts, uint32_t flags, const uint64_t *out_feas, const uint64_t *out_fit,
                    const int32_t *out_binding) {
    const uint32_t known = KSCHED_FIT | KSCHED_SEL | KSCHED_TAINT | KSCHED_PICK_SAMPLED | KSCHED_PICK_BESTFIT |
                           KSCHED_WANT_FIT_MASK;
    if (flags & ~known) return KSCHED_E_INVAL;
    if ((flags & KSCHED_PICK_SAMPLED) && (flags & KSCHED_PICK_BESTFIT)) return KSCHED_E_INVAL;
    if (p > 0 && (!pcpu || !pmem)) return KSCHED_E_INVAL;
    if (flags & KSCHED_PICK_SAMPLED) {
        if (!out_binding || attempts == 0 || attempts > KSCHED_MAX_ATTEMPTS) return KSCHED_E_INVAL;
        if (p > 0 && !samples) return KSCHED_E_INVAL;
    }
    if ((flags & KSCHED_PICK_BESTFIT) && !out_binding) return KSCHED_E_INVAL;
    if ((flags & KSCHED_WANT_FIT_MASK) && !out_fit) return KSCHED_E_INVAL;
    if (!(flags & KSCHED_WANT_FIT_MASK) && out_fit) return KSCHED_E_INVAL;
    if (!(flags & (KSCHED_PICK_SAMPLED | KSCHED_PICK_BESTFIT)) && !out_feas && !out_fit) return KSCHED_E_INVAL;
    (void)c;
    return KSCHED_OK;
}

}  // namespace

extern "C" {

uint32_t ksched_abi_version(void) { return KSCHED_ABI_VERSION; }

int ksched_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

uint32_t ksched_mask_words(uint32_t n_nodes) { return (uint32_t)(((uint64_t)n_nodes + 63u) / 64u); }

const char *ksched_strerror(int code) {
    switch (code) {
        case KSCHED_OK: return "ok";
        case KSCHED_E_INVAL: return "invalid argument";
        case KSCHED_E_NODEVICE: return "no HIP device (this library has no CPU fallback)";
        case KSCHED_E_HIP: return "HIP runtime error";
        case KSCHED_E_NOMEM: return "out of memory";
        case KSCHED_E_STATE: return "ksched_set_nodes has not been called";
        case KSCHED_E_UNSUPPORTED: return "unsupported request";
        case KSCHED_E_RCCL: return "RCCL error (see ksched_comm_last_error)";
        default: return "unknown error";
    }
}

int ksched_create(ksched_ctx **out, int device_id) try {
    if (!out) return KSCHED_E_INVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return KSCHED_E_NODEVICE;
    if (device_id < 0 || device_id >= count) return KSCHED_E_INVAL;
    ksched_ctx *c = new (std::nothrow) ksched_ctx();
    if (!c) return KSCHED_E_NOMEM;
    c->device = device_id;
    DeviceGuard g(device_id);
    if (!g.ok || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_build, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_scratch, hipEventDisableTiming) != hipSuccess) {
        if (c->ev_build) (void)hipEventDestroy(c->ev_build);
        if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return KSCHED_E_HIP;
    }
    c->change_stream = c->stream;
    *out = c;
    return KSCHED_OK;
} KSCHED_ABI_CATCH(nullptr)

void ksched_destroy(ksched_ctx *c) try {
    if (!c) return;
    {
        DeviceGuard g(c->device);
        (void)hipDeviceSynchronize();
        c->ncpu.release(); c->nmem.release(); c->nrec.release(); c->nlab.release(); c->ntaint.release();
        c->bf_order.release(); c->bf_rank.release(); c->bf_mem.release(); c->bf_cpu.release(); c->cpu_sorted.release(); c->bf_rows.release(); c->bf_samples.release(); c->bf_levels.release(); c->bf_fallback.release(); c->bf_fallback_zeroed_cap = 0;
        c->pcpu.release(); c->pmem.release(); c->psel.release(); c->psamples.release();
        c->ptol.release(); c->feas.release(); c->fit.release(); c->binding.release(); c->gathered.release(); c->xpairs.release(); c->xreason.release();
        c->scratch_mask.release(); c->trace.release();
        c->by_cpu.release(); c->cpurank.release(); c->d_stage.release();
        for (int b = 0; b < 2; ++b) { c->srt_k0[b].release(); c->srt_k1[b].release(); c->srt_idx[b].release(); }
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        for (auto &x : c->pick_acc) x.buf.release();
        for (auto &u : c->user_streams) (void)hipEventDestroy(u.ev);
        if (c->ev_build) (void)hipEventDestroy(c->ev_build);
        if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
        if (c->ev_scratch) (void)hipEventDestroy(c->ev_scratch);
        indexed_release(c->idx);
        {  // masks the caller never handed back (ksched_mask_alloc)
            MaskRegistry &reg = mask_registry();
            std::lock_guard<std::mutex> rl(reg.mu);
            for (size_t i = reg.live.size(); i-- > 0;)
                if (reg.live[i].owner == c) {
                    (void)mask_release(reg.live[i]);
                    reg.live.erase(reg.live.begin() + (long)i);
                }
        }
        for (auto &ep : c->ev_pool) {
            (void)hipEventDestroy(ep.a);
            (void)hipEventDestroy(ep.b);
        }
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
} catch (...) {  // nothing unwinds across the C ABI
}

const char *ksched_last_error(const ksched_ctx *c) { return c ? c->last_error.c_str() : ""; }
uint32_t ksched_num_nodes(const ksched_ctx *c) { return (c && c->have_nodes) ? c->n : 0; }
uint32_t ksched_num_keys(const ksched_ctx *c) { return (c && c->have_nodes) ? c->nkeys : 0; }
const char *ksched_last_kernel(const ksched_ctx *c) { return c ? c->last_kernel : "none"; }
const char *ksched_last_pick(const ksched_ctx *c) { return c ? c->last_pick : "none"; }

int ksched_set_option(ksched_ctx *c, int option, int64_t value) try {
    if (!c) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    switch (option) {
        case KSCHED_OPT_KERNEL:
            if (value != KSCHED_KERNEL_AUTO && value != KSCHED_KERNEL_DIRECT && value != KSCHED_KERNEL_FUSED) return KSCHED_E_INVAL;
            c->opt_kernel = (int)value;
            return KSCHED_OK;
        case KSCHED_OPT_TIMING:
            if (value < 0 || value > 1000000) return KSCHED_E_INVAL;
            c->opt_timing = (uint32_t)value;
            c->timing_seq = 0;
            return KSCHED_OK;
        case KSCHED_OPT_DEBUG:
            c->opt_debug = (uint32_t)value;
            return KSCHED_OK;
        case KSCHED_OPT_TRACE:
            c->opt_trace = value != 0;
            return KSCHED_OK;
        case KSCHED_OPT_PICK_FROM_MASK:
            c->opt_pick_from_mask = value != 0;
            return KSCHED_OK;
        case KSCHED_OPT_BESTFIT_STAGES:
            if (value < 0 || value > 2) return KSCHED_E_INVAL;
            c->opt_bestfit_stages = (int)value;
            return KSCHED_OK;
        case KSCHED_OPT_INDEX_BUILD:
            if (value != 0 && value != 1) return KSCHED_E_INVAL;
            c->opt_index_build = (int)value;
            return KSCHED_OK;
        case KSCHED_OPT_SNAPSHOT_STREAM:
            if (value != 0 && value != 1) return KSCHED_E_INVAL;
            c->opt_own_stream = value == 1;
            return KSCHED_OK;
        case KSCHED_OPT_FUSED_PICK:
            if (value < 0 || value > 3) return KSCHED_E_INVAL;
            c->opt_fused_pick = (int)value;
            return KSCHED_OK;
        case KSCHED_OPT_PIPE_MODE:
            if (value < 0 || value > (int64_t)KSCHED_PIPE_MAX_STREAMS) return KSCHED_E_INVAL;
            c->opt_pipe_mode = (int)value;
            return KSCHED_OK;
        case KSCHED_OPT_GRID_CUS:
            if (value != 0 && (value < 8 || value > 256)) return KSCHED_E_INVAL;
            c->opt_grid_cus = (uint32_t)value;
            return KSCHED_OK;
        case KSCHED_OPT_ROUND_ORDER:
            if (value < 0 || value > 2) return KSCHED_E_INVAL;
            c->opt_round_order = (int)value;
            return KSCHED_OK;
        case KSCHED_OPT_MASK_PROBE:
            if (value < 1 || value > 16) return KSCHED_E_INVAL;
            c->opt_mask_probe = (uint32_t)value;
            return KSCHED_OK;
        case KSCHED_OPT_FAULT:  // low byte: 0 off, 1 std::bad_alloc, 2 std::runtime_error; bits 8..: fault points to pass first
            if (ksched_test_hooks_enabled == nullptr) {  // the shipped library: no fault injection (tests/cpp/test_hooks.cpp is not linked in)
                c->last_error = "KSCHED_OPT_FAULT exists in the test build of the library only";
                return KSCHED_E_UNSUPPORTED;
            }
            if (value < 0 || (value & 0xFF) > 2 || value > 0xFFFFFF) return KSCHED_E_INVAL;
            c->fault_kind = (uint32_t)(value & 0xFF);
            c->fault_skip = (uint32_t)(value >> 8);
            return KSCHED_OK;

        default:
            return KSCHED_E_INVAL;
    }
} KSCHED_ABI_CATCH(c)

int ksched_forget_stream(ksched_ctx *c, void *hip_stream) try {
    if (!c) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    stream_forget(c, (hipStream_t)hip_stream);
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

int ksched_set_nodes(ksched_ctx *c, uint32_t n, const int64_t *cpu, const int64_t *mem, const uint32_t *lab,
                     uint32_t n_keys, const uint64_t *taints) try {
    if (!c) return KSCHED_E_INVAL;
    if (n > 0 && (!cpu || !mem)) return KSCHED_E_INVAL;
    if (n_keys > KSCHED_MAX_KEYS) return KSCHED_E_INVAL;
    if (n_keys > 0 && n > 0 && !lab) return KSCHED_E_INVAL;
    // one pass over the label columns: the largest id per key (the index layout needs it) doubles as the validity check
    uint32_t lab_max[KSCHED_MAX_KEYS] = {};
    for (uint32_t k = 0; k < n_keys; ++k) {
        uint32_t mx = 0;
        const uint32_t *col = lab + (size_t)k * n;
        for (uint32_t i = 0; i < n; ++i) mx = std::max(mx, col[i]);
        if (mx == KSCHED_SEL_NEVER) return KSCHED_E_INVAL;
        lab_max[k] = mx;
    }
    uint64_t all_taints = 0;
    if (taints)
        for (uint32_t i = 0; i < n; ++i) all_taints |= taints[i];
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    c->have_nodes = false;  // stays false if anything below fails (or throws): a half-built snapshot is never evaluated
    fault_point(c);
    // evaluations already enqueued on the caller's streams read the previous snapshot: the ctx's stream waits for them
    // (events; the host does not)
    if (int rc = snapshot_begin(c)) return rc;
    c->n = n;
    c->nkeys = n_keys;
    c->W = ksched_mask_words(n);
    c->have_taints = taints != nullptr;
    c->bf_dirty = true;
    c->bf_rows_built = false;
    c->idx.built = false;
    HIPCHK(c, c->ncpu.reserve(n));
    HIPCHK(c, c->nmem.reserve(n));
    HIPCHK(c, c->nrec.reserve((size_t)n * kNodeRecWords));
    HIPCHK(c, c->nlab.reserve((size_t)n * n_keys));
    HIPCHK(c, c->ntaint.reserve(n));
    HIPCHK(c, c->bf_order.reserve(n));
    HIPCHK(c, c->bf_rank.reserve(n));
    HIPCHK(c, c->bf_mem.reserve((size_t)n + 8));  // (+8: the 8-ary searches read whole blocks of eight)
    HIPCHK(c, c->bf_cpu.reserve(n));
    HIPCHK(c, c->cpu_sorted.reserve((size_t)n + 8));
    hipStream_t s = c->change_stream;  // (snapshot_begin chose it)
    // the per-tile bitmap index: layout on the host (it fixes kernel arguments and LDS sizes), contents on the device
    IndexedLayout l{};
    const char *why = "";
    const bool indexed = indexed_plan(l, n, n_keys, lab_max, all_taints, &why);
    uint32_t meta[72] = {};
    if (indexed) indexed_meta(l, meta);
    // the caller's arrays (and the index's meta words) -> pinned staging -> asynchronous copies on the chosen stream; nothing is
    // copied from pageable or stack memory, so the call never waits for work already queued on that stream
    const size_t b_col = (size_t)n * 8, b_lab = (size_t)n * n_keys * 4, b_taint = taints ? b_col : 0, b_meta = indexed ? sizeof meta : 0;
    const size_t o_meta = 2 * b_col + b_lab + b_taint;
    uint8_t *h = nullptr;
    if (n > 0 || b_meta) {
        if (int rc = stage_reserve(c, o_meta + b_meta, &h)) return rc;
    }
    if (n > 0) {
        memcpy(h, cpu, b_col);
        memcpy(h + b_col, mem, b_col);
        if (b_lab) memcpy(h + 2 * b_col, lab, b_lab);
        if (b_taint) memcpy(h + 2 * b_col + b_lab, taints, b_taint);
        HIPCHK(c, hipMemcpyAsync(c->ncpu.ptr, h, b_col, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->nmem.ptr, h + b_col, b_col, hipMemcpyHostToDevice, s));
        if (b_lab) HIPCHK(c, hipMemcpyAsync(c->nlab.ptr, h + 2 * b_col, b_lab, hipMemcpyHostToDevice, s));
        if (b_taint) HIPCHK(c, hipMemcpyAsync(c->ntaint.ptr, h + 2 * b_col + b_lab, b_taint, hipMemcpyHostToDevice, s));
    }
    if (indexed) {
        hipError_t e = indexed_reserve(c->idx, l);
        if (e != hipSuccess) return fail_hip(c, e, "indexed_reserve");
        memcpy(h + o_meta, meta, b_meta);
        HIPCHK(c, hipMemcpyAsync(c->idx.d_lab_meta, h + o_meta, b_meta, hipMemcpyHostToDevice, s));
    }
    if (h) HIPCHK(c, hipEventRecord(c->ev_stage, s));  // the last copy out of the staging block
    if (n > 0) {
        hipLaunchKernelGGL(k_build_nrec, dim3((n + 255u) / 256u), dim3(256), 0, s, (const int64_t *)c->ncpu.ptr, (const int64_t *)c->nmem.ptr,
                           taints ? (const uint64_t *)c->ntaint.ptr : nullptr, (const uint32_t *)c->nlab.ptr, n_keys, c->nrec.ptr, n);
        HIPCHK(c, hipGetLastError());
    }
    if (indexed) {
        c->idx.lay = l;
        if (c->opt_index_build == 1) {
            hipError_t e = indexed_build_host(c->idx, l, cpu, mem, lab, taints, s);
            if (e != hipSuccess) return fail_hip(c, e, "indexed_build_host");
        } else {
            if (int rc = launch_build_named(c)) return rc;
            if (int rc = launch_build_lists(c)) return rc;
            if (int rc = launch_build_fit(c, nullptr, 0)) return rc;
        }
        c->idx.built = true;
        c->index_reason.clear();
    } else {
        c->index_reason = why;
    }
    if (int rc = snapshot_end(c)) return rc;
    c->have_nodes = true;
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

int ksched_update_nodes(ksched_ctx *c, uint32_t count, const uint32_t *node_index, const int64_t *cpu, const int64_t *mem) try {
    if (!c) return KSCHED_E_INVAL;
    if (count > 0 && (!node_index || !cpu || !mem)) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    for (uint32_t i = 0; i < count; ++i)
        if (node_index[i] >= c->n) return KSCHED_E_INVAL;
    if (count == 0) return KSCHED_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    fault_point(c);  // (nothing has changed yet: an exception up to snapshot_begin leaves the snapshot as it was)
    // a node listed twice takes its last values: keep the last occurrence of every index (the patch kernel's threads are unordered)
    std::vector<uint32_t> keep;
    if (count > 1) {
        std::vector<uint32_t> order(count);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return node_index[a] < node_index[b]; });
        for (uint32_t i = 0; i < count; ++i)
            if (i + 1 == count || node_index[order[i + 1]] != node_index[order[i]]) keep.push_back(order[i]);
    } else {
        keep.push_back(0u);
    }
    const uint32_t m = (uint32_t)keep.size();
    std::vector<uint32_t> tiles;
    for (uint32_t i : keep) tiles.push_back(node_index[i] / kTileNodes);  // ascending already
    tiles.erase(std::unique(tiles.begin(), tiles.end()), tiles.end());
    // evaluations already enqueued read the snapshot as it was (events, no host wait)
    if (int rc = snapshot_begin(c)) return rc;
    hipStream_t s = c->change_stream;  // (snapshot_begin chose it)
    auto fail = [&](int rc) {
        c->have_nodes = false;  // columns, index and best-fit order may now disagree: refuse evaluations until the next ksched_set_nodes
        return rc;
    };
    PatchArgs pa{};
    pa.ncpu = c->ncpu.ptr;
    pa.nmem = c->nmem.ptr;
    pa.nrec = c->nrec.ptr;
    pa.count = m;
    const uint32_t *d_tiles = nullptr;
    const bool want_tiles = c->idx.built;
    if (m <= kPatchInline && tiles.size() <= kPatchInline) {
        // small updates (the watch-event case) travel in kernel arguments: no copy, no staging buffer
        for (uint32_t j = 0; j < m; ++j) {
            pa.idx_in[j] = node_index[keep[j]];
            pa.cpu_in[j] = cpu[keep[j]];
            pa.mem_in[j] = mem[keep[j]];
        }
    } else {
        const size_t b_idx = ((size_t)m * 4 + 7) & ~(size_t)7, b_val = (size_t)m * 8, b_tiles = tiles.size() * 4;
        uint8_t *h = nullptr;
        if (int rc = stage_reserve(c, b_idx + 2 * b_val + b_tiles, &h)) return fail(rc);
        uint32_t *hi = reinterpret_cast<uint32_t *>(h);
        int64_t *hc = reinterpret_cast<int64_t *>(h + b_idx), *hm = hc + m;
        for (uint32_t j = 0; j < m; ++j) {
            hi[j] = node_index[keep[j]];
            hc[j] = cpu[keep[j]];
            hm[j] = mem[keep[j]];
        }
        memcpy(h + b_idx + 2 * b_val, tiles.data(), b_tiles);
        if (c->d_stage.reserve(b_idx + 2 * b_val + b_tiles) != hipSuccess) return fail(KSCHED_E_NOMEM);
        if (hipMemcpyAsync(c->d_stage.ptr, h, b_idx + 2 * b_val + b_tiles, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipEventRecord(c->ev_stage, s) != hipSuccess)
            return fail(KSCHED_E_HIP);
        pa.idx = reinterpret_cast<const uint32_t *>(c->d_stage.ptr);
        pa.cpu = reinterpret_cast<const int64_t *>(c->d_stage.ptr + b_idx);
        pa.mem = pa.cpu + m;
        d_tiles = reinterpret_cast<const uint32_t *>(c->d_stage.ptr + b_idx + 2 * b_val);
    }
    if (want_tiles && !d_tiles) {  // small update: the tile list rides in the patch kernel's arguments
        if (c->d_stage.reserve(kPatchInline * 4) != hipSuccess) return fail(KSCHED_E_NOMEM);
        pa.tile_out = reinterpret_cast<uint32_t *>(c->d_stage.ptr);
        pa.ntiles = (uint32_t)tiles.size();
        for (size_t j = 0; j < tiles.size(); ++j) pa.tiles_in[j] = tiles[j];
        d_tiles = pa.tile_out;
    }
    hipLaunchKernelGGL(k_patch_nodes, dim3((m + 255u) / 256u), dim3(256), 0, s, pa);
    if (hipGetLastError() != hipSuccess) return fail(KSCHED_E_HIP);
    if (want_tiles) {
        // only the touched 1024-node tiles are re-indexed (fit rows, search trees, cnt tables); label and taint rows are untouched
        if (int rc = launch_build_fit(c, d_tiles, (uint32_t)tiles.size())) return fail(rc);
    }
    c->bf_dirty = true;  // the best-fit order is rebuilt by the next PICK_BESTFIT request, not here
    if (int rc = snapshot_end(c)) return fail(rc);
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

int ksched_eval_device_pitched(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                               const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags,
                               uint64_t *out_feas, uint64_t *out_fit, int32_t *out_binding, uint32_t mask_pitch_words,
                               void *hip_stream) try {
    if (!c) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    int rc = check_eval_args(c, p, pcpu, pmem, samples, attempts, flags, out_feas, out_fit, out_binding);
    if (rc) return rc;
    if (mask_pitch_words < c->W) return KSCHED_E_INVAL;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    return eval_on_device(c, p, pcpu, pmem, psel, ptol, samples, attempts, flags, out_feas, out_fit, out_binding,
                          mask_pitch_words, (hipStream_t)hip_stream);
} KSCHED_ABI_CATCH(c)

int ksched_eval_device(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                       const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags,
                       uint64_t *out_feas, uint64_t *out_fit, int32_t *out_binding, void *hip_stream) try {
    return ksched_eval_device_pitched(c, p, pcpu, pmem, psel, ptol, samples, attempts, flags, out_feas, out_fit, out_binding,
                                      ksched_mask_words(ksched_num_nodes(c)), hip_stream);
} KSCHED_ABI_CATCH(c)

uint32_t ksched_mask_pitch(uint32_t n_nodes) { return (ksched_mask_words(n_nodes) + 15u) & ~15u; }

struct ksched_pipe {
    ksched_ctx *ctx = nullptr;
    uint32_t depth = 0;
    hipStream_t s_mask = nullptr, s_pick = nullptr;
    std::vector<hipStream_t> extra;  // streams 2 .. of the alternate mode over more than two streams (created on first use)
    std::vector<hipEvent_t> mask_done, pick_done;
    std::vector<hipStream_t> slot_stream;  // the stream that carries the slot's mask kernel (alternate mode: also its pick)
    std::vector<hipStream_t> pick_stream;  // the stream that carried the slot's latest pick (ksched_pipe_slot_stream)
    std::vector<uint8_t> last_split;       // the slot's latest submit ran in the split mode (its mask kernel and its pick on different streams)
};

int ksched_pipe_create(ksched_ctx *c, uint32_t depth, ksched_pipe **out) try {
    if (!c || !out || depth == 0 || depth > 64) return KSCHED_E_INVAL;  // (a slot is two events and the caller's buffers: 64 is plenty for "enough masks in flight to exceed the Infinity Cache")
    *out = nullptr;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    ksched_pipe *q = new (std::nothrow) ksched_pipe();
    if (!q) return KSCHED_E_NOMEM;
    q->ctx = c;
    q->depth = depth;
    bool ok = hipStreamCreateWithFlags(&q->s_mask, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&q->s_pick, hipStreamNonBlocking) == hipSuccess;
    q->slot_stream.assign(depth, nullptr);
    q->pick_stream.assign(depth, nullptr);
    q->last_split.assign(depth, 0);
    for (uint32_t i = 0; ok && i < 2 * depth; ++i) {
        hipEvent_t e;
        ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (ok) (i < depth ? q->mask_done : q->pick_done).push_back(e);
    }
    if (!ok) {
        ksched_pipe_destroy(q);
        return KSCHED_E_HIP;
    }
    *out = q;
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

void ksched_pipe_destroy(ksched_pipe *q) try {
    if (!q) return;
    {
        DeviceGuard g(q->ctx->device);
        if (q->s_mask) (void)hipStreamSynchronize(q->s_mask);
        if (q->s_pick) (void)hipStreamSynchronize(q->s_pick);
        for (auto st : q->extra) (void)hipStreamSynchronize(st);
        {
            std::lock_guard<std::mutex> lk(q->ctx->mu);  // the ctx must not record events on streams that are about to go away
            stream_forget(q->ctx, q->s_mask);
            stream_forget(q->ctx, q->s_pick);
            for (auto st : q->extra) stream_forget(q->ctx, st);
        }
        for (auto e : q->mask_done) (void)hipEventDestroy(e);
        for (auto e : q->pick_done) (void)hipEventDestroy(e);
        if (q->s_mask) (void)hipStreamDestroy(q->s_mask);
        if (q->s_pick) (void)hipStreamDestroy(q->s_pick);
        for (auto st : q->extra) (void)hipStreamDestroy(st);
    }
    delete q;
} catch (...) {  // nothing unwinds across the C ABI
}

void *ksched_pipe_stream(ksched_pipe *q, int which) try {
    if (!q || which < 0) return nullptr;
    if (which < 2) return (void *)(which == 0 ? q->s_mask : q->s_pick);
    std::lock_guard<std::mutex> lk(q->ctx->mu);  // (ksched_pipe_submit grows `extra` under the same lock)
    return (size_t)(which - 2) < q->extra.size() ? (void *)q->extra[(size_t)(which - 2)] : nullptr;
} catch (...) {
    return nullptr;
}

// the stream that carried the slot's latest evaluation in the alternate mode / its pick in the split mode: what a consumer of the
// slot's bindings (the all-gather) has to be enqueued behind
void *ksched_pipe_slot_stream(ksched_pipe *q, uint32_t slot) try {
    if (!q || slot >= q->depth) return nullptr;
    std::lock_guard<std::mutex> lk(q->ctx->mu);
    return (void *)q->pick_stream[slot];
} catch (...) {
    return nullptr;
}

int ksched_pipe_submit(ksched_pipe *q, uint32_t slot, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                       const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags, uint64_t *mask,
                       uint32_t mask_pitch_words, int32_t *binding) try {
    if (!q || slot >= q->depth) return KSCHED_E_INVAL;
    ksched_ctx *c = q->ctx;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    const uint32_t pick = flags & (KSCHED_PICK_SAMPLED | KSCHED_PICK_BESTFIT);
    if (!pick || (flags & KSCHED_WANT_FIT_MASK) || !mask) return KSCHED_E_INVAL;
    int rc = check_eval_args(c, p, pcpu, pmem, samples, attempts, flags, mask, nullptr, binding);
    if (rc) return rc;
    if (mask_pitch_words < c->W) return KSCHED_E_INVAL;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    // Does the pick read the mask?  By default it does not (sampled: the drawn candidates are tested from the columns;
    // best fit: bitmaps kept in best-fit order), so the two streams need no ordering at all: each is in order by itself
    // (mask kernels of successive batches on one, picks on the other), which also covers the reuse of a slot's buffers.
    const bool pick_reads_mask = c->opt_pick_from_mask || ((pick & KSCHED_PICK_BESTFIT) && !bf_rows_expected(c));
    if (c->opt_pipe_mode >= 1 && !pick_reads_mask) {
        // alternate: the whole evaluation of the slot on ONE of the pipe's streams (one launch when the pick rides in the mask
        // kernel), stream = slot mod k.  A slot comes back to the same stream as long as the mode stays what it is, so the reuse of
        // its buffers is ordered by the stream itself; when the slot's previous use ran elsewhere -- another k, or the split mode,
        // whose mask kernel sits on the mask stream un-ordered against anything -- its new stream first waits for that use.
        const uint32_t k = std::max(2u, (uint32_t)c->opt_pipe_mode), which = slot % k;
        while (which >= 2u && q->extra.size() < (size_t)which - 1u) {
            hipStream_t ns = nullptr;
            HIPCHK(c, hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
            q->extra.push_back(ns);
        }
        hipStream_t st = which == 0u ? q->s_mask : which == 1u ? q->s_pick : q->extra[which - 2u];
        if (q->pick_stream[slot] && (q->pick_stream[slot] != st || q->last_split[slot])) HIPCHK(c, hipStreamWaitEvent(st, q->pick_done[slot], 0));
        if (q->last_split[slot] && q->slot_stream[slot] != st) HIPCHK(c, hipStreamWaitEvent(st, q->mask_done[slot], 0));  // (the split mode records it after every mask kernel)
        q->last_split[slot] = 0;
        q->slot_stream[slot] = st;
        q->pick_stream[slot] = st;
        rc = eval_on_device(c, p, pcpu, pmem, psel, ptol, samples, attempts, flags, mask, nullptr, binding, mask_pitch_words, st);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(q->pick_done[slot], st));
        return KSCHED_OK;
    }
    hipStream_t sm = q->s_mask;
    if (q->pick_stream[slot] && q->pick_stream[slot] != q->s_pick) {  // the slot last ran in the alternate mode on another stream
        HIPCHK(c, hipStreamWaitEvent(q->s_pick, q->pick_done[slot], 0));
        HIPCHK(c, hipStreamWaitEvent(sm, q->pick_done[slot], 0));
    }
    q->slot_stream[slot] = sm;
    q->pick_stream[slot] = q->s_pick;
    if (pick_reads_mask) HIPCHK(c, hipStreamWaitEvent(sm, q->pick_done[slot], 0));  // the slot's mask may be overwritten once its pick has run
    q->last_split[slot] = 1;
    rc = eval_on_device(c, p, pcpu, pmem, psel, ptol, nullptr, 0, flags & ~pick, mask, nullptr, nullptr, mask_pitch_words, sm);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(q->mask_done[slot], sm));  // (always: a later submit of this slot in the alternate mode orders itself behind this mask kernel)
    if (pick_reads_mask) {
        HIPCHK(c, hipStreamWaitEvent(q->s_pick, q->mask_done[slot], 0));
        if (p > 0) {
            if (c->n == 0) HIPCHK(c, hipMemsetAsync(binding, 0xFF, (size_t)p * sizeof(int32_t), q->s_pick));
            else if ((rc = launch_pick(c, p, mask, mask_pitch_words, pmem, samples, attempts, flags, binding, q->s_pick))) return rc;
        }
    } else {
        // bindings-only evaluation on the pick stream: launches the pick kernel alone
        rc = eval_on_device(c, p, pcpu, pmem, psel, ptol, samples, attempts, flags, nullptr, nullptr, binding, mask_pitch_words, q->s_pick);
        if (rc) return rc;
    }
    HIPCHK(c, hipEventRecord(q->pick_done[slot], q->s_pick));
    return KSCHED_OK;
} KSCHED_ABI_CATCH((q ? q->ctx : nullptr))

int ksched_pipe_wait(ksched_pipe *q, uint32_t slot, void *hip_stream) try {
    if (!q || slot >= q->depth) return KSCHED_E_INVAL;
    ksched_ctx *c = q->ctx;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    if (hip_stream) HIPCHK(c, hipStreamWaitEvent((hipStream_t)hip_stream, q->pick_done[slot], 0));
    else HIPCHK(c, hipEventSynchronize(q->pick_done[slot]));
    return KSCHED_OK;
} KSCHED_ABI_CATCH((q ? q->ctx : nullptr))

int ksched_pipe_wait_mask(ksched_pipe *q, uint32_t slot, void *hip_stream) try {
    if (!q || slot >= q->depth) return KSCHED_E_INVAL;
    ksched_ctx *c = q->ctx;
    std::lock_guard<std::mutex> lk(c->mu);  // (ksched_pipe_submit records the same events)
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    // The mask stream is in order: "everything enqueued on it so far" contains the slot's latest mask kernel.  Recorded here, on
    // demand, so that the submit path carries no event for consumers that never read the masks.
    HIPCHK(c, hipEventRecord(q->mask_done[slot], q->slot_stream[slot] ? q->slot_stream[slot] : q->s_mask));
    if (hip_stream) HIPCHK(c, hipStreamWaitEvent((hipStream_t)hip_stream, q->mask_done[slot], 0));
    else HIPCHK(c, hipEventSynchronize(q->mask_done[slot]));
    return KSCHED_OK;
} KSCHED_ABI_CATCH((q ? q->ctx : nullptr))

int ksched_pick_device(ksched_ctx *c, uint32_t p, const uint64_t *feasible, uint32_t mask_pitch_words, const int64_t *req_mem_bytes,
                       const uint32_t *samples, uint32_t attempts, uint32_t flags, int32_t *out_binding, void *hip_stream) try {
    if (!c) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    const bool pick_s = flags & KSCHED_PICK_SAMPLED, pick_b = flags & KSCHED_PICK_BESTFIT;
    if (pick_s == pick_b || (flags & ~(KSCHED_PICK_SAMPLED | KSCHED_PICK_BESTFIT | KSCHED_FIT | KSCHED_SEL | KSCHED_TAINT))) return KSCHED_E_INVAL;
    if (!out_binding || (p > 0 && c->n > 0 && !feasible) || mask_pitch_words < c->W) return KSCHED_E_INVAL;
    if (pick_s && (attempts == 0 || attempts > KSCHED_MAX_ATTEMPTS || (p > 0 && !samples))) return KSCHED_E_INVAL;
    if (pick_b && (flags & KSCHED_FIT) && p > 0 && !req_mem_bytes) return KSCHED_E_INVAL;
    if (p == 0) return KSCHED_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    hipStream_t s = (hipStream_t)hip_stream;
    if (c->n == 0) {  // choose() on an empty store yields None on every attempt (src/main.rs:56,70)
        HIPCHK(c, hipMemsetAsync(out_binding, 0xFF, (size_t)p * sizeof(int32_t), s));
        return KSCHED_OK;
    }
    return launch_pick(c, p, feasible, mask_pitch_words, req_mem_bytes, samples, attempts, flags, out_binding, s);
} KSCHED_ABI_CATCH(c)

// The pick alone from HOST masks (packed rows of W words): copies in, ksched_pick_device on the ctx's stream, copies out, waits.
int ksched_pick(ksched_ctx *c, uint32_t p, const uint64_t *feasible, const int64_t *req_mem_bytes, const uint32_t *samples, uint32_t attempts,
                uint32_t flags, int32_t *out_binding) try {
    if (!c) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    const bool pick_s = flags & KSCHED_PICK_SAMPLED, pick_b = flags & KSCHED_PICK_BESTFIT;
    if (pick_s == pick_b || (flags & ~(KSCHED_PICK_SAMPLED | KSCHED_PICK_BESTFIT | KSCHED_FIT | KSCHED_SEL | KSCHED_TAINT))) return KSCHED_E_INVAL;
    if (!out_binding || (p > 0 && c->n > 0 && !feasible)) return KSCHED_E_INVAL;
    if (pick_s && (attempts == 0 || attempts > KSCHED_MAX_ATTEMPTS || (p > 0 && !samples))) return KSCHED_E_INVAL;
    if (pick_b && (flags & KSCHED_FIT) && p > 0 && !req_mem_bytes) return KSCHED_E_INVAL;
    if (p == 0) return KSCHED_OK;
    if (c->n == 0) {  // choose() on an empty store yields None on every attempt (src/main.rs:56,70)
        for (uint32_t i = 0; i < p; ++i) out_binding[i] = -1;
        return KSCHED_OK;
    }
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    fault_point(c);
    hipStream_t s = c->stream;
    if (pick_b)
        if (int rcb = ensure_bestfit(c)) return rcb;
    if (int rce = stream_enter(c, s)) return rce;  // behind the latest snapshot change
    const size_t W = c->W;
    HIPCHK(c, c->feas.reserve((size_t)p * W));
    HIPCHK(c, c->binding.reserve(p));
    HIPCHK(c, hipMemcpyAsync(c->feas.ptr, feasible, (size_t)p * W * 8, hipMemcpyHostToDevice, s));
    if (req_mem_bytes) {
        HIPCHK(c, c->pmem.reserve(p));
        HIPCHK(c, hipMemcpyAsync(c->pmem.ptr, req_mem_bytes, (size_t)p * 8, hipMemcpyHostToDevice, s));
    }
    if (pick_s) {
        HIPCHK(c, c->psamples.reserve((size_t)p * attempts));
        HIPCHK(c, hipMemcpyAsync(c->psamples.ptr, samples, (size_t)p * attempts * 4, hipMemcpyHostToDevice, s));
    }
    if (int rc = launch_pick(c, p, c->feas.ptr, (uint32_t)W, req_mem_bytes ? c->pmem.ptr : nullptr, pick_s ? c->psamples.ptr : nullptr, attempts, flags,
                             c->binding.ptr, s))
        return rc;
    HIPCHK(c, hipMemcpyAsync(out_binding, c->binding.ptr, (size_t)p * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

// ---- the host-pointer evaluation in two halves (include/ksched.h "one host thread, several devices") -------------------------
// eval_begin_locked: copy the batch in, enqueue the evaluation and the copies of the masks back -- all on the ctx's own stream, no
// host wait.  The bindings stay in a ctx-owned device buffer of `capacity` >= p entries; entries [p, capacity) are -1 (the padding
// an all-gather of unequal shards needs).  The caller holds the ctx's mutex and has checked the arguments.
namespace {
int eval_begin_locked(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel, uint32_t sel_stride,
                      const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags, uint64_t *out_feas,
                      uint64_t *out_fit, uint32_t capacity, int32_t **binding_dev) {
    hipStream_t s = c->stream;
    const size_t W = c->W;
    const size_t pitch = ksched_mask_pitch(c->n);
    const bool pick = flags & (KSCHED_PICK_SAMPLED | KSCHED_PICK_BESTFIT);
    const bool use_sel = (flags & KSCHED_SEL) && psel && c->nkeys > 0;
    const bool use_tol = (flags & KSCHED_TAINT) && ptol;
    int32_t *d_bind = nullptr;
    if (pick) {
        const size_t cap = std::max<size_t>(capacity, p);
        HIPCHK(c, c->binding.reserve(cap));
        d_bind = c->binding.ptr;
        if (cap > p) HIPCHK(c, hipMemsetAsync(d_bind + p, 0xFF, (cap - p) * sizeof(int32_t), s));  // -1: "no node" rows past this shard's end
    }
    if (binding_dev) *binding_dev = d_bind;
    if (p == 0) return KSCHED_OK;  // (a rank whose shard is empty still takes part in the exchange with `capacity` rows of -1)

    HIPCHK(c, c->pcpu.reserve(p));
    HIPCHK(c, c->pmem.reserve(p));
    HIPCHK(c, hipMemcpyAsync(c->pcpu.ptr, pcpu, (size_t)p * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->pmem.ptr, pmem, (size_t)p * 8, hipMemcpyHostToDevice, s));
    if (use_sel) {
        HIPCHK(c, c->psel.reserve((size_t)p * c->nkeys));
        if (sel_stride == p)
            HIPCHK(c, hipMemcpyAsync(c->psel.ptr, psel, (size_t)p * c->nkeys * 4, hipMemcpyHostToDevice, s));
        else  // rows [lo, lo + p) of a [n_keys][sel_stride] array: column k of the shard starts sel_stride entries after column k - 1's
            HIPCHK(c, hipMemcpy2DAsync(c->psel.ptr, (size_t)p * 4, psel, (size_t)sel_stride * 4, (size_t)p * 4, c->nkeys, hipMemcpyHostToDevice, s));
    }
    if (use_tol) {
        HIPCHK(c, c->ptol.reserve(p));
        HIPCHK(c, hipMemcpyAsync(c->ptol.ptr, ptol, (size_t)p * 8, hipMemcpyHostToDevice, s));
    }
    if (flags & KSCHED_PICK_SAMPLED) {
        HIPCHK(c, c->psamples.reserve((size_t)p * attempts));
        HIPCHK(c, hipMemcpyAsync(c->psamples.ptr, samples, (size_t)p * attempts * 4, hipMemcpyHostToDevice, s));
    }
    uint64_t *d_feas = nullptr, *d_fit = nullptr;
    // a mask is needed when the caller wants it, or when the pick reads it (best fit; sampled only with KSCHED_OPT_PICK_FROM_MASK)
    const bool pick_reads_mask = (flags & (KSCHED_PICK_BESTFIT | KSCHED_PICK_SAMPLED)) &&
                                 (c->opt_pick_from_mask || ((flags & KSCHED_PICK_BESTFIT) && !bf_rows_expected(c)));
    if (out_feas || pick_reads_mask) {
        HIPCHK(c, c->feas.reserve((size_t)p * pitch));
        d_feas = c->feas.ptr;
    }
    if (out_fit) {
        HIPCHK(c, c->fit.reserve((size_t)p * pitch));
        d_fit = c->fit.ptr;
    }
    int rc = eval_on_device(c, p, c->pcpu.ptr, c->pmem.ptr, use_sel ? c->psel.ptr : nullptr, use_tol ? c->ptol.ptr : nullptr,
                            c->psamples.ptr, attempts, flags, d_feas, d_fit, d_bind, (uint32_t)pitch, s);
    if (rc) return rc;
    // device rows are line-aligned (pitch words apart); the caller's rows are packed (W words)
    if (out_feas && W)
        HIPCHK(c, hipMemcpy2DAsync(out_feas, W * 8, d_feas, pitch * 8, W * 8, p, hipMemcpyDeviceToHost, s));
    if (out_fit && W)
        HIPCHK(c, hipMemcpy2DAsync(out_fit, W * 8, d_fit, pitch * 8, W * 8, p, hipMemcpyDeviceToHost, s));
    return KSCHED_OK;
}
}  // namespace

int ksched_eval(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags, uint64_t *out_feas,
                uint64_t *out_fit, int32_t *out_binding) try {
    if (!c) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    int rc = check_eval_args(c, p, pcpu, pmem, samples, attempts, flags, out_feas, out_fit, out_binding);
    if (rc) return rc;
    if (p == 0) return KSCHED_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    int32_t *d_bind = nullptr;
    rc = eval_begin_locked(c, p, pcpu, pmem, psel, p, ptol, samples, attempts, flags, out_feas, out_fit, p, &d_bind);
    if (rc) return rc;
    if (d_bind && out_binding) HIPCHK(c, hipMemcpyAsync(out_binding, d_bind, (size_t)p * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

void ksched_shard_bounds(uint32_t p, uint32_t nranks, uint32_t rank, uint32_t *lo, uint32_t *hi, uint32_t *count_per_rank) {
    // rank r owns pod rows [r * shard, min(p, (r + 1) * shard)), shard = ceil(p / nranks): the last ranks may own fewer rows, or none
    const uint64_t shard = nranks ? ((uint64_t)p + nranks - 1u) / nranks : p;
    const uint64_t l = std::min<uint64_t>(p, shard * rank), h = std::min<uint64_t>(p, l + shard);
    if (lo) *lo = (uint32_t)l;
    if (hi) *hi = (uint32_t)h;
    if (count_per_rank) *count_per_rank = (uint32_t)shard;
}

int ksched_eval_begin(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel, uint32_t sel_stride,
                      const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags, uint64_t *out_feas,
                      uint64_t *out_fit, uint32_t binding_capacity, int32_t **binding_dev, void **hip_stream) try {
    if (!c || !binding_dev || !hip_stream) return KSCHED_E_INVAL;
    *binding_dev = nullptr;
    *hip_stream = nullptr;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    if (psel && sel_stride < p) return KSCHED_E_INVAL;
    int32_t sentinel = 0;  // (the bindings stay on the device: check_eval_args only wants to know that a pick has somewhere to go)
    int rc = check_eval_args(c, p, pcpu, pmem, samples, attempts, flags, out_feas, out_fit,
                             (flags & (KSCHED_PICK_SAMPLED | KSCHED_PICK_BESTFIT)) ? &sentinel : nullptr);
    if (rc) return rc;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    rc = eval_begin_locked(c, p, pcpu, pmem, psel, sel_stride, ptol, samples, attempts, flags, out_feas, out_fit, binding_capacity, binding_dev);
    if (rc) return rc;
    *hip_stream = (void *)c->stream;
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

// ---- mask buffers owned by the library (mask_alloc.hpp; profiles/r06_mask_alloc.md) ---------------------------------------------
namespace {
// one buffer through one allocation path (never AUTO / PROBE).  *unsupported: a measurement path asked of the shipped library.
hipError_t mask_alloc_path(ksched_ctx *c, size_t bytes, uint32_t how, MaskAllocation &a, bool *unsupported = nullptr) {
    a = MaskAllocation();
    a.bytes = bytes;
    a.device = c->device;
    a.owner = c;
    a.how = how;
    if (how == KSCHED_MASK_ALLOC_PLAIN) return hipMalloc(&a.ptr, bytes);
    // VMM, VMM_MIN, CONTIGUOUS, SCATTER_*: tests/cpp/test_hooks.cpp, linked into the test build only (mask_alloc.hpp)
    if (ksched_test_mask_alloc == nullptr) {
        if (unsupported) *unsupported = true;
        return hipErrorNotSupported;
    }
    const int rc = ksched_test_mask_alloc(c->device, bytes, how, &a.ptr, &a.test_token);
    return rc == 0 ? hipSuccess : rc > 0 ? (hipError_t)rc : hipErrorInvalidValue;
}

// Probe-and-keep (KSCHED_MASK_ALLOC_PROBE).  The rate of the mask kernel into a buffer is a property of the buffer's physical placement
// (HBM-side write stalls, TCC_EA0_WRREQ_DRAM_CREDIT_STALL: profiles/r06_mask_alloc.md) that no allocation path selects and user space
// cannot see -- but it is sticky per allocation and the kernel itself measures it: `k` candidates over different paths, ALL alive at once
// (a freed buffer's pages come straight back), the fused mask kernel timed into each (fit only, zero requests: every pair feasible
// or not by the node's sign, the same 128-byte segments at the same addresses as any other instantiation), the fastest kept.
int mask_alloc_probe(ksched_ctx *c, uint32_t p, uint32_t pitch, size_t bytes, MaskAllocation &best) {
    // (hipMalloc only: the virtual-memory paths gave the same ladder of rates and showed stale reads on a mapping's first use after memory-pool
    // activity in the process -- profiles/r06_mask_alloc.md section 3; candidates kept alive side by side land on different physical blocks anyway)
    static const uint32_t paths[] = {KSCHED_MASK_ALLOC_PLAIN, KSCHED_MASK_ALLOC_PLAIN, KSCHED_MASK_ALLOC_PLAIN, KSCHED_MASK_ALLOC_PLAIN};  // (hipMalloc and nothing else)
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    const uint32_t k = (uint32_t)std::max<size_t>(1, std::min<size_t>(c->opt_mask_probe, free_b / 4 / std::max<size_t>(bytes, 1)));
    std::vector<MaskAllocation> cand;
    auto release_all = [&](size_t keep) {
        for (size_t i = 0; i < cand.size(); ++i)
            if (i != keep) (void)mask_release(cand[i]);
    };
    for (uint32_t i = 0; i < k; ++i) {
        MaskAllocation a;
        hipError_t e = mask_alloc_path(c, bytes, paths[i % 4u], a);
        if (e != hipSuccess || !a.ptr) {
            (void)hipGetLastError();
            if (!cand.empty()) break;  // fewer candidates than asked for
            e = mask_alloc_path(c, bytes, KSCHED_MASK_ALLOC_PLAIN, a);
            if (e != hipSuccess || !a.ptr) {
                c->last_error = std::string("ksched_mask_alloc: ") + hipGetErrorString(e);
                (void)hipGetLastError();
                return e == hipErrorOutOfMemory ? KSCHED_E_NOMEM : KSCHED_E_HIP;
            }
        }
        cand.push_back(a);
    }
    size_t keep = 0;
    std::fill(std::begin(c->mask_probe_us), std::end(c->mask_probe_us), 0.0);
    if (cand.size() > 1) {
        // zero requests for `p` pods in the ctx's own operand scratch (the host-pointer path's; nothing else uses it while the ctx is locked)
        hipError_t e = c->pcpu.reserve(p);
        if (e == hipSuccess) e = c->pmem.reserve(p);
        if (e == hipSuccess) e = hipMemsetAsync(c->pcpu.ptr, 0, (size_t)p * 8u, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(c->pmem.ptr, 0, (size_t)p * 8u, c->stream);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        const uint32_t timing = c->opt_timing;
        const int kern = c->opt_kernel;
        c->opt_timing = 0;
        c->opt_kernel = KSCHED_KERNEL_FUSED;
        int rc = e == hipSuccess ? KSCHED_OK : KSCHED_E_HIP;
        double best_us = 0;
        constexpr int kReps = 4;
        for (int pass = 0; pass < 2 && rc == KSCHED_OK; ++pass)  // two passes: the first candidates of the first pass carry the warm-up
            for (size_t i = 0; i < cand.size() && rc == KSCHED_OK; ++i) {
                rc = eval_on_device(c, p, c->pcpu.ptr, c->pmem.ptr, nullptr, nullptr, nullptr, 0, KSCHED_FIT, (uint64_t *)cand[i].ptr, nullptr, nullptr, pitch, c->stream);
                if (rc) break;
                if (hipEventRecord(e0, c->stream) != hipSuccess) rc = KSCHED_E_HIP;
                for (int r = 0; r < kReps && rc == KSCHED_OK; ++r)
                    rc = eval_on_device(c, p, c->pcpu.ptr, c->pmem.ptr, nullptr, nullptr, nullptr, 0, KSCHED_FIT, (uint64_t *)cand[i].ptr, nullptr, nullptr, pitch, c->stream);
                if (rc == KSCHED_OK && (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = KSCHED_E_HIP;
                float ms = 0;
                if (rc == KSCHED_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = KSCHED_E_HIP;
                if (rc == KSCHED_OK && pass == 1) {
                    const double us = (double)ms * 1e3 / kReps;
                    if (i < 16) c->mask_probe_us[i] = us;
                    if (i == 0 || us < best_us) {
                        best_us = us;
                        keep = i;
                    }
                }
            }
        c->opt_timing = timing;
        c->opt_kernel = kern;
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipStreamSynchronize(c->stream);
        if (rc != KSCHED_OK) {  // the probe could not run (no bitmap index for this snapshot, ...): the first candidate as it is
            (void)hipGetLastError();
            keep = 0;
        }
    }
    release_all(keep);
    best = cand[keep];
    return KSCHED_OK;
}
}  // namespace

int ksched_mask_alloc(ksched_ctx *c, uint32_t p, uint32_t how, uint64_t **out_mask, uint32_t *out_pitch_words) try {
    if (!c || !out_mask) return KSCHED_E_INVAL;
    *out_mask = nullptr;
    switch (how) {  // (3, 6, 7, 10: paths measured in round 6 and removed)
        case KSCHED_MASK_ALLOC_AUTO: case KSCHED_MASK_ALLOC_PLAIN: case KSCHED_MASK_ALLOC_PROBE: case KSCHED_MASK_ALLOC_VMM: case KSCHED_MASK_ALLOC_VMM_MIN:
        case KSCHED_MASK_ALLOC_CONTIGUOUS: case KSCHED_MASK_ALLOC_SCATTER_2M: case KSCHED_MASK_ALLOC_SCATTER_16M: break;
        default: return KSCHED_E_INVAL;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) {
        c->last_error = "ksched_mask_alloc before ksched_set_nodes: the row pitch follows the node count";
        return KSCHED_E_STATE;
    }
    fault_point(c);
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    const uint32_t pitch = ksched_mask_pitch(c->n);
    if (out_pitch_words) *out_pitch_words = pitch;
    const size_t bytes = std::max<size_t>((size_t)p * pitch * 8u, 128u);
    uint32_t eff = how;
    if (how == KSCHED_MASK_ALLOC_AUTO) eff = (bytes >= KSCHED_MASK_PROBE_MIN_BYTES && c->idx.built && c->opt_mask_probe > 1u) ? KSCHED_MASK_ALLOC_PROBE : KSCHED_MASK_ALLOC_PLAIN;
    MaskAllocation a;
    std::fill(std::begin(c->mask_probe_us), std::end(c->mask_probe_us), 0.0);
    if (eff == KSCHED_MASK_ALLOC_PROBE) {
        const int rc = mask_alloc_probe(c, p, pitch, bytes, a);
        if (rc) return rc;
    } else {
        bool unsupported = false;
        const hipError_t e = mask_alloc_path(c, bytes, eff, a, &unsupported);
        if (unsupported) {
            c->last_error = "ksched_mask_alloc: this allocation path is a measurement path of the test build of the library (tests/cpp/hooks); the shipped library allocates with hipMalloc only";
            return KSCHED_E_UNSUPPORTED;
        }
        if (e != hipSuccess || !a.ptr) {
            c->last_error = std::string("ksched_mask_alloc: ") + hipGetErrorString(e);
            (void)hipGetLastError();
            return e == hipErrorOutOfMemory ? KSCHED_E_NOMEM : KSCHED_E_HIP;
        }
    }
    {
        MaskRegistry &reg = mask_registry();
        std::lock_guard<std::mutex> rl(reg.mu);
        reg.live.push_back(a);
    }
    *out_mask = (uint64_t *)a.ptr;
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

int ksched_mask_probe_report(ksched_ctx *c, double *out_us, uint32_t cap) try {
    if (!c || (cap && !out_us)) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    uint32_t n = 0;
    while (n < 16u && c->mask_probe_us[n] > 0.0) ++n;
    n = std::min(n, cap);
    for (uint32_t i = 0; i < n; ++i) out_us[i] = c->mask_probe_us[i];
    return (int)n;
} KSCHED_ABI_CATCH(c)

int ksched_mask_free(ksched_ctx *c, uint64_t *mask) try {
    if (!c) return KSCHED_E_INVAL;
    if (!mask) return KSCHED_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    MaskAllocation a;
    {
        MaskRegistry &reg = mask_registry();
        std::lock_guard<std::mutex> rl(reg.mu);
        auto it = std::find_if(reg.live.begin(), reg.live.end(), [&](const MaskAllocation &m) { return m.ptr == (void *)mask; });
        if (it == reg.live.end()) {
            c->last_error = "ksched_mask_free: not a pointer ksched_mask_alloc returned (or freed already)";
            return KSCHED_E_INVAL;
        }
        a = *it;
        reg.live.erase(it);
    }
    // evaluations still writing it (any stream of the device) finish first: unmapping under a running kernel is a fault
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, mask_release(a));
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

int ksched_gather_buffer(ksched_ctx *c, uint32_t count, int32_t **dev) try {
    if (!c || !dev) return KSCHED_E_INVAL;
    *dev = nullptr;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    if (count > c->gathered.cap) HIPCHK(c, hipStreamSynchronize(c->stream));  // (growing frees the old table: nothing may still be reading it)
    HIPCHK(c, c->gathered.reserve(count));
    *dev = c->gathered.ptr;
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

int ksched_eval_end(ksched_ctx *c, const int32_t *bindings_dev, uint32_t count, int32_t *out_host) try {
    if (!c) return KSCHED_E_INVAL;
    if (count > 0 && out_host && !bindings_dev) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    if (count > 0 && out_host) HIPCHK(c, hipMemcpyAsync(out_host, bindings_dev, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

int ksched_reason(const uint64_t *feasible_row, const uint64_t *fit_row, uint32_t node, uint32_t flags) try {
    if (!feasible_row) return KSCHED_E_INVAL;
    const uint32_t w = node >> 6, b = node & 63u;
    if ((feasible_row[w] >> b) & 1ull) return KSCHED_REASON_OK;
    // reference order: resources first (src/predicates.rs:68-70), then the selector (:72-74)
    if ((flags & KSCHED_FIT) && fit_row && !((fit_row[w] >> b) & 1ull)) return KSCHED_REASON_NOT_ENOUGH_RESOURCES;
    if ((flags & KSCHED_SEL) && !(flags & KSCHED_TAINT)) return KSCHED_REASON_NODE_SELECTOR_MISMATCH;
    if ((flags & KSCHED_TAINT) && !(flags & KSCHED_SEL)) return KSCHED_REASON_TAINT_NOT_TOLERATED;
    // both extension and selector active: the two masks cannot tell them apart
    return KSCHED_REASON_NODE_SELECTOR_MISMATCH;
} KSCHED_ABI_CATCH(nullptr)

int ksched_explain(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel, const uint64_t *ptol,
                   uint32_t count, const uint32_t *pair_pod, const uint32_t *pair_node, uint32_t flags, int32_t *out_reason) try {
    if (!c) return KSCHED_E_INVAL;
    if (flags & ~(KSCHED_FIT | KSCHED_SEL | KSCHED_TAINT)) return KSCHED_E_INVAL;
    if (count > 0 && (!pair_pod || !pair_node || !out_reason)) return KSCHED_E_INVAL;
    if (count > 0 && (flags & KSCHED_FIT) && (!pcpu || !pmem)) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    for (uint32_t i = 0; i < count; ++i)
        if (pair_pod[i] >= p || pair_node[i] >= c->n) return KSCHED_E_INVAL;
    if (count == 0) return KSCHED_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    fault_point(c);
    hipStream_t s = c->stream;
    if (int rce = stream_enter(c, s)) return rce;  // behind the latest snapshot change, whichever stream carried it
    const bool use_fit = flags & KSCHED_FIT;
    const bool use_sel = (flags & KSCHED_SEL) && psel && c->nkeys > 0;
    const bool use_taint = (flags & KSCHED_TAINT) && c->have_taints;
    if (use_fit) {
        HIPCHK(c, c->pcpu.reserve(p));
        HIPCHK(c, c->pmem.reserve(p));
        HIPCHK(c, hipMemcpyAsync(c->pcpu.ptr, pcpu, (size_t)p * 8, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->pmem.ptr, pmem, (size_t)p * 8, hipMemcpyHostToDevice, s));
    }
    if (use_sel) {
        HIPCHK(c, c->psel.reserve((size_t)p * c->nkeys));
        HIPCHK(c, hipMemcpyAsync(c->psel.ptr, psel, (size_t)p * c->nkeys * 4, hipMemcpyHostToDevice, s));
    }
    if (use_taint && ptol) {
        HIPCHK(c, c->ptol.reserve(p));
        HIPCHK(c, hipMemcpyAsync(c->ptol.ptr, ptol, (size_t)p * 8, hipMemcpyHostToDevice, s));
    }
    HIPCHK(c, c->xpairs.reserve((size_t)count * 2));
    HIPCHK(c, c->xreason.reserve(count));
    HIPCHK(c, hipMemcpyAsync(c->xpairs.ptr, pair_pod, (size_t)count * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->xpairs.ptr + count, pair_node, (size_t)count * 4, hipMemcpyHostToDevice, s));
    ExplainArgs q{};
    q.nrec = c->nrec.ptr;
    q.nlab = c->nlab.ptr;
    q.ntaint = use_taint ? c->ntaint.ptr : nullptr;
    q.pcpu = c->pcpu.ptr;
    q.pmem = c->pmem.ptr;
    q.psel = use_sel ? c->psel.ptr : nullptr;
    q.ptol = (use_taint && ptol) ? c->ptol.ptr : nullptr;
    q.pair_pod = c->xpairs.ptr;
    q.pair_node = c->xpairs.ptr + count;
    q.reason = c->xreason.ptr;
    q.count = count;
    q.p = p;
    q.n = c->n;
    q.nkeys = use_sel ? c->nkeys : 0u;
    q.do_fit = use_fit ? 1u : 0u;
    q.do_taint = use_taint ? 1u : 0u;
    hipLaunchKernelGGL(k_explain_pairs, dim3((count + 255u) / 256u), dim3(256), 0, s, q);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out_reason, c->xreason.ptr, (size_t)count * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

// ---- multi-GPU: RCCL all-gather of the bindings (comm_rccl.hpp) --------------------------------------------------

struct ksched_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, nranks = 1;
};

namespace {
thread_local std::string g_comm_error;

int comm_fail(const char *what, ncclResult_t r) {
    RcclApi &api = rccl_api();
    g_comm_error = std::string(what) + ": " + ((api.ok && api.GetErrorString) ? api.GetErrorString(r) : "RCCL unavailable");
    return KSCHED_E_RCCL;
}
int comm_unavailable() {
    g_comm_error = rccl_api().error;
    return KSCHED_E_RCCL;
}
int comm_caught(int code, const char *what) noexcept {
    try {
        g_comm_error = std::string("exception inside the library: ") + (what ? what : "?");
    } catch (...) {
    }
    return code;
}
#define KSCHED_ABI_CATCH_COMM                                                                 \
    catch (const std::bad_alloc &) { return comm_caught(KSCHED_E_NOMEM, "std::bad_alloc"); }  \
    catch (const std::exception &e_) { return comm_caught(KSCHED_E_INVAL, e_.what()); }        \
    catch (...) { return comm_caught(KSCHED_E_INVAL, "unknown exception"); }
}  // namespace

const char *ksched_comm_last_error(void) { return g_comm_error.c_str(); }

int ksched_comm_unique_id(uint8_t *id) try {
    if (!id) return KSCHED_E_INVAL;
    RcclApi &api = rccl_api();
    if (!api.ok) return comm_unavailable();
    static_assert(sizeof(ncclUniqueId) == KSCHED_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    ncclResult_t r = api.GetUniqueId(&u);
    if (r != ncclSuccess) return comm_fail("ncclGetUniqueId", r);
    memcpy(id, &u, sizeof u);
    return KSCHED_OK;
} KSCHED_ABI_CATCH_COMM

int ksched_comm_create(ksched_ctx *c, const uint8_t *id, int rank, int nranks, ksched_comm **out) try {
    if (!c || !id || !out || nranks <= 0 || rank < 0 || rank >= nranks) return KSCHED_E_INVAL;
    *out = nullptr;
    RcclApi &api = rccl_api();
    if (!api.ok) return comm_unavailable();
    DeviceGuard g(c->device);  // ncclCommInitRank binds the communicator to the current device
    if (!g.ok) return KSCHED_E_HIP;
    ksched_comm *q = new (std::nothrow) ksched_comm();
    if (!q) return KSCHED_E_NOMEM;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t r = api.CommInitRank(&q->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        delete q;
        return comm_fail("ncclCommInitRank", r);
    }
    q->device = c->device;
    q->rank = rank;
    q->nranks = nranks;
    *out = q;
    return KSCHED_OK;
} KSCHED_ABI_CATCH_COMM

int ksched_comm_create_local(ksched_ctx *const *ctxs, int n, ksched_comm **out) try {
    if (!ctxs || !out || n <= 0 || n > 64) return KSCHED_E_INVAL;
    for (int i = 0; i < n; ++i) {
        out[i] = nullptr;
        if (!ctxs[i]) return KSCHED_E_INVAL;
    }
    RcclApi &api = rccl_api();
    if (!api.ok) return comm_unavailable();
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
    std::vector<ncclComm_t> comms(n, nullptr);
    ncclResult_t r = api.CommInitAll(comms.data(), n, devs.data());
    if (r != ncclSuccess) return comm_fail("ncclCommInitAll", r);
    for (int i = 0; i < n; ++i) {
        ksched_comm *q = new (std::nothrow) ksched_comm();
        if (!q) {
            for (int j = 0; j < n; ++j) {
                if (j < i) delete out[j];
                out[j] = nullptr;
                (void)api.CommDestroy(comms[j]);
            }
            return KSCHED_E_NOMEM;
        }
        q->comm = comms[i];
        q->device = devs[i];
        q->rank = i;
        q->nranks = n;
        out[i] = q;
    }
    return KSCHED_OK;
} KSCHED_ABI_CATCH_COMM

void ksched_comm_destroy(ksched_comm *q) try {
    if (!q) return;
    RcclApi &api = rccl_api();
    if (api.ok && q->comm) {
        DeviceGuard g(q->device);
        (void)api.CommDestroy(q->comm);
    }
    delete q;
} catch (...) {  // nothing unwinds across the C ABI
}

int ksched_comm_rank(const ksched_comm *q) { return q ? q->rank : -1; }
int ksched_comm_size(const ksched_comm *q) { return q ? q->nranks : 0; }

int ksched_allgather_bindings(ksched_comm *q, const int32_t *local, int32_t *gathered, uint32_t count_per_rank, void *hip_stream) try {
    if (!q || !q->comm) return KSCHED_E_INVAL;
    if (count_per_rank > 0 && (!local || !gathered)) return KSCHED_E_INVAL;
    if (count_per_rank == 0) return KSCHED_OK;
    RcclApi &api = rccl_api();
    if (!api.ok) return comm_unavailable();
    DeviceGuard g(q->device);
    if (!g.ok) return KSCHED_E_HIP;
    ncclResult_t r = api.AllGather(local, gathered, count_per_rank, ncclInt32, q->comm, (hipStream_t)hip_stream);
    return r == ncclSuccess ? KSCHED_OK : comm_fail("ncclAllGather", r);
} KSCHED_ABI_CATCH_COMM

int ksched_allgather_bindings_local(ksched_comm *const *comms, int n, const int32_t *const *local, int32_t *const *gathered,
                                    uint32_t count_per_rank, void *const *hip_streams) try {
    if (!comms || n <= 0 || !local || !gathered) return KSCHED_E_INVAL;
    for (int i = 0; i < n; ++i)
        if (!comms[i] || !comms[i]->comm || (count_per_rank > 0 && (!local[i] || !gathered[i]))) return KSCHED_E_INVAL;
    if (count_per_rank == 0) return KSCHED_OK;
    RcclApi &api = rccl_api();
    if (!api.ok) return comm_unavailable();
    // one process drives every device: the per-device calls of one collective must be fused in a group
    ncclResult_t r = api.GroupStart();
    if (r != ncclSuccess) return comm_fail("ncclGroupStart", r);
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < n; ++i) {
        DeviceGuard g(comms[i]->device);
        r = api.AllGather(local[i], gathered[i], count_per_rank, ncclInt32, comms[i]->comm, hip_streams ? (hipStream_t)hip_streams[i] : nullptr);
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    r = api.GroupEnd();
    if (first != ncclSuccess || r != ncclSuccess) {
        // The collective is at best half issued: ranks that did enqueue it would wait for the others for ever, and whoever then
        // synchronises their stream hangs with them.  Give the whole clique up -- ncclCommAbort takes its outstanding work down --
        // and leave the handles empty: every later call with them fails with KSCHED_E_INVAL, ksched_comm_destroy still frees them.
        for (int i = 0; i < n; ++i) {
            DeviceGuard g(comms[i]->device);
            if (api.CommAbort) (void)api.CommAbort(comms[i]->comm);
            else (void)api.CommDestroy(comms[i]->comm);
            comms[i]->comm = nullptr;
        }
        const int rc = first != ncclSuccess ? comm_fail("ncclAllGather", first) : comm_fail("ncclGroupEnd", r);
        g_comm_error += " -- the communicator clique has been aborted; create a new one";
        return rc;
    }
    return KSCHED_OK;
} KSCHED_ABI_CATCH_COMM

int ksched_index_checksum(ksched_ctx *c, uint64_t *out) try {
    if (!c || !out) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    out[0] = out[1] = 0;
    if (!c->idx.built) return KSCHED_OK;
    DeviceGuard g(c->device);
    fault_point(c);
    const IndexedLayout &l = c->idx.lay;
    std::vector<uint64_t> tab((size_t)l.tiles * l.rows * kTileWords), aux((size_t)l.tiles * kAuxWords);
    HIPCHK(c, hipEventSynchronize(c->ev_build));  // the latest snapshot change, whichever stream carried it
    HIPCHK(c, hipMemcpy(tab.data(), c->idx.d_tables, tab.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(aux.data(), c->idx.d_aux, aux.size() * 8, hipMemcpyDeviceToHost));
    auto fnv = [](const std::vector<uint64_t> &v) {
        uint64_t h = 0xCBF29CE484222325ull;
        for (uint64_t w : v) {
            h ^= w;
            h *= 0x100000001B3ull;
            h ^= h >> 29;
        }
        return h ? h : 1ull;
    };
    std::vector<uint64_t> lists((size_t)l.tiles * l.nlist * kListBytes / 8);
    if (!lists.empty()) HIPCHK(c, hipMemcpy(lists.data(), c->idx.d_list, lists.size() * 8, hipMemcpyDeviceToHost));
    out[0] = fnv(tab);
    out[1] = fnv(aux) ^ (lists.empty() ? 0ull : fnv(lists) * 0x9E3779B97F4A7C15ull);
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

int ksched_trace_read(ksched_ctx *c, uint64_t *out, uint32_t max_blocks) try {
    if (!c || !out) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->trace.ptr) return 0;
    DeviceGuard g(c->device);
    HIPCHK(c, hipDeviceSynchronize());
    const uint32_t nb = std::min<uint32_t>(max_blocks, 8192u);
    HIPCHK(c, hipMemcpy(out, c->trace.ptr, (size_t)nb * KSCHED_TRACE_WORDS * 8, hipMemcpyDeviceToHost));
    return (int)nb;
} KSCHED_ABI_CATCH(c)

int ksched_kernel_time_ms(ksched_ctx *c, double *total_ms, uint64_t *launches) try {
    if (!c) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    double tot = 0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        HIPCHK(c, hipEventSynchronize(c->ev_pool[i].b));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = c->ev_used;
    c->ev_used = 0;
    return KSCHED_OK;
} KSCHED_ABI_CATCH(c)

int ksched_kernel_time_samples(ksched_ctx *c, double *out_ms, uint32_t cap) try {
    if (!c || (cap > 0 && !out_ms)) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const size_t n = std::min<size_t>(c->ev_used, cap);
    for (size_t i = 0; i < n; ++i) {
        HIPCHK(c, hipEventSynchronize(c->ev_pool[i].b));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b));
        out_ms[i] = ms;
    }
    c->ev_used = 0;
    return (int)n;
} KSCHED_ABI_CATCH(c)

}  // extern "C"
