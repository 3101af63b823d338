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
#include <string>
#include <vector>

#include "comm_rccl.hpp"
#include "kernels_direct.hpp"
#include "kernels_fused.hpp"
#include "tile_index.hpp"

using namespace ksched;

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
    DevBuf<int64_t> ncm;  // {avail_cpu, avail_mem} interleaved per node (k_select_sampled: one 16-byte gather per candidate)
    DevBuf<uint32_t> nlab;
    DevBuf<uint64_t> ntaint;
    DevBuf<uint32_t> bf_order, bf_rank;
    DevBuf<int64_t> bf_mem, bf_cpu;  // node columns once more, in best-fit order
    DevBuf<int64_t> cpu_sorted;      // ascending avail_cpu (rank of a cpu request)
    DevBuf<int64_t> bf_samples;      // sample arrays of bf_mem and cpu_sorted: [mem s1][mem s2][cpu s1][cpu s2]
    uint32_t bf_n1 = 0, bf_n2 = 0;
    DevBuf<uint64_t> bf_rows;        // [rows][Wbf] bitmaps over best-fit positions (k_pick_bestfit_rows); built with the tile index
    bool bf_rows_built = false;
    uint32_t bf_row_cpu0 = 0, bf_q = 1;
    std::vector<uint32_t> h_order;   // bf_order on the host (upload_bestfit_order -> upload_bestfit_rows)
    std::vector<uint32_t> h_lab;     // host images of the label and taint columns (re-permuted when the order changes)
    std::vector<uint64_t> h_taint;
    IndexedSnapshot idx;  // per-tile bitmap index (tile_index.hpp)
    std::vector<int64_t> h_cpu, h_mem;  // host image of `available` (ksched_update_nodes patches single rows)

    // scratch for the host-pointer path
    DevBuf<int64_t> pcpu, pmem;
    DevBuf<uint32_t> psel, psamples;
    DevBuf<uint64_t> ptol, feas, fit;
    DevBuf<int32_t> binding;
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
    DevBuf<uint64_t> trace;
    uint32_t trace_blocks_last = 0;
    const char *last_kernel = "none";

    // timing
    struct EvPair {
        hipEvent_t a, b;
    };
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
};

namespace {

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

// best-fit candidate order of the snapshot: ascending (avail_mem, avail_cpu, node)
int upload_bestfit_order(ksched_ctx *c) {
    const uint32_t n = c->n;
    const int64_t *cpu = c->h_cpu.data(), *mem = c->h_mem.data();
    std::vector<uint32_t> order(n), rank(n);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        if (mem[x] != mem[y]) return mem[x] < mem[y];
        if (cpu[x] != cpu[y]) return cpu[x] < cpu[y];
        return x < y;
    });
    std::vector<int64_t> bfmem(n), bfcpu(n);
    for (uint32_t i = 0; i < n; ++i) {
        rank[order[i]] = i;
        bfmem[i] = mem[order[i]];
        bfcpu[i] = cpu[order[i]];
    }
    HIPCHK(c, hipMemcpy(c->bf_mem.ptr, bfmem.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->bf_cpu.ptr, bfcpu.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->bf_order.ptr, order.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->bf_rank.ptr, rank.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    c->h_order = std::move(order);
    return KSCHED_OK;
}

// Bitmaps over best-fit positions for k_pick_bestfit_rows: the tile index's named rows (valid, taint groups, label
// values: same row numbers) with the nodes in bf_order, followed by 257 cpu threshold rows.  Rebuilt whenever the order
// changes (ksched_set_nodes, ksched_update_nodes): O(n * (keys + 16 * taint groups)) bit operations on the host.
int upload_bestfit_rows(ksched_ctx *c) {
    c->bf_rows_built = false;
    if (!c->idx.built || c->n == 0) return KSCHED_OK;
    const IndexedLayout &l = c->idx.lay;
    const uint32_t n = c->n, Wbf = (n + 63u) / 64u;
    const uint32_t named = l.row_cpu;  // rows [0, named): zero, valid, taint rows, label rows
    const uint32_t levels = 256u, q = (n + levels - 1u) / levels;
    const uint32_t rows = named + levels + 1u;
    const std::vector<uint32_t> &order = c->h_order;  // the best-fit order just made by upload_bestfit_order
    std::vector<uint64_t> R((size_t)rows * Wbf, 0ull);
    auto setbit = [&](uint32_t row, uint32_t i) { R[(size_t)row * Wbf + (i >> 6)] |= 1ull << (i & 63u); };
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t node = order[i];
        setbit(l.row_valid, i);
        for (uint32_t k = 0; k < c->nkeys; ++k) {
            const uint32_t id = c->h_lab[(size_t)k * n + node];
            if (id) setbit(l.lab_base[k] + id - 1u, i);
        }
        if (c->have_taints)
            for (uint32_t g = 0; g < l.ngroups; ++g) {
                const uint32_t tg = (uint32_t)((c->h_taint[node] >> (4u * g)) & 15ull);
                for (uint32_t sub = 0; sub < 16; ++sub)
                    if ((tg & ~sub) == 0) setbit(l.row_taint + 16u * g + sub, i);
            }
    }
    if (!c->have_taints)  // no node has a taint: every (group, subset) row is the all-valid row
        for (uint32_t r = l.row_taint; r < l.row_taint + 16u * l.ngroups; ++r)
            std::copy(R.begin() + (size_t)l.row_valid * Wbf, R.begin() + (size_t)(l.row_valid + 1) * Wbf, R.begin() + (size_t)r * Wbf);
    // cpu threshold rows: cpurank = position in ascending (cpu, node) order; row[t] = {i : cpurank >= t * q}
    std::vector<uint32_t> by_cpu(n), pos_of(n);
    std::iota(by_cpu.begin(), by_cpu.end(), 0u);
    std::stable_sort(by_cpu.begin(), by_cpu.end(), [&](uint32_t x, uint32_t y) { return c->h_cpu[x] < c->h_cpu[y]; });
    std::vector<int64_t> sorted(n);
    std::vector<uint32_t> bfpos(n);
    for (uint32_t i = 0; i < n; ++i) bfpos[order[i]] = i;
    for (uint32_t rnk = 0; rnk < n; ++rnk) {
        sorted[rnk] = c->h_cpu[by_cpu[rnk]];
        pos_of[rnk] = bfpos[by_cpu[rnk]];
    }
    // from the top down: row[levels] = {rank >= levels * q} (empty: levels * q >= n), row[t] = row[t + 1] | {ranks in [t q, (t + 1) q)}
    for (int t = (int)levels; t >= 0; --t) {
        uint64_t *row = R.data() + (size_t)(named + (uint32_t)t) * Wbf;
        if ((uint32_t)t < levels) std::copy(row + Wbf, row + 2 * (size_t)Wbf, row);
        const uint64_t lo = std::min<uint64_t>(n, (uint64_t)t * q), hi = ((uint32_t)t == levels) ? n : std::min<uint64_t>(n, (uint64_t)(t + 1) * q);
        for (uint64_t rnk = lo; rnk < hi; ++rnk) setbit(named + (uint32_t)t, pos_of[rnk]);
    }
    // sample arrays for the three-round searches: last element of every block of 64 / of 4096
    {
        const uint32_t n1 = (n + 63u) / 64u, n2 = (n + 4095u) / 4096u;
        std::vector<int64_t> smp(2 * (size_t)(n1 + n2));
        for (int which = 0; which < 2; ++which) {
            int64_t *s1 = smp.data() + (size_t)which * (n1 + n2), *s2 = s1 + n1;
            auto at = [&](uint32_t i) { return which == 0 ? c->h_mem[order[i]] : sorted[i]; };
            for (uint32_t j = 0; j < n1; ++j) s1[j] = at(std::min(n, (j + 1u) * 64u) - 1u);
            for (uint32_t j = 0; j < n2; ++j) s2[j] = at(std::min<uint64_t>(n, (uint64_t)(j + 1u) * 4096u) - 1u);
        }
        HIPCHK(c, c->bf_samples.reserve(smp.size()));
        HIPCHK(c, hipMemcpy(c->bf_samples.ptr, smp.data(), smp.size() * 8, hipMemcpyHostToDevice));
        c->bf_n1 = n1;
        c->bf_n2 = n2;
    }
    HIPCHK(c, c->bf_rows.reserve(R.size()));
    HIPCHK(c, hipMemcpy(c->bf_rows.ptr, R.data(), R.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->cpu_sorted.ptr, sorted.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    c->bf_row_cpu0 = named;
    c->bf_q = q;
    c->bf_rows_built = true;
    return KSCHED_OK;
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

SelectArgs make_select_args(const ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                            const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags, int32_t *out_binding) {
    const bool sel = (flags & KSCHED_SEL) && psel && c->nkeys > 0;
    const bool taint = (flags & KSCHED_TAINT) && c->have_taints;
    SelectArgs q{};
    q.ncm = c->ncm.ptr;
    q.nlab = c->nlab.ptr;
    q.ntaint = taint ? c->ntaint.ptr : nullptr;
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
    const dim3 grid((p + 255) / 256), block(256);
    if (attempts <= 5) hipLaunchKernelGGL(k_select_sampled<5>, grid, block, 0, s, q);
    else hipLaunchKernelGGL(k_select_sampled<8>, grid, block, 0, s, q);
    HIPCHK(c, hipGetLastError());
    return KSCHED_OK;
}

int eval_on_device(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                   const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags,
                   uint64_t *out_feas, uint64_t *out_fit, int32_t *out_binding, uint32_t pitch, hipStream_t s) {
    const bool pick_s = flags & KSCHED_PICK_SAMPLED, pick_b = flags & KSCHED_PICK_BESTFIT;
    if (p == 0) return KSCHED_OK;
    if (c->n == 0) {
        // no nodes: empty mask rows, no binding possible (reference: choose() on an empty store
        // yields None on every attempt, src/main.rs:56,70)
        if ((pick_s || pick_b) && out_binding) HIPCHK(c, hipMemsetAsync(out_binding, 0xFF, (size_t)p * sizeof(int32_t), s));
        return KSCHED_OK;
    }
    // The sampled pick tests only the drawn candidates, from the columns: it does not need the mask, so it goes
    // first (nothing waits on a mask kernel) and a bindings-only request launches no mask kernel at all.
    // KSCHED_OPT_PICK_FROM_MASK restores the mask-reading pick (same results; kept as a cross-check).
    const bool select_direct = pick_s && !c->opt_pick_from_mask;
    if (select_direct) {
        int rcs = launch_select(c, p, pcpu, pmem, psel, ptol, samples, attempts, flags, out_binding, s);
        if (rcs) return rcs;
        if (!out_feas && !out_fit) return KSCHED_OK;
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
        q.Wbf = c->W;
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
        hipLaunchKernelGGL(k_pick_bestfit_rows, dim3((p + 3) / 4), dim3(256), 0, s, q);
        HIPCHK(c, hipGetLastError());
        if (!out_feas && !out_fit) return KSCHED_OK;
    }
    uint64_t *feas = out_feas;
    if (!feas) {  // the mask kernels always write the feasible mask: a pick that reads it, or a fit-mask-only request, gets a scratch one
        HIPCHK(c, c->scratch_mask.reserve((size_t)p * pitch));
        feas = c->scratch_mask.ptr;
    }

    // kernel choice is needed before timing: the fused kernel carries its events on the dispatch
    // packet itself (hipExtLaunchKernel), the multi-launch paths are bracketed by stream events.
    int rc;
    // kernel choice: fused (one launch over the bitmap index) when the snapshot has an index that
    // fits LDS, else the always-applicable direct kernel; KSCHED_OPT_KERNEL can force one.
    const bool can_fused = fused_applicable(c->idx, flags);
    int kern = c->opt_kernel;
    if (kern == KSCHED_KERNEL_AUTO) kern = can_fused ? KSCHED_KERNEL_FUSED : KSCHED_KERNEL_DIRECT;
    if (kern == KSCHED_KERNEL_FUSED && !can_fused) {
        c->last_error = "fused kernel not applicable to this snapshot/request (bitmap index does not fit LDS)";
        return KSCHED_E_UNSUPPORTED;
    }
    size_t slot = 0;
    const bool timed = c->opt_timing && (c->timing_seq++ % c->opt_timing) == 0;
    if (timed) {
        int trc = timing_slot(c, &slot);
        if (trc) return trc;
        if (kern != KSCHED_KERNEL_FUSED) HIPCHK(c, hipEventRecord(c->ev_pool[slot].a, s));
    }
    if (kern == KSCHED_KERNEL_FUSED) {
        constexpr uint32_t kTraceBlocks = 8192;
        if (c->opt_trace) {
            DeviceGuard g2(c->device);
            HIPCHK(c, c->trace.reserve((size_t)kTraceBlocks * KSCHED_TRACE_WORDS));
            HIPCHK(c, hipMemsetAsync(c->trace.ptr, 0, (size_t)kTraceBlocks * KSCHED_TRACE_WORDS * 8, s));
        }
        hipError_t e = run_fused(c->idx, p, pcpu, pmem, psel, ptol, flags, feas, out_fit, pitch, s, c->opt_debug,
                                 timed ? c->ev_pool[slot].a : nullptr, timed ? c->ev_pool[slot].b : nullptr,
                                 c->opt_trace ? c->trace.ptr : nullptr, kTraceBlocks);
        if (e != hipSuccess) return fail_hip(c, e, "run_fused");
        c->last_kernel = "fused";
        rc = KSCHED_OK;
    } else {
        rc = run_direct(c, p, pcpu, pmem, psel, ptol, flags, feas, out_fit, pitch, s);
    }
    if (rc) return rc;
    if (timed && kern != KSCHED_KERNEL_FUSED) HIPCHK(c, hipEventRecord(c->ev_pool[slot].b, s));

    if (select_direct || bestfit_rows || !(pick_s || pick_b)) return KSCHED_OK;
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

int ksched_create(ksched_ctx **out, int device_id) {
    if (!out) return KSCHED_E_INVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return KSCHED_E_NODEVICE;
    if (device_id < 0 || device_id >= count) return KSCHED_E_INVAL;
    ksched_ctx *c = new (std::nothrow) ksched_ctx();
    if (!c) return KSCHED_E_NOMEM;
    c->device = device_id;
    DeviceGuard g(device_id);
    if (!g.ok || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return KSCHED_E_HIP;
    }
    *out = c;
    return KSCHED_OK;
}

void ksched_destroy(ksched_ctx *c) {
    if (!c) return;
    {
        DeviceGuard g(c->device);
        (void)hipDeviceSynchronize();
        c->ncpu.release(); c->nmem.release(); c->ncm.release(); c->nlab.release(); c->ntaint.release();
        c->bf_order.release(); c->bf_rank.release(); c->bf_mem.release(); c->bf_cpu.release(); c->cpu_sorted.release(); c->bf_rows.release(); c->bf_samples.release();
        c->pcpu.release(); c->pmem.release(); c->psel.release(); c->psamples.release();
        c->ptol.release(); c->feas.release(); c->fit.release(); c->binding.release(); c->xpairs.release(); c->xreason.release();
        c->scratch_mask.release(); c->trace.release();
        indexed_release(c->idx);
        for (auto &ep : c->ev_pool) {
            (void)hipEventDestroy(ep.a);
            (void)hipEventDestroy(ep.b);
        }
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

const char *ksched_last_error(const ksched_ctx *c) { return c ? c->last_error.c_str() : ""; }
uint32_t ksched_num_nodes(const ksched_ctx *c) { return (c && c->have_nodes) ? c->n : 0; }
uint32_t ksched_num_keys(const ksched_ctx *c) { return (c && c->have_nodes) ? c->nkeys : 0; }
const char *ksched_last_kernel(const ksched_ctx *c) { return c ? c->last_kernel : "none"; }

int ksched_set_option(ksched_ctx *c, int option, int64_t value) {
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

        default:
            return KSCHED_E_INVAL;
    }
}

int ksched_set_nodes(ksched_ctx *c, uint32_t n, const int64_t *cpu, const int64_t *mem, const uint32_t *lab,
                     uint32_t n_keys, const uint64_t *taints) {
    if (!c) return KSCHED_E_INVAL;
    if (n > 0 && (!cpu || !mem)) return KSCHED_E_INVAL;
    if (n_keys > KSCHED_MAX_KEYS) return KSCHED_E_INVAL;
    if (n_keys > 0 && n > 0 && !lab) return KSCHED_E_INVAL;
    if (lab)
        for (size_t i = 0; i < (size_t)n_keys * n; ++i)
            if (lab[i] == KSCHED_SEL_NEVER) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    // the previous snapshot may still be in use by enqueued work
    HIPCHK(c, hipDeviceSynchronize());
    c->have_nodes = false;
    c->n = n;
    c->nkeys = n_keys;
    c->W = ksched_mask_words(n);
    c->have_taints = taints != nullptr;
    HIPCHK(c, c->ncpu.reserve(n));
    HIPCHK(c, c->nmem.reserve(n));
    HIPCHK(c, c->ncm.reserve((size_t)n * 2));
    HIPCHK(c, c->nlab.reserve((size_t)n * n_keys));
    HIPCHK(c, c->ntaint.reserve(n));
    HIPCHK(c, c->bf_order.reserve(n));
    HIPCHK(c, c->bf_rank.reserve(n));
    HIPCHK(c, c->bf_mem.reserve(n));
    HIPCHK(c, c->bf_cpu.reserve(n));
    HIPCHK(c, c->cpu_sorted.reserve(n));
    int rc_bf = KSCHED_OK;
    if (n > 0) {
        HIPCHK(c, hipMemcpy(c->ncpu.ptr, cpu, (size_t)n * 8, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->nmem.ptr, mem, (size_t)n * 8, hipMemcpyHostToDevice));
        if (n_keys) HIPCHK(c, hipMemcpy(c->nlab.ptr, lab, (size_t)n * n_keys * 4, hipMemcpyHostToDevice));
        if (taints) HIPCHK(c, hipMemcpy(c->ntaint.ptr, taints, (size_t)n * 8, hipMemcpyHostToDevice));
        c->h_cpu.assign(cpu, cpu + n);
        c->h_mem.assign(mem, mem + n);
        if (n_keys) c->h_lab.assign(lab, lab + (size_t)n * n_keys);
        else c->h_lab.clear();
        if (taints) c->h_taint.assign(taints, taints + n);
        else c->h_taint.clear();
        {
            std::vector<int64_t> cm((size_t)n * 2);
            for (uint32_t i = 0; i < n; ++i) {
                cm[2 * (size_t)i] = cpu[i];
                cm[2 * (size_t)i + 1] = mem[i];
            }
            HIPCHK(c, hipMemcpy(c->ncm.ptr, cm.data(), cm.size() * 8, hipMemcpyHostToDevice));
        }
        rc_bf = upload_bestfit_order(c);
    } else {
        c->h_cpu.clear();
        c->h_mem.clear();
    }
    if (rc_bf) return rc_bf;
    hipError_t e = indexed_build(c->idx, n, cpu, mem, lab, n_keys, taints);
    if (e != hipSuccess) return fail_hip(c, e, "indexed_build");
    if (int rcr = upload_bestfit_rows(c)) return rcr;
    c->have_nodes = true;
    return KSCHED_OK;
}

int ksched_update_nodes(ksched_ctx *c, uint32_t count, const uint32_t *node_index, const int64_t *cpu, const int64_t *mem) {
    if (!c) return KSCHED_E_INVAL;
    if (count > 0 && (!node_index || !cpu || !mem)) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    for (uint32_t i = 0; i < count; ++i)
        if (node_index[i] >= c->n) return KSCHED_E_INVAL;
    if (count == 0) return KSCHED_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    // evaluations already enqueued read the snapshot as it was
    HIPCHK(c, hipDeviceSynchronize());
    std::vector<uint32_t> tiles;
    for (uint32_t i = 0; i < count; ++i) {  // a node listed twice takes its last values
        const uint32_t n = node_index[i];
        c->h_cpu[n] = cpu[i];
        c->h_mem[n] = mem[i];
        tiles.push_back(n / kTileNodes);
    }
    std::sort(tiles.begin(), tiles.end());
    tiles.erase(std::unique(tiles.begin(), tiles.end()), tiles.end());
    // device columns (direct kernel, best-fit pick): sparse rows one by one, else the tile ranges that changed
    if (count <= 32) {
        for (uint32_t i = 0; i < count; ++i) {
            const uint32_t n = node_index[i];
            HIPCHK(c, hipMemcpy(c->ncpu.ptr + n, &c->h_cpu[n], 8, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(c->nmem.ptr + n, &c->h_mem[n], 8, hipMemcpyHostToDevice));
            const int64_t pair[2] = {c->h_cpu[n], c->h_mem[n]};
            HIPCHK(c, hipMemcpy(c->ncm.ptr + 2 * (size_t)n, pair, 16, hipMemcpyHostToDevice));
        }
    } else {
        for (uint32_t t : tiles) {
            const size_t lo = (size_t)t * kTileNodes, len = std::min<size_t>(kTileNodes, c->n - lo);
            HIPCHK(c, hipMemcpy(c->ncpu.ptr + lo, c->h_cpu.data() + lo, len * 8, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(c->nmem.ptr + lo, c->h_mem.data() + lo, len * 8, hipMemcpyHostToDevice));
            std::vector<int64_t> cm(len * 2);
            for (size_t i = 0; i < len; ++i) {
                cm[2 * i] = c->h_cpu[lo + i];
                cm[2 * i + 1] = c->h_mem[lo + i];
            }
            HIPCHK(c, hipMemcpy(c->ncm.ptr + 2 * lo, cm.data(), cm.size() * 8, hipMemcpyHostToDevice));
        }
    }
    int rc = upload_bestfit_order(c);
    if (rc) return rc;
    if (c->idx.built)
        for (uint32_t t : tiles) {
            hipError_t e = indexed_update_tile(c->idx, t, c->h_cpu.data(), c->h_mem.data());
            if (e != hipSuccess) return fail_hip(c, e, "indexed_update_tile");
        }
    return upload_bestfit_rows(c);
}

int ksched_eval_device_pitched(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                               const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags,
                               uint64_t *out_feas, uint64_t *out_fit, int32_t *out_binding, uint32_t mask_pitch_words,
                               void *hip_stream) {
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
}

int ksched_eval_device(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                       const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags,
                       uint64_t *out_feas, uint64_t *out_fit, int32_t *out_binding, void *hip_stream) {
    return ksched_eval_device_pitched(c, p, pcpu, pmem, psel, ptol, samples, attempts, flags, out_feas, out_fit, out_binding,
                                      ksched_mask_words(ksched_num_nodes(c)), hip_stream);
}

uint32_t ksched_mask_pitch(uint32_t n_nodes) { return (ksched_mask_words(n_nodes) + 15u) & ~15u; }

struct ksched_pipe {
    ksched_ctx *ctx = nullptr;
    uint32_t depth = 0;
    hipStream_t s_mask = nullptr, s_pick = nullptr;
    std::vector<hipEvent_t> mask_done, pick_done;
};

int ksched_pipe_create(ksched_ctx *c, uint32_t depth, ksched_pipe **out) {
    if (!c || !out || depth == 0 || depth > 16) return KSCHED_E_INVAL;
    *out = nullptr;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    ksched_pipe *q = new (std::nothrow) ksched_pipe();
    if (!q) return KSCHED_E_NOMEM;
    q->ctx = c;
    q->depth = depth;
    bool ok = hipStreamCreateWithFlags(&q->s_mask, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&q->s_pick, hipStreamNonBlocking) == hipSuccess;
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
}

void ksched_pipe_destroy(ksched_pipe *q) {
    if (!q) return;
    {
        DeviceGuard g(q->ctx->device);
        if (q->s_mask) (void)hipStreamSynchronize(q->s_mask);
        if (q->s_pick) (void)hipStreamSynchronize(q->s_pick);
        for (auto e : q->mask_done) (void)hipEventDestroy(e);
        for (auto e : q->pick_done) (void)hipEventDestroy(e);
        if (q->s_mask) (void)hipStreamDestroy(q->s_mask);
        if (q->s_pick) (void)hipStreamDestroy(q->s_pick);
    }
    delete q;
}

void *ksched_pipe_stream(ksched_pipe *q, int which) { return q ? (void *)(which == 0 ? q->s_mask : q->s_pick) : nullptr; }

int ksched_pipe_submit(ksched_pipe *q, uint32_t slot, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                       const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags, uint64_t *mask,
                       uint32_t mask_pitch_words, int32_t *binding) {
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
    const bool pick_reads_mask = c->opt_pick_from_mask || ((pick & KSCHED_PICK_BESTFIT) && !c->bf_rows_built);
    if (pick_reads_mask) HIPCHK(c, hipStreamWaitEvent(q->s_mask, q->pick_done[slot], 0));  // the slot's mask may be overwritten once its pick has run
    rc = eval_on_device(c, p, pcpu, pmem, psel, ptol, nullptr, 0, flags & ~pick, mask, nullptr, nullptr, mask_pitch_words, q->s_mask);
    if (rc) return rc;
    if (pick_reads_mask) {
        HIPCHK(c, hipEventRecord(q->mask_done[slot], q->s_mask));
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
}

int ksched_pipe_wait(ksched_pipe *q, uint32_t slot, void *hip_stream) {
    if (!q || slot >= q->depth) return KSCHED_E_INVAL;
    ksched_ctx *c = q->ctx;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    if (hip_stream) HIPCHK(c, hipStreamWaitEvent((hipStream_t)hip_stream, q->pick_done[slot], 0));
    else HIPCHK(c, hipEventSynchronize(q->pick_done[slot]));
    return KSCHED_OK;
}

int ksched_pick_device(ksched_ctx *c, uint32_t p, const uint64_t *feasible, uint32_t mask_pitch_words, const int64_t *req_mem_bytes,
                       const uint32_t *samples, uint32_t attempts, uint32_t flags, int32_t *out_binding, void *hip_stream) {
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
}

int ksched_eval(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                const uint64_t *ptol, const uint32_t *samples, uint32_t attempts, uint32_t flags, uint64_t *out_feas,
                uint64_t *out_fit, int32_t *out_binding) {
    if (!c) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_nodes) return KSCHED_E_STATE;
    int rc = check_eval_args(c, p, pcpu, pmem, samples, attempts, flags, out_feas, out_fit, out_binding);
    if (rc) return rc;
    if (p == 0) return KSCHED_OK;
    DeviceGuard g(c->device);
    if (!g.ok) return KSCHED_E_HIP;
    hipStream_t s = c->stream;
    const size_t W = c->W;
    const size_t pitch = ksched_mask_pitch(c->n);
    const bool pick = flags & (KSCHED_PICK_SAMPLED | KSCHED_PICK_BESTFIT);
    const bool use_sel = (flags & KSCHED_SEL) && psel && c->nkeys > 0;
    const bool use_tol = (flags & KSCHED_TAINT) && ptol;

    HIPCHK(c, c->pcpu.reserve(p));
    HIPCHK(c, c->pmem.reserve(p));
    HIPCHK(c, hipMemcpyAsync(c->pcpu.ptr, pcpu, (size_t)p * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->pmem.ptr, pmem, (size_t)p * 8, hipMemcpyHostToDevice, s));
    if (use_sel) {
        HIPCHK(c, c->psel.reserve((size_t)p * c->nkeys));
        HIPCHK(c, hipMemcpyAsync(c->psel.ptr, psel, (size_t)p * c->nkeys * 4, hipMemcpyHostToDevice, s));
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
    int32_t *d_bind = nullptr;
    // a mask is needed when the caller wants it, or when the pick reads it (best fit; sampled only with KSCHED_OPT_PICK_FROM_MASK)
    const bool pick_reads_mask = (flags & (KSCHED_PICK_BESTFIT | KSCHED_PICK_SAMPLED)) &&
                                 (c->opt_pick_from_mask || ((flags & KSCHED_PICK_BESTFIT) && !c->bf_rows_built));
    if (out_feas || pick_reads_mask) {
        HIPCHK(c, c->feas.reserve((size_t)p * pitch));
        d_feas = c->feas.ptr;
    }
    if (out_fit) {
        HIPCHK(c, c->fit.reserve((size_t)p * pitch));
        d_fit = c->fit.ptr;
    }
    if (pick) {
        HIPCHK(c, c->binding.reserve(p));
        d_bind = c->binding.ptr;
    }
    rc = eval_on_device(c, p, c->pcpu.ptr, c->pmem.ptr, use_sel ? c->psel.ptr : nullptr, use_tol ? c->ptol.ptr : nullptr,
                        c->psamples.ptr, attempts, flags, d_feas, d_fit, d_bind, (uint32_t)pitch, s);
    if (rc) return rc;
    // device rows are line-aligned (pitch words apart); the caller's rows are packed (W words)
    if (out_feas && W)
        HIPCHK(c, hipMemcpy2DAsync(out_feas, W * 8, d_feas, pitch * 8, W * 8, p, hipMemcpyDeviceToHost, s));
    if (out_fit && W)
        HIPCHK(c, hipMemcpy2DAsync(out_fit, W * 8, d_fit, pitch * 8, W * 8, p, hipMemcpyDeviceToHost, s));
    if (pick && out_binding) HIPCHK(c, hipMemcpyAsync(out_binding, d_bind, (size_t)p * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return KSCHED_OK;
}

int ksched_reason(const uint64_t *feasible_row, const uint64_t *fit_row, uint32_t node, uint32_t flags) {
    if (!feasible_row) return KSCHED_E_INVAL;
    const uint32_t w = node >> 6, b = node & 63u;
    if ((feasible_row[w] >> b) & 1ull) return KSCHED_REASON_OK;
    // reference order: resources first (src/predicates.rs:68-70), then the selector (:72-74)
    if ((flags & KSCHED_FIT) && fit_row && !((fit_row[w] >> b) & 1ull)) return KSCHED_REASON_NOT_ENOUGH_RESOURCES;
    if ((flags & KSCHED_SEL) && !(flags & KSCHED_TAINT)) return KSCHED_REASON_NODE_SELECTOR_MISMATCH;
    if ((flags & KSCHED_TAINT) && !(flags & KSCHED_SEL)) return KSCHED_REASON_TAINT_NOT_TOLERATED;
    // both extension and selector active: the two masks cannot tell them apart
    return KSCHED_REASON_NODE_SELECTOR_MISMATCH;
}

int ksched_explain(ksched_ctx *c, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel, const uint64_t *ptol,
                   uint32_t count, const uint32_t *pair_pod, const uint32_t *pair_node, uint32_t flags, int32_t *out_reason) {
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
    hipStream_t s = c->stream;
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
    q.ncm = c->ncm.ptr;
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
}

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
}  // namespace

const char *ksched_comm_last_error(void) { return g_comm_error.c_str(); }

int ksched_comm_unique_id(uint8_t *id) {
    if (!id) return KSCHED_E_INVAL;
    RcclApi &api = rccl_api();
    if (!api.ok) return comm_unavailable();
    static_assert(sizeof(ncclUniqueId) == KSCHED_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    ncclResult_t r = api.GetUniqueId(&u);
    if (r != ncclSuccess) return comm_fail("ncclGetUniqueId", r);
    memcpy(id, &u, sizeof u);
    return KSCHED_OK;
}

int ksched_comm_create(ksched_ctx *c, const uint8_t *id, int rank, int nranks, ksched_comm **out) {
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
}

int ksched_comm_create_local(ksched_ctx *const *ctxs, int n, ksched_comm **out) {
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
}

void ksched_comm_destroy(ksched_comm *q) {
    if (!q) return;
    RcclApi &api = rccl_api();
    if (api.ok && q->comm) {
        DeviceGuard g(q->device);
        (void)api.CommDestroy(q->comm);
    }
    delete q;
}

int ksched_comm_rank(const ksched_comm *q) { return q ? q->rank : -1; }
int ksched_comm_size(const ksched_comm *q) { return q ? q->nranks : 0; }

int ksched_allgather_bindings(ksched_comm *q, const int32_t *local, int32_t *gathered, uint32_t count_per_rank, void *hip_stream) {
    if (!q || !q->comm) return KSCHED_E_INVAL;
    if (count_per_rank > 0 && (!local || !gathered)) return KSCHED_E_INVAL;
    if (count_per_rank == 0) return KSCHED_OK;
    RcclApi &api = rccl_api();
    if (!api.ok) return comm_unavailable();
    DeviceGuard g(q->device);
    if (!g.ok) return KSCHED_E_HIP;
    ncclResult_t r = api.AllGather(local, gathered, count_per_rank, ncclInt32, q->comm, (hipStream_t)hip_stream);
    return r == ncclSuccess ? KSCHED_OK : comm_fail("ncclAllGather", r);
}

int ksched_allgather_bindings_local(ksched_comm *const *comms, int n, const int32_t *const *local, int32_t *const *gathered,
                                    uint32_t count_per_rank, void *const *hip_streams) {
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
    if (first != ncclSuccess) return comm_fail("ncclAllGather", first);
    return r == ncclSuccess ? KSCHED_OK : comm_fail("ncclGroupEnd", r);
}

int ksched_trace_read(ksched_ctx *c, uint64_t *out, uint32_t max_blocks) {
    if (!c || !out) return KSCHED_E_INVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->trace.ptr) return 0;
    DeviceGuard g(c->device);
    HIPCHK(c, hipDeviceSynchronize());
    const uint32_t nb = std::min<uint32_t>(max_blocks, 8192u);
    HIPCHK(c, hipMemcpy(out, c->trace.ptr, (size_t)nb * KSCHED_TRACE_WORDS * 8, hipMemcpyDeviceToHost));
    return (int)nb;
}

int ksched_kernel_time_ms(ksched_ctx *c, double *total_ms, uint64_t *launches) {
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
}

int ksched_kernel_time_samples(ksched_ctx *c, double *out_ms, uint32_t cap) {
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
}

}  // extern "C"
