// kernels_direct.hpp -- "direct" mask kernel: lanes = nodes, the compare IS the ballot.
//
// Computes, for a batch of pods against the node snapshot, the feasibility bit of
// check_node_validity (reference src/predicates.rs:63-77):
//     fit  = req_cpu[p] <= avail_cpu[n] && req_mem[p] <= avail_mem[n]      (src/predicates.rs:42)
//     sel  = for every key k: sel[k][p] == 0 || sel[k][p] == lab[k][n]     (src/predicates.rs:45-61)
//     tnt  = (taints[n] & ~tol[p]) == 0                                    (extension E2)
// and writes pod-major uint64 mask rows.
//
// Mapping (gfx950, wave64): one wave owns CW consecutive 64-node word columns, whose node
// columns live in VGPRs for the whole kernel.  It walks pods 64 at a time; pod operands are
// wave-uniform, so they come in through the scalar cache (s_load) and each v_cmp against them
// yields the 64-node result word directly in an SGPR pair - no shuffle, no reduction.  The word
// of pod j is parked in lane j (v_writelane), and after 64 pods every lane stores CW
// consecutive words of "its" pod row.  The four waves of a block sit on adjacent columns, so a
// block emits 16 consecutive words (128 B) per pod row.
//
// Cost model (measured numbers live in profiles/HISTORY.md): per (pod, 64 nodes) the VALU issues are
// 2 x v_cmp_i64 (fit) + one v_cmp_u32 per *constrained* key (unconstrained keys are skipped by a
// scalar branch) + 3 for taints + 2 x v_writelane.  That makes this kernel VALU-bound well below
// the HBM write roofline; it is the general, always-applicable path.  The fused kernel
// (kernels_fused.hpp) is the fast path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ksched.h"
#include "kernarg.hpp"

// clang (ROCm 7.2) exposes no __builtin_amdgcn_writelane; bind the LLVM intrinsic by name.
// v_writelane_b32: lane `lane` of the result takes the wave-uniform `val`, the others keep `old`.
extern "C" __device__ uint32_t ksched_writelane_u32(uint32_t val, uint32_t lane, uint32_t old) __asm(
    "llvm.amdgcn.writelane.i32");

namespace ksched {

constexpr int kDirectCW = 4;      // word columns per wave
constexpr int kDirectWaves = 4;   // waves per block  -> 16 words = 1024 nodes per block
constexpr int kDirectKeys = 8;    // label keys handled per pass

struct DirectArgs {
    uint32_t n, p, W;
    uint32_t pitch;                // words between consecutive pod rows of the output masks (>= W)
    uint32_t key0, nkeys;          // this pass handles keys [key0, key0 + nkeys), nkeys <= kDirectKeys
    uint32_t pod_tiles_per_block;  // 64-pod tiles walked by one block
    uint32_t do_fit;               // KSCHED_FIT selected
    uint32_t accumulate;           // AND into out_feas instead of overwriting (key passes after the first)
};

// Pointers are separate __restrict__ kernel parameters (not struct members) so that the
// compiler can prove the pod columns are never clobbered by the mask stores and keeps the
// wave-uniform pod loads on the scalar unit (s_load_*), leaving the VALU to the compares.
template <bool SEL, bool TAINT, bool WANT_FIT>
__global__ __launch_bounds__(64 * kDirectWaves) void k_eval_direct(
    const int64_t *__restrict__ g_ncpu, const int64_t *__restrict__ g_nmem, const uint32_t *__restrict__ g_nlab,
    const uint64_t *__restrict__ g_ntaint, const int64_t *__restrict__ g_pcpu, const int64_t *__restrict__ g_pmem,
    const uint32_t *__restrict__ g_psel, const uint64_t *__restrict__ g_ptol, uint64_t *__restrict__ out_feas,
    uint64_t *__restrict__ out_fit, const DirectArgs a) {
    kernarg_warm<10 * 8 + sizeof(DirectArgs)>();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t w0 = (blockIdx.x * kDirectWaves + wave) * kDirectCW;  // first word column of this wave
    if (w0 >= a.W) return;

    // ---- node columns of this wave -> registers -------------------------------------------
    int64_t ncpu[kDirectCW], nmem[kDirectCW];
    uint32_t nlab[kDirectKeys][kDirectCW];
    uint64_t ntaint[kDirectCW];
    uint64_t valid[kDirectCW];
#pragma unroll
    for (int c = 0; c < kDirectCW; ++c) {
        const uint32_t node = (w0 + c) * 64u + lane;
        const bool in = node < a.n;
        valid[c] = __ballot(in);
        ncpu[c] = in ? g_ncpu[node] : 0;
        nmem[c] = in ? g_nmem[node] : 0;
        if (SEL) {
#pragma unroll
            for (int k = 0; k < kDirectKeys; ++k)
                nlab[k][c] = (in && (uint32_t)k < a.nkeys) ? g_nlab[(size_t)(a.key0 + k) * a.n + node] : 0u;
        }
        if (TAINT) ntaint[c] = (in && g_ntaint) ? g_ntaint[node] : 0ull;
    }

    const uint32_t tile0 = blockIdx.y * a.pod_tiles_per_block;
    for (uint32_t t = 0; t < a.pod_tiles_per_block; ++t) {
        const uint32_t p0 = (tile0 + t) * 64u;
        if (p0 >= a.p) break;
        const uint32_t jmax = min(64u, a.p - p0);

        uint32_t flo[kDirectCW], fhi[kDirectCW];  // feasible word of pod (p0 + lane)
        uint32_t rlo[kDirectCW], rhi[kDirectCW];  // fit-only word
#pragma unroll
        for (int c = 0; c < kDirectCW; ++c) { flo[c] = fhi[c] = rlo[c] = rhi[c] = 0u; }

        for (uint32_t j = 0; j < jmax; ++j) {
            const uint32_t p = p0 + j;  // wave-uniform -> scalar loads
            uint64_t m[kDirectCW];
#pragma unroll
            for (int c = 0; c < kDirectCW; ++c) m[c] = valid[c];
            if (a.do_fit) {
                const int64_t rc = g_pcpu[p];
                const int64_t rm = g_pmem[p];
#pragma unroll
                for (int c = 0; c < kDirectCW; ++c) m[c] &= __ballot(rc <= ncpu[c]) & __ballot(rm <= nmem[c]);
            }
            if (WANT_FIT) {
#pragma unroll
                for (int c = 0; c < kDirectCW; ++c) {
                    rlo[c] = ksched_writelane_u32((uint32_t)m[c], j, rlo[c]);
                    rhi[c] = ksched_writelane_u32((uint32_t)(m[c] >> 32), j, rhi[c]);
                }
            }
            if (SEL) {
#pragma unroll
                for (int k = 0; k < kDirectKeys; ++k) {
                    const uint32_t s = ((uint32_t)k < a.nkeys) ? g_psel[(size_t)(a.key0 + k) * a.p + p] : 0u;
                    if (s != 0u) {  // wave-uniform branch: unconstrained keys cost no VALU work
#pragma unroll
                        for (int c = 0; c < kDirectCW; ++c) m[c] &= __ballot(s == nlab[k][c]);
                    }
                }
            }
            if (TAINT) {
                const uint64_t ntol = g_ptol ? ~g_ptol[p] : ~0ull;
#pragma unroll
                for (int c = 0; c < kDirectCW; ++c) m[c] &= __ballot((ntaint[c] & ntol) == 0ull);
            }
#pragma unroll
            for (int c = 0; c < kDirectCW; ++c) {
                flo[c] = ksched_writelane_u32((uint32_t)m[c], j, flo[c]);
                fhi[c] = ksched_writelane_u32((uint32_t)(m[c] >> 32), j, fhi[c]);
            }
        }

        // ---- lane l stores CW consecutive words of pod row p0 + l --------------------------
        if (lane < jmax) {
            const size_t row = (size_t)(p0 + lane) * a.pitch;
#pragma unroll
            for (int c = 0; c < kDirectCW; ++c) {
                if (w0 + c < a.W) {
                    const uint64_t f = ((uint64_t)fhi[c] << 32) | flo[c];
                    if (out_feas) {
                        if (a.accumulate)
                            out_feas[row + w0 + c] &= f;
                        else
                            out_feas[row + w0 + c] = f;
                    }
                    if (WANT_FIT && out_fit) out_fit[row + w0 + c] = ((uint64_t)rhi[c] << 32) | rlo[c];
                }
            }
        }
    }
}

// ---- picks ----------------------------------------------------------------------------------

// select_node_for_pod (reference src/main.rs:51-71) with injected draws: the first of the
// `attempts` sampled node indices whose feasible bit is set wins; none -> -1 (NoNodeFound,
// src/main.rs:117).  One lane per pod.
__global__ __launch_bounds__(256) void k_pick_sampled(const uint64_t *__restrict__ mask, const uint32_t *__restrict__ samples,
                                                       int32_t *__restrict__ binding, uint32_t p, uint32_t n, uint32_t pitch,
                                                       uint32_t attempts) {
    const uint32_t pod = blockIdx.x * blockDim.x + threadIdx.x;
    if (pod >= p) return;
    int32_t b = -1;
    const uint64_t *row = mask + (size_t)pod * pitch;
    const uint32_t *smp = samples + (size_t)pod * attempts;
    // eight draws at a time: all sample indices, then all mask words, are in flight together (two
    // dependent memory round trips per eight attempts instead of two per attempt)
    for (uint32_t i0 = 0; i0 < attempts && b < 0; i0 += 8u) {
        uint32_t s[8];
        uint64_t w[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) s[j] = (i0 + j < attempts) ? smp[i0 + j] : 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) w[j] = (s[j] < n) ? row[s[j] >> 6] : 0ull;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j)
            if (b < 0 && ((w[j] >> (s[j] & 63u)) & 1ull)) b = (int32_t)s[j];  // first feasible draw wins (src/main.rs:61-65)
    }
    binding[pod] = b;
}

// select_node_for_pod evaluated the reference's own way (src/main.rs:53-66): draw a candidate, run
// check_node_validity(pod, candidate) on THAT pair, first Ok wins -- straight from the pod and node columns,
// no mask involved, so this kernel neither waits for the mask kernel nor reads its 128-byte lines back
// (the mask-reading pick above moves 500 k random sectors at C3; this one gathers from node columns that
// stay in L2).  Same predicates, same arithmetic as k_eval_direct: req <= avail on signed i64
// (src/predicates.rs:42), label id equality with 0 = absent / unconstrained (src/predicates.rs:45-61),
// (taints & ~tolerations) == 0.  One lane per pod; ATT = attempts handled per pass (all candidates of a
// pass are loaded together: two dependent memory round trips per pass).  (Running this inside the fused
// mask kernel's staging wait was tried: its ~50 registers per lane spill there and its scattered gathers
// compete with the staging DMA for the texture-address unit -- the mask kernel grew by 10 us.)
// Node records for the picks that test single candidates: everything check_node_validity reads about ONE node in ONE 64-byte
// line -- a drawn candidate costs one cache-line request instead of one per column it touches (it was 1 + one per constrained key
// + 1 with taints: up to ten scattered lines per candidate; profiles/r01_h5_sq_counters_C3.json showed the kernel waiting on them
// for half of its cycles).  16-byte units: [0] {avail_cpu, avail_mem}  [1] {taints, 0}  [2] label ids of keys 0..3  [3] of keys 4..7.
// Keys beyond the eighth stay in the label columns (read only when a pod constrains them).
constexpr uint32_t kNodeRecWords = 8;  // int64 words per record

struct SelectArgs {
    const int64_t *nrec;          // [n][kNodeRecWords] node records
    const uint32_t *nlab;         // [nkeys][n] (keys >= 8 only)
    const int64_t *pcpu, *pmem;   // pod columns [p]
    const uint32_t *psel;         // [nkeys][p] or nullptr
    const uint64_t *ptol;         // [p] or nullptr
    const uint32_t *samples;      // [p][attempts]
    int32_t *binding;             // [p]
    uint32_t p, n, nkeys, attempts, do_fit, do_taint;
};

// ATT = draws handled per pass; the first EAGER of them are fetched and tested at once, the rest only by the lanes that are
// still undecided (src/main.rs:53-66 stops at the first Ok too).  Shipped: EAGER = 1 (fewest cache-line requests; measured best
// inside the step, where the mask kernel has just swept the caches).
template <int ATT, int EAGER>
__device__ __forceinline__ int32_t select_one_pod(const SelectArgs &a, uint32_t pod) {
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define KSCHED_G(T, P) ((const __attribute__((address_space(1))) T *)(P))
    const __attribute__((address_space(1))) i64x2 *rec2 = KSCHED_G(i64x2, a.nrec);
    const __attribute__((address_space(1))) u32x4 *rec4 = KSCHED_G(u32x4, a.nrec);
    const __attribute__((address_space(1))) uint32_t *nlab = KSCHED_G(uint32_t, a.nlab), *psel = KSCHED_G(uint32_t, a.psel);
    const __attribute__((address_space(1))) uint32_t *smp = KSCHED_G(uint32_t, a.samples) + (size_t)pod * a.attempts;
    const int64_t rc = a.do_fit ? KSCHED_G(int64_t, a.pcpu)[pod] : 0, rm = a.do_fit ? KSCHED_G(int64_t, a.pmem)[pod] : 0;
    const uint64_t tol = (a.do_taint && a.ptol) ? KSCHED_G(uint64_t, a.ptol)[pod] : 0ull;
#undef KSCHED_G
    uint32_t sel[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) sel[k] = (a.psel && k < a.nkeys) ? psel[(size_t)k * a.p + pod] : 0u;
    const bool want_lo = (sel[0] | sel[1] | sel[2] | sel[3]) != 0u, want_hi = (sel[4] | sel[5] | sel[6] | sel[7]) != 0u;
    bool want_more = false;  // a constraint on a key beyond the eighth
    if (a.psel)
        for (uint32_t k = 8; k < a.nkeys; ++k) want_more |= psel[(size_t)k * a.p + pod] != 0u;

    // check_node_validity(pod, node) (src/predicates.rs:63-77) from the node's record; units the pod does not need are not fetched
    struct Cand {
        i64x2 cm, tt;
        u32x4 lo, hi;
    };
    auto fetch = [&](uint32_t node, Cand &c) {
        if (a.do_fit) c.cm = rec2[(size_t)node * 4u];
        if (a.do_taint) c.tt = rec2[(size_t)node * 4u + 1u];
        if (want_lo) c.lo = rec4[(size_t)node * 4u + 2u];
        if (want_hi) c.hi = rec4[(size_t)node * 4u + 3u];
    };
    auto passes = [&](uint32_t node, const Cand &c) -> bool {
        bool f = true;
        if (a.do_fit) f = rc <= c.cm.x && rm <= c.cm.y;                      // src/predicates.rs:42
        if (a.do_taint) f = f && ((uint64_t)c.tt.x & ~tol) == 0ull;
        if (want_lo) f = f && (sel[0] == 0u || sel[0] == c.lo.x) && (sel[1] == 0u || sel[1] == c.lo.y) && (sel[2] == 0u || sel[2] == c.lo.z) &&
                         (sel[3] == 0u || sel[3] == c.lo.w);                  // src/predicates.rs:48-57
        if (want_hi) f = f && (sel[4] == 0u || sel[4] == c.hi.x) && (sel[5] == 0u || sel[5] == c.hi.y) && (sel[6] == 0u || sel[6] == c.hi.z) &&
                         (sel[7] == 0u || sel[7] == c.hi.w);
        if (f && want_more)
            for (uint32_t k = 8; k < a.nkeys; ++k) {
                const uint32_t want = psel[(size_t)k * a.p + pod];
                if (want != 0u && want != nlab[(size_t)k * a.n + node]) f = false;
            }
        return f;
    };
    int32_t b = -1;
    for (uint32_t i0 = 0; i0 < a.attempts && b < 0; i0 += ATT) {
        uint32_t s[ATT];
#pragma unroll
        for (int j = 0; j < ATT; ++j) s[j] = (i0 + j < a.attempts) ? smp[i0 + j] : 0xFFFFFFFFu;
        Cand c[ATT];
#pragma unroll
        for (int j = 0; j < EAGER; ++j) fetch(s[j] < a.n ? s[j] : 0u, c[j]);  // an index >= n is an infeasible draw (include/ksched.h)
#pragma unroll
        for (int j = 0; j < EAGER; ++j)
            if (b < 0 && s[j] < a.n && passes(s[j], c[j])) b = (int32_t)s[j];  // first feasible draw wins (src/main.rs:61-65)
        if (b < 0) {  // the undecided lanes only: their remaining draws of the pass, fetched together
#pragma unroll
            for (int j = EAGER; j < ATT; ++j) fetch(s[j] < a.n ? s[j] : 0u, c[j]);
#pragma unroll
            for (int j = EAGER; j < ATT; ++j)
                if (b < 0 && s[j] < a.n && passes(s[j], c[j])) b = (int32_t)s[j];
        }
    }
    return b;
}

template <int ATT, int EAGER>
__global__ __launch_bounds__(256) void k_select_sampled(const SelectArgs q) {
    kernarg_warm<sizeof(SelectArgs)>();
    const uint32_t pod = blockIdx.x * blockDim.x + threadIdx.x;
    if (pod < q.p) q.binding[pod] = select_one_pod<ATT, EAGER>(q, pod);
}

// check_node_validity for a list of (pod, node) pairs, with the REASON: what the reference logs at WARN for every rejected
// candidate (src/main.rs:62 prints the InvalidNodeReason of src/predicates.rs:63-77).  Order: resources first
// (src/predicates.rs:68-70), then the selector (:72-74), then the taint extension (E2).  One lane per pair, straight from
// the columns (same compares as k_select_sampled).  Unlike two masks, this tells selector and taint failures apart.
struct ExplainArgs {
    const int64_t *nrec;          // [n][kNodeRecWords] node records (cpu, mem in words 0, 1)
    const uint32_t *nlab;         // [nkeys][n]
    const uint64_t *ntaint;       // [n] or nullptr
    const int64_t *pcpu, *pmem;   // [p]
    const uint32_t *psel;         // [nkeys][p] or nullptr
    const uint64_t *ptol;         // [p] or nullptr
    const uint32_t *pair_pod, *pair_node;  // [count]
    int32_t *reason;              // [count]: KSCHED_REASON_*
    uint32_t count, p, n, nkeys, do_fit, do_taint;
};

__global__ __launch_bounds__(256) void k_explain_pairs(const ExplainArgs q) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q.count) return;
    const uint32_t pod = q.pair_pod[i], node = q.pair_node[i];
    int32_t r = KSCHED_REASON_OK;
    if (q.do_fit && !(q.pcpu[pod] <= q.nrec[(size_t)kNodeRecWords * node] && q.pmem[pod] <= q.nrec[(size_t)kNodeRecWords * node + 1])) {
        r = KSCHED_REASON_NOT_ENOUGH_RESOURCES;  // src/predicates.rs:42,68-70
    } else {
        if (q.psel)
            for (uint32_t k = 0; k < q.nkeys; ++k) {  // src/predicates.rs:48-57
                const uint32_t want = q.psel[(size_t)k * q.p + pod];
                if (want != 0u && want != q.nlab[(size_t)k * q.n + node]) r = KSCHED_REASON_NODE_SELECTOR_MISMATCH;
            }
        if (r == KSCHED_REASON_OK && q.do_taint && q.ntaint && (q.ntaint[node] & ~(q.ptol ? q.ptol[pod] : 0ull)) != 0ull)
            r = KSCHED_REASON_TAINT_NOT_TOLERATED;
    }
    q.reason[i] = r;
}

// Best fit (extension E1): argmin over feasible nodes of (avail_mem - req_mem, avail_cpu - req_cpu,
// node index), lexicographic.  Both residuals are pod-independent shifts of the node's own
// (avail_mem, avail_cpu), so the order of candidates is a property of the snapshot: bf_order
// lists nodes in that order, bf_rank is its inverse, bf_mem[i] = avail_mem[bf_order[i]] (sorted),
// and the pick is the first feasible node in bf_order.  One wave per pod:
//   1. with KSCHED_FIT, every candidate before the first i with bf_mem[i] >= req_mem fails on
//      memory, so a 64-ary search over bf_mem (3 rounds for N <= 262144) finds where to start;
//   2. probe candidates 64 at a time (gather their mask bits, ballot, first set lane wins);
//   3. after kBestfitProbe fruitless probes fall back to a coalesced scan of the pod's row,
//      taking the minimum bf_rank over its set bits (cost ~ number of feasible nodes, which is
//      small for exactly the pods that get here).
constexpr int kBestfitProbe = 8;
__global__ __launch_bounds__(256) void k_pick_bestfit(const uint64_t *__restrict__ mask, const uint32_t *__restrict__ bf_order,
                                                       const uint32_t *__restrict__ bf_rank, const int64_t *__restrict__ bf_mem,
                                                       const int64_t *__restrict__ req_mem, int32_t *__restrict__ binding,
                                                       uint32_t p, uint32_t n, uint32_t W, uint32_t pitch, uint32_t do_fit) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t pod = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pod >= p) return;
    const uint64_t *row = mask + (size_t)pod * pitch;
    uint32_t start = 0;
    if (do_fit) {
        const int64_t req = req_mem[pod];
        uint32_t lo = 0, len = n;  // the answer (first i with bf_mem[i] >= req, or lo + len) lies in [lo, lo + len]
        while (len > 0) {
            const uint32_t step = (len + 63u) / 64u;
            const uint32_t end = lo + len;
            const uint32_t seg = lo + lane * step;  // lane's segment [seg, min(seg + step, end))
            bool less = false;
            if (seg < end) less = bf_mem[min(seg + step, end) - 1u] < req;  // last element of the segment
            const uint32_t nseg = (len + step - 1u) / step;
            const uint32_t c = (uint32_t)__popcll(__ballot(less));  // segments entirely below req (a prefix)
            lo += c * step;
            len = (c >= nseg) ? 0u : (min(lo + step, end) - lo - 1u);
            if (c >= nseg) lo = end;
        }
        start = lo;
    }
    for (int it = 0; it < kBestfitProbe; ++it) {
        const uint32_t idx = start + (uint32_t)it * 64u + lane;
        uint32_t node = 0;
        bool bit = false;
        if (idx < n) {
            node = bf_order[idx];
            bit = (row[node >> 6] >> (node & 63u)) & 1ull;
        }
        const uint64_t b = __ballot(bit);
        if (b) {
            const int first = __builtin_ctzll(b);
            const uint32_t win = __shfl(node, first, 64);
            if (lane == 0) binding[pod] = (int32_t)win;
            return;
        }
        if (start + (uint32_t)(it + 1) * 64u >= n) {
            if (lane == 0) binding[pod] = -1;
            return;
        }
    }
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t w = lane; w < W; w += 64u) {
        uint64_t word = row[w];
        while (word) {
            const uint32_t node = w * 64u + (uint32_t)__builtin_ctzll(word);
            best = min(best, bf_rank[node]);
            word &= word - 1ull;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off, 64));
    if (lane == 0) binding[pod] = (best == 0xFFFFFFFFu) ? -1 : (int32_t)bf_order[best];
}

// Best fit from bitmaps in best-fit order (the fused kernel's idea applied to the pick): the node bitmaps a pod ANDs --
// one row per (label key, value), one per (taint group, tolerated subset), the all-valid row -- are kept a second time
// with the nodes in bf_order, as whole rows of Wbf = ceil(n / 64) words.  The pod's candidates at or after `start` (the
// first position whose memory can hold the pod) that pass selector and taints are then the set bits of the AND of a few
// rows, 64 words = 4096 candidates per wave round with coalesced loads.  The cpu test uses 257 threshold rows in the
// same order, row[t] = {i : cpurank[i] >= t * q} (cpurank = position in ascending cpu order, q = ceil(n / 256)): with
// r = #nodes whose cpu is below the request, row[ceil(r / q)] holds only nodes that fit and row[floor(r / q)] all that
// do, so only candidates in the difference (at most q nodes of the whole snapshot) are tested individually.  The first
// set bit of the lowest word wins.  No mask is read: like the sampled pick, this runs before and independently of
// the mask kernel, and a bindings-only request launches no mask kernel.
// The first stage appends the pods it hands over to kBfSublists lists, wave w to list w % kBfSublists: ONE list with one counter made
// 2 000 waves queue for a returning atomic on one address (~8 ns each, up to 10 us of waiting per wave at the C5 shard).
constexpr uint32_t kBfSublists = 128;
struct BestfitRowsArgs {
    const uint64_t *rows;          // [rows][Wbf] bitmaps over bf_order positions; Wbf = ceil(n / 64) rounded up to a multiple of 8 (rows are whole 64-byte lines)
    const uint32_t *lab_meta;      // lab_base[32], lab_max[32] (row numbers shared with the tile index)
    const int64_t *cpu_sorted;     // [n] ascending avail_cpu
    const int64_t *bf_mem, *bf_cpu;  // node columns in best-fit order
    const int64_t *mem_s1, *mem_s2, *cpu_s1, *cpu_s2;  // sample arrays of bf_mem / cpu_sorted (wave_lower_bound_sampled), or nullptr
    const uint32_t *bf_order;
    const int64_t *pcpu, *pmem;
    const uint32_t *psel;          // [nkeys][p] or nullptr
    const uint64_t *ptol;          // [p] or nullptr
    int32_t *binding;
    uint32_t p, n, Wbf, nkeys, ngroups, row_valid, row_zero, row_taint, row_cpu0, q, do_fit, do_taint;
    uint32_t lab_base8[8], lab_max8[8];  // the first eight keys' row numbers as arguments (no dependent load)
    // second stage of the two-stage pick (k_pick_bestfit_lanes): the handed-over pods' 64-byte records sit in kBfSublists sub-lists of
    // `sub_cap` slots each, sub_count[32 * c] = entries of sub-list c; block b (one wave) takes entry b / kBfSublists of sub-list b % kBfSublists
    const uint32_t *sub_count;
    const uint32_t *pod_recs;
    uint32_t sub_cap;
    // 8-ary level arrays of bf_mem / cpu_sorted for the lane-per-pod searches: level k (1..nlev) holds the last element of every
    // block of 8^k entries; [mem level 1][mem level 2]...[cpu level 1]...; lvl_off[k - 1] = offset of level k inside one half
    const int64_t *lvl;
    uint32_t nlev, lvl_half, lvl_off[6];
    uint32_t *handover_recs;   // [kBfSublists][sub_cap][16]: first stage -> second stage
    uint32_t *handover_count;  // [kBfSublists][32] (one counter per 128-byte line), this call's
    uint32_t *zero_next;       // the counters of the NEXT two-stage call (three sets in rotation): zeroed here, so that no call pays a memset launch
    uint32_t lane_blocks;  // 64-byte blocks of candidate words (8 x 64 positions each) a lane looks at before handing the pod over
    // snapshots with list keys (tile_index.hpp): pods that constrain one are collected for k_pick_bestfit_listed
    uint32_t nlist, list_col[2];
    uint64_t *listed_mask;     // [ceil(p / 64)]
    // KSCHED_OPT_DEBUG bit 20 (tools/bestfit_ab.py --trace): 100 MHz time stamps, 8 words per wave of the first stage, then 4 per wave of the second
    uint64_t *trace, *trace2;
};

// first i in [0, n) with arr[i] >= key (n if none), by the whole wave: 64-ary search, three rounds for n <= 262144
__device__ __forceinline__ uint32_t wave_lower_bound(const int64_t *__restrict__ arr, uint32_t n, int64_t key, uint32_t lane) {
    uint32_t lo = 0, len = n;  // the answer lies in [lo, lo + len]
    while (len > 0) {
        const uint32_t step = (len + 63u) / 64u;
        const uint32_t end = lo + len;
        const uint32_t seg = lo + lane * step;  // lane's segment [seg, min(seg + step, end))
        bool less = false;
        if (seg < end) less = arr[min(seg + step, end) - 1u] < key;  // last element of the segment
        const uint32_t nseg = (len + step - 1u) / step;
        const uint32_t c = (uint32_t)__popcll(__ballot(less));  // segments entirely below key (a prefix)
        lo += c * step;
        len = (c >= nseg) ? 0u : (min(lo + step, end) - lo - 1u);
        if (c >= nseg) lo = end;
    }
    return lo;
}

// The same lower bound with two sample arrays: s1[j] = last element of block j of 64, s2[j] = last element of block j of
// 4096.  A plain 64-ary step reads 64 far-apart cache lines to use 8 bytes of each (125 k pods x 2 searches x ~120 lines
// was most of the best-fit pick's time at the C5 shard); with the samples every round reads 64 CONSECUTIVE entries:
// three rounds, ~9 lines per search.  n <= 64^3.
__device__ __forceinline__ uint32_t wave_lower_bound_sampled(const int64_t *__restrict__ arr, const int64_t *__restrict__ s1,
                                                             const int64_t *__restrict__ s2, uint32_t n, int64_t key, uint32_t lane) {
    const uint32_t n1 = (n + 63u) / 64u, n2 = (n + 4095u) / 4096u;  // entries of s1, s2
    // blocks of 4096 entirely below the key (a prefix, since arr is sorted)
    const uint32_t cA = (uint32_t)__popcll(__ballot(lane < n2 && s2[lane] < key));
    if (cA >= n2) return n;
    // within that block: its (up to) 64 blocks of 64
    const uint32_t j1 = cA * 64u + lane;
    const uint32_t cB = (uint32_t)__popcll(__ballot(j1 < n1 && s1[j1] < key));
    const uint32_t b1 = cA * 64u + cB;  // < n1: block cA is not entirely below the key
    const uint32_t i = b1 * 64u + lane;
    const uint32_t cC = (uint32_t)__popcll(__ballot(i < n && arr[i] < key));
    return b1 * 64u + cC;
}

// two lower bounds at once (the rounds of the two searches are independent of each other: issue their loads together)
__device__ __forceinline__ void wave_lower_bound_sampled2(const int64_t *__restrict__ a0, const int64_t *__restrict__ a0s1,
                                                          const int64_t *__restrict__ a0s2, int64_t key0, const int64_t *__restrict__ a1,
                                                          const int64_t *__restrict__ a1s1, const int64_t *__restrict__ a1s2, int64_t key1,
                                                          uint32_t n, uint32_t lane, int64_t v0, int64_t v1, uint32_t &out0, uint32_t &out1) {
    // v0, v1: this lane's entries of the two top-level sample arrays (a0s2[lane], a1s2[lane]); they do not depend on the keys,
    // so the caller loads them together with the keys (one dependent round trip less)
    const uint32_t n1 = (n + 63u) / 64u, n2 = (n + 4095u) / 4096u;
    (void)a0s2;
    (void)a1s2;
    const uint32_t cA0 = (uint32_t)__popcll(__ballot(lane < n2 && v0 < key0)), cA1 = (uint32_t)__popcll(__ballot(lane < n2 && v1 < key1));
    // a search that has run off the end keeps reading block 0 (harmless) and is fixed up at the end
    const uint32_t bA0 = (cA0 >= n2) ? 0u : cA0, bA1 = (cA1 >= n2) ? 0u : cA1;
    const uint32_t j0 = bA0 * 64u + lane, j1 = bA1 * 64u + lane;
    const int64_t w0 = (j0 < n1) ? a0s1[j0] : 0, w1 = (j1 < n1) ? a1s1[j1] : 0;
    const uint32_t cB0 = (uint32_t)__popcll(__ballot(j0 < n1 && w0 < key0)), cB1 = (uint32_t)__popcll(__ballot(j1 < n1 && w1 < key1));
    const uint32_t b0 = bA0 * 64u + cB0, b1 = bA1 * 64u + cB1;
    const uint32_t i0 = b0 * 64u + lane, i1 = b1 * 64u + lane;
    const int64_t x0 = (i0 < n) ? a0[i0] : 0, x1 = (i1 < n) ? a1[i1] : 0;
    const uint32_t cC0 = (uint32_t)__popcll(__ballot(i0 < n && x0 < key0)), cC1 = (uint32_t)__popcll(__ballot(i1 < n && x1 < key1));
    out0 = (cA0 >= n2) ? n : b0 * 64u + cC0;
    out1 = (cA1 >= n2) ? n : b1 * 64u + cC1;
}

// One round of the scan below: G groups of 64 words from word `wb` on (lane = word within the group), every row load of the round in
// flight together.  Returns true when the pod is decided, `out` = the chosen node (every lane holds the same value).
template <int G>
__device__ __forceinline__ bool bestfit_rows_round(const BestfitRowsArgs &q, uint32_t pod, uint32_t lane, uint32_t start, uint32_t wb, uint32_t r_hi,
                                                   uint32_t r_lo, const uint32_t (&lrow)[8], const uint32_t (&sel)[8], uint64_t tol, int64_t req_c,
                                                   int32_t &out) {
    // LOADS FIRST, ANDs AFTER: every row word of the round is requested before anything waits.  (Written as `if (constrained) base &=
    // row[..]` the compiler waits for each row inside its own branch: the round became up to a dozen dependent round trips instead of one --
    // found in the disassembly, `s_waitcnt vmcnt(0)` behind every single load.)
    uint64_t base[G], hi[G], lo[G], xl[G][8], xt[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint32_t w = wb + (uint32_t)g * 64u + lane;
        const uint32_t wc = (w < q.Wbf) ? w : 0u;
        // (every row holds bits of live positions only, so with the cpu rows in the AND the all-valid row adds nothing: one load less)
        base[g] = q.do_fit ? ~0ull : q.rows[(size_t)q.row_valid * q.Wbf + wc];
        hi[g] = q.do_fit ? q.rows[(size_t)r_hi * q.Wbf + wc] : 0ull;
        lo[g] = q.do_fit ? q.rows[(size_t)r_lo * q.Wbf + wc] : 0ull;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            xl[g][k] = ~0ull;
            if (sel[k] != 0u) xl[g][k] = q.rows[(size_t)lrow[k] * q.Wbf + wc];
        }
        // (the four taint rows UNCONDITIONALLY when the predicate is on -- a group the snapshot does not have reads the all-valid row --: behind a
        // branch per group the compiler waited for each of these loads before issuing the next, `tools/audit_asm.py` counts the loads in flight)
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) xt[g][t] = ~0ull;
        if (q.do_taint) {
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) {
                const uint32_t row = (t < q.ngroups) ? q.row_taint + 16u * t + (uint32_t)((tol >> (4u * t)) & 15ull) : q.row_valid;
                xt[g][t] = q.rows[(size_t)row * q.Wbf + wc];
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint32_t w = wb + (uint32_t)g * 64u + lane;
        const bool in = w < q.Wbf;
        const uint32_t wc = in ? w : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) base[g] &= xl[g][k];
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) base[g] &= xt[g][t];
        for (uint32_t k = 8; k < q.nkeys; ++k) {  // (rare: more than eight label keys, more than sixteen taints)
            const uint32_t s = q.psel[(size_t)k * q.p + pod];
            if (s != 0u) base[g] &= q.rows[(size_t)((s <= q.lab_meta[32u + k]) ? q.lab_meta[k] + s - 1u : q.row_zero) * q.Wbf + wc];
        }
        if (q.do_taint)
            for (uint32_t t = 4; t < q.ngroups; ++t)
                base[g] &= q.rows[(size_t)(q.row_taint + 16u * t + (uint32_t)((tol >> (4u * t)) & 15ull)) * q.Wbf + wc];
        if (!in) base[g] = 0;
        if (w == (start >> 6)) base[g] &= ~0ull << (start & 63u);  // positions before `start` cannot hold the pod's memory
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {  // groups hold ascending words: the first group with a hit decides
        const uint32_t w = wb + (uint32_t)g * 64u + lane;
        const uint64_t sure = q.do_fit ? (base[g] & hi[g]) : base[g];
        const uint64_t maybe = q.do_fit ? (base[g] & lo[g] & ~hi[g]) : 0ull;
        // this lane's first feasible position (rarely more than one trip: `maybe` holds < 1/256 of the nodes)
        uint32_t found = 0xFFFFFFFFu;
        uint64_t cand = sure | maybe;
        while (cand) {
            const uint32_t b = (uint32_t)__builtin_ctzll(cand);
            const uint32_t i = w * 64u + b;
            if (((sure >> b) & 1ull) || req_c <= q.bf_cpu[i]) {
                found = i;
                break;
            }
            cand &= cand - 1ull;
        }
        const uint64_t hit = __ballot(found != 0xFFFFFFFFu);
        if (hit) {  // lanes hold ascending words: the lowest lane with a hit holds the best-fit node
            const uint32_t first = (uint32_t)__shfl((int)found, __builtin_ctzll(hit), 64);
            out = (int32_t)q.bf_order[first];
            return true;
        }
    }
    return false;
}

// The scan of one pod by the whole wave, given start = first position whose memory can hold the pod, r = #nodes with cpu below the
// request, and the first word to look at (w_first >= start >> 6; the caller vouches that no feasible position lies before it).
// 64 words per round, one dependent round trip each (twelve for a pod no node can hold at 50 k nodes).  Wider rounds (128 / 256 words)
// were measured and lose: their registers (112 for 256 words) halve the waves a CU holds, and this stage is bound by wave slots.
// Returns the chosen node (every lane holds the same value).
__device__ __forceinline__ int32_t bestfit_rows_scan(const BestfitRowsArgs &q, uint32_t pod, uint32_t lane, uint32_t start, uint32_t r, uint32_t w_first,
                                                     int64_t req_c, uint64_t tol, const uint32_t (&sel)[8]) {
    uint32_t r_hi = q.row_valid, r_lo = q.row_valid;
    if (q.do_fit) {
        r_hi = q.row_cpu0 + (r + q.q - 1u) / q.q;  // only nodes that fit
        r_lo = q.row_cpu0 + r / q.q;               // every node that fits
    }
    if (start >= q.n || w_first >= q.Wbf) return -1;
    // the rows this pod ANDs (wave-uniform): selector keys, taint groups; row numbers of the first eight keys from the arguments
    uint32_t lrow[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) lrow[k] = (sel[k] <= q.lab_max8[k]) ? q.lab_base8[k] + sel[k] - 1u : q.row_zero;
    int32_t out = -1;
    uint32_t wb = w_first;
    for (; wb < q.Wbf; wb += 64u)
        if (bestfit_rows_round<1>(q, pod, lane, start, wb, r_hi, r_lo, lrow, sel, tol, req_c, out)) return out;
    return -1;
}

// one pod, by the whole wave: the two rank searches, then the scan
__device__ __forceinline__ int32_t bestfit_rows_one_pod(const BestfitRowsArgs &q, uint32_t pod, uint32_t lane) {
    // Everything that depends only on the pod is loaded up front, together (the chain of dependent memory round trips
    // is what this kernel's time is made of: operands -> two more search rounds -> rows -> winner's node id).
    const uint32_t n2 = (q.n + 4095u) / 4096u;
    const bool sampled = q.do_fit && q.mem_s1 != nullptr;
    const int64_t top_m = (sampled && lane < n2) ? q.mem_s2[lane] : 0, top_c = (sampled && lane < n2) ? q.cpu_s2[lane] : 0;
    const int64_t req_c = q.do_fit ? q.pcpu[pod] : 0, req_m = q.do_fit ? q.pmem[pod] : 0;
    uint32_t start = 0, r = 0;
    if (q.do_fit) {
        if (q.mem_s1) {
            wave_lower_bound_sampled2(q.bf_mem, q.mem_s1, q.mem_s2, req_m, q.cpu_sorted, q.cpu_s1, q.cpu_s2, req_c, q.n, lane, top_m, top_c, start, r);
        } else {
            start = wave_lower_bound(q.bf_mem, q.n, req_m, lane);
            r = wave_lower_bound(q.cpu_sorted, q.n, req_c, lane);
        }
    }
    const uint64_t tol = (q.do_taint && q.ptol) ? q.ptol[pod] : 0ull;
    uint32_t sel[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) sel[k] = (q.psel && k < q.nkeys) ? q.psel[(size_t)k * q.p + pod] : 0u;  // wave-uniform
    return bestfit_rows_scan(q, pod, lane, start, r, start >> 6, req_c, tol, sel);
}

// diagnostics (BestfitRowsArgs::trace): lane 0 of the wave stamps the 100 MHz clock once the value DEP has arrived
#define KSCHED_BF_STAMP(BUF, SLOT, DEP)                                                                        \
    if ((BUF) && (threadIdx.x & 63u) == 0u) {                                                                  \
        uint64_t t_;                                                                                           \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"((uint32_t)(DEP)) : "memory"); \
        (BUF)[SLOT] = t_;                                                                                      \
    }
// One wave per pod, one launch slot per pod.  (A persistent grid of 8192 waves walking the pods was measured slower,
// 210 us against 147 us at the C5 shard: the kernel is bound by instruction issue -- ~280 scalar and ~240 vector
// instructions per pod, mostly address arithmetic and wave-uniform control flow, rocprofv3 SQ counters -- not by wave
// launches or memory latency, and the short-lived waves balance better.)
__global__ __launch_bounds__(256) void k_pick_bestfit_rows(const BestfitRowsArgs q) {
    kernarg_warm<sizeof(BestfitRowsArgs)>();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= q.p) return;
    const int32_t b = bestfit_rows_one_pod(q, wave, lane);
    if (lane == 0) q.binding[wave] = b;
}

// Second stage of the two-stage pick: the pods the lane-per-pod kernel could not decide in its first words.  One short-lived wave per
// handed-over pod, ONE WAVE PER BLOCK: with four waves per block a CU's SIMD whose six slots hold deep pods (ten times the median scan)
// blocks the placement of whole blocks while eighteen other slots idle (traced: 3 900 of 6 141 slots in use).  The grid covers a quarter
// of the lists' capacity, entries beyond that are walked (when > 1/4 of all pods are handed over).  Its own kernel (not a mode of
// k_pick_bestfit_rows): the stage is bound by wave slots, and without the one-stage kernel's rank searches it needs fewer registers.
__global__ __launch_bounds__(64) void k_pick_bestfit_handed(const BestfitRowsArgs q) {
    kernarg_warm<sizeof(BestfitRowsArgs)>();
    const uint32_t lane = threadIdx.x & 63u;
    {
        const uint32_t c = blockIdx.x % kBfSublists, stride = gridDim.x / kBfSublists;
        uint32_t i = blockIdx.x / kBfSublists;
        // the first stage hands over everything it had in registers -- one 64-byte record {pod, start, cpu rank, next word,
        // tolerations, cpu request, selector ids 0..7}: no rank search and no operand round trip here, the row loads go out at once.
        // The record is requested TOGETHER with the sub-list's count (the slot exists whether or not it was filled): one dependent
        // round trip less in every wave's life, and wave slots are what this stage is bound by.
        // (inline-asm loads: written as plain loads the compiler sinks them below the count's wait -- they are only used inside the loop)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *rec = reinterpret_cast<const u32x4 *>(q.pod_recs) + (size_t)(c * q.sub_cap + i) * 4u;
        u32x4 h, o, s0, s1;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(h) : "v"(rec) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(o) : "v"(rec) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(s0) : "v"(rec) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(s1) : "v"(rec) : "memory");
        const uint32_t count = q.sub_count[32u * c];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(h), "+v"(o), "+v"(s0), "+v"(s1) : "s"(count) : "memory");  // (after the count's load has been issued)
        // (every lane read the same record: back to scalar registers -- the rows' addresses are wave-uniform, and VGPRs are wave slots here)
        auto uni = [](u32x4 &v) {
            v.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.x);
            v.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.y);
            v.z = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.z);
            v.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.w);
        };
        uni(h);
        uni(o);
        uni(s0);
        uni(s1);
        for (; i < count; i += stride) {
        const uint32_t slot = c * q.sub_cap + i;
        uint64_t *const tr = q.trace2 ? q.trace2 + (size_t)slot * 4u : nullptr;
        KSCHED_BF_STAMP(tr, 0, slot);
        if (i >= stride) {  // (a walk: more than a quarter of the batch was handed over)
            rec = reinterpret_cast<const u32x4 *>(q.pod_recs) + (size_t)slot * 4u;
            h = rec[0], o = rec[1], s0 = rec[2], s1 = rec[3];
        }
        const uint32_t sel[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const int32_t b = bestfit_rows_scan(q, h.x, lane, h.y, h.z, h.w, (int64_t)(((uint64_t)o.w << 32) | o.z), ((uint64_t)o.y << 32) | o.x, sel);
        if (lane == 0) q.binding[h.x] = b;
        KSCHED_BF_STAMP(tr, 1, (uint32_t)b);
        if (tr && lane == 0) {
            uint32_t xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc), "=s"(hw));
            tr[2] = ((uint64_t)h.w << 32) | (uint32_t)b;  // (first word looked at, result)
            tr[3] = ((uint64_t)xcc << 32) | hw;           // where the wave ran
        }
        }
    }
}

// Best fit, first stage, ONE LANE PER POD.  The wave-per-pod kernel above spends a whole wave's chain of dependent round trips on
// every pod (rank searches -> rows -> winner; 135 us for 125k pods at the C5 shard, occupancy x latency bound), although the
// answer almost always sits in the first word of candidates: the pick is the first set bit, at or after `start`, of the AND of a
// few row words.  Here a lane does the whole pod: two interleaved 8-ary searches over level arrays (each round reads one 64-byte
// line per search: six dependent rounds for n <= 262144), then the row words at `start`'s word and the next one.  A pod with no
// feasible candidate among those <= 128 positions is appended to a list for the wave-per-pod kernel (well under 1 % of the pods).
// Same result as the wave-per-pod kernel by construction: both take the first position >= start whose bits are set in every row
// the pod ANDs and whose cpu fits.
__global__ __launch_bounds__(256) void k_pick_bestfit_lanes(const BestfitRowsArgs q) {
    kernarg_warm<sizeof(BestfitRowsArgs)>();
    const uint32_t pod = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0u && q.zero_next && threadIdx.x < kBfSublists) q.zero_next[32u * threadIdx.x] = 0u;  // (kBfSublists <= 256)
    if ((pod & ~63u) >= q.p) return;  // a whole wave past the end
    uint64_t *const tr = q.trace ? q.trace + (size_t)(pod >> 6) * 8u : nullptr;
    KSCHED_BF_STAMP(tr, 0, pod);
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    // No lane leaves before the wave has written its hand-over masks: a lane past the end, or one whose pod constrains a list key (no
    // bitmap rows: such a pod is picked from the key's sorted lists, k_pick_bestfit_listed), just skips the work.
    const bool live = pod < q.p;
    bool listed = false;
    if (live && q.nlist && q.psel)
        for (uint32_t j = 0; j < q.nlist; ++j) listed |= q.psel[(size_t)q.list_col[j] * q.p + pod] != 0u;
    const bool work = live && !listed;
    int32_t found = -1;
    bool undecided = false;
    uint32_t start = 0, r = 0;
    int64_t req_c = 0;
    uint64_t tol = 0ull;
    uint32_t sel[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (work) {
    req_c = q.do_fit ? q.pcpu[pod] : 0;
    const int64_t req_m = q.do_fit ? q.pmem[pod] : 0;
    tol = (q.do_taint && q.ptol) ? q.ptol[pod] : 0ull;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) sel[k] = (q.psel && k < q.nkeys) ? q.psel[(size_t)k * q.p + pod] : 0u;
    KSCHED_BF_STAMP(tr, 1, (uint32_t)req_c ^ (uint32_t)req_m ^ (uint32_t)tol ^ sel[0] ^ sel[1] ^ sel[2] ^ sel[3] ^ sel[4] ^ sel[5] ^ sel[6] ^ sel[7]);
    if (q.do_fit) {
        // lower bounds of req_m in bf_mem and of req_c in cpu_sorted, level by level from the top (block = 8 entries = one line); the two
        // searches' loads of a level go out together (written as two calls of one helper they were issued and waited for one after the other)
        auto count8x2 = [&](const int64_t *am, uint32_t basem, const int64_t *ac, uint32_t basec, uint32_t limit, uint32_t &cm, uint32_t &cc) {
            const bool okm = basem < limit, okc = basec < limit;  // (a search that has run off the end -- key above every entry -- reads block 0 and counts 0)
            const i64x2 *vm = reinterpret_cast<const i64x2 *>(am + (okm ? basem : 0u));  // bases are multiples of 8: 64-byte aligned
            const i64x2 *vc = reinterpret_cast<const i64x2 *>(ac + (okc ? basec : 0u));  // (reads past `limit` stay inside the padded arrays)
            const i64x2 m0 = vm[0], m1 = vm[1], m2 = vm[2], m3 = vm[3], c0 = vc[0], c1 = vc[1], c2 = vc[2], c3 = vc[3];
            const int64_t em[8] = {m0.x, m0.y, m1.x, m1.y, m2.x, m2.y, m3.x, m3.y}, ec[8] = {c0.x, c0.y, c1.x, c1.y, c2.x, c2.y, c3.x, c3.y};
            uint32_t nm = 0, nc = 0;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                nm += (okm && basem + j < limit && em[j] < req_m) ? 1u : 0u;
                nc += (okc && basec + j < limit && ec[j] < req_c) ? 1u : 0u;
            }
            cm = nm;
            cc = nc;
        };
        uint32_t bm = 0, bc = 0;  // block index at the level above
        for (uint32_t k = q.nlev; k >= 1u; --k) {
            uint32_t nk = q.n;  // entries of level k = ceil(n / 8^k)
            for (uint32_t t = 0; t < k; ++t) nk = (nk + 7u) / 8u;
            const int64_t *lm = q.lvl + q.lvl_off[k - 1u], *lc = lm + q.lvl_half;
            uint32_t cm, cc;
            count8x2(lm, bm * 8u, lc, bc * 8u, nk, cm, cc);
            bm = bm * 8u + cm;
            bc = bc * 8u + cc;
        }
        {
            uint32_t cm, cc;
            count8x2(q.bf_mem, bm * 8u, q.cpu_sorted, bc * 8u, q.n, cm, cc);
            start = bm * 8u + cm;
            r = bc * 8u + cc;
        }
        start = min(start, q.n);
        r = min(r, q.n);
    }
    KSCHED_BF_STAMP(tr, 2, start ^ r);
    undecided = start < q.n;
    // a required value no node carries (KSCHED_SEL_NEVER, or an id beyond the key's largest): the pod's AND of rows is empty --
    // no node, and no scan to the end of the snapshot to find that out (1 % of the constrained keys in the C5 workload)
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k)
        if (sel[k] != 0u && sel[k] > q.lab_max8[k]) undecided = false;
    if (undecided) {
        const uint32_t r_hi = q.do_fit ? q.row_cpu0 + (r + q.q - 1u) / q.q : q.row_valid;  // only nodes that fit
        const uint32_t r_lo = q.do_fit ? q.row_cpu0 + r / q.q : q.row_valid;               // every node that fits
        uint32_t lrow[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) lrow[k] = (sel[k] <= q.lab_max8[k]) ? q.lab_base8[k] + sel[k] - 1u : q.row_zero;
        const uint32_t w0 = start >> 6;
        struct Word {
            uint64_t base, hi, lo;
        };
        // first candidate of word `w` that fits (sure bits at once, the others by their exact cpu); returns whether one was found
        auto take_word = [&](uint32_t w, Word x) -> bool {
            const uint64_t sure = q.do_fit ? (x.base & x.hi) : x.base;
            uint64_t cand = q.do_fit ? (sure | (x.base & x.lo & ~x.hi)) : x.base;
            while (cand) {
                const uint32_t b = (uint32_t)__builtin_ctzll(cand);
                const uint32_t i = w * 64u + b;
                if (((sure >> b) & 1ull) || req_c <= q.bf_cpu[i]) {
                    found = (int32_t)q.bf_order[i];
                    return true;
                }
                cand &= cand - 1ull;
            }
            return false;
        };
        // One trip = HALF A LINE (4 words = 256 positions) of every row, as two 16-byte requests per row, and ALL of a trip's requests in
        // flight before anything waits: loads first, ANDs after.  (Written as `if (constrained) base &= row[..]` the compiler waits for each
        // row inside its own branch -- `s_waitcnt vmcnt(0)` behind every load in the disassembly -- and a trip was up to fourteen dependent
        // round trips instead of one.)  The stage looks at the 64-byte block of `start`'s word and the next `lane_blocks - 1` blocks:
        // 9 .. 16 words by default, two to four trips; the hand-over point is a 64-byte boundary.
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const uint32_t h_end = min(((w0 >> 3) + q.lane_blocks) * 2u, q.Wbf >> 2);  // half-blocks [w0 >> 2, h_end) are this stage's (Wbf is a multiple of 8)
        for (uint32_t hb = w0 >> 2; hb < h_end && undecided; ++hb) {
            const uint64_t *const at = q.rows + 4u * hb;
            auto ld = [&](uint32_t row, u64x2 &a, u64x2 &b) {
                const u64x2 *v = reinterpret_cast<const u64x2 *>(at + (size_t)row * q.Wbf);
                a = v[0];
                b = v[1];
            };
            const u64x2 ones = {~0ull, ~0ull}, zero = {0ull, 0ull};
            u64x2 ha = zero, hb2 = zero, la = zero, lb = zero, ba = ones, bb = ones, xa[8], xb[8], ta[4], tb[4];
            // (every row holds bits of live positions only: with the cpu rows in the AND the all-valid row adds nothing -- one request less)
            if (q.do_fit) {
                ld(r_hi, ha, hb2);
                ld(r_lo, la, lb);
            } else {
                ld(q.row_valid, ba, bb);
            }
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                xa[k] = ones;
                xb[k] = ones;
                if (sel[k] != 0u) ld(lrow[k], xa[k], xb[k]);
            }
#pragma unroll
            for (uint32_t g = 0; g < 4; ++g) {
                ta[g] = ones;
                tb[g] = ones;
                if (q.do_taint && g < q.ngroups) ld(q.row_taint + 16u * g + (uint32_t)((tol >> (4u * g)) & 15ull), ta[g], tb[g]);
            }
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                ba &= xa[k];
                bb &= xb[k];
            }
#pragma unroll
            for (uint32_t g = 0; g < 4; ++g) {
                ba &= ta[g];
                bb &= tb[g];
            }
            for (uint32_t k = 8; k < q.nkeys; ++k) {  // (rare: more than eight label keys, more than sixteen taints)
                const uint32_t s = q.psel[(size_t)k * q.p + pod];
                if (s != 0u) {
                    u64x2 a, b;
                    ld((s <= q.lab_meta[32u + k]) ? q.lab_meta[k] + s - 1u : q.row_zero, a, b);
                    ba &= a;
                    bb &= b;
                }
            }
            if (q.do_taint)
                for (uint32_t g = 4; g < q.ngroups; ++g) {
                    u64x2 a, b;
                    ld(q.row_taint + 16u * g + (uint32_t)((tol >> (4u * g)) & 15ull), a, b);
                    ba &= a;
                    bb &= b;
                }
            const uint64_t base[4] = {ba.x, ba.y, bb.x, bb.y}, hi[4] = {ha.x, ha.y, hb2.x, hb2.y}, lo[4] = {la.x, la.y, lb.x, lb.y};
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t w = 4u * hb + j;
                if (!undecided || w < w0) continue;  // (words of the half-block before `start`'s)
                Word x{base[j], hi[j], lo[j]};
                if (w == w0) x.base &= ~0ull << (start & 63u);  // positions before `start` cannot hold the pod's memory
                if (take_word(w, x)) undecided = false;
            }
        }
        if (undecided && h_end >= (q.Wbf >> 2)) undecided = false;  // those were the last words: no feasible node
    }
    }  // if (work)
    KSCHED_BF_STAMP(tr, 3, (uint32_t)found ^ (uint32_t)undecided);
    // the pods of this wave that go on to the second stage: ONE returning atomic per wave, on the counter of the wave's sub-list
    const uint64_t um = __ballot(undecided), lm = __ballot(listed);
    const uint32_t lane = threadIdx.x & 63u, sub = (pod >> 6) % kBfSublists;
    if (lane == 0u) {
        if (q.listed_mask) q.listed_mask[pod >> 6] = lm;  // (listed pods: wave = pod in k_pick_bestfit_listed, one mask word per wave, nothing to zero)
        if (tr) tr[5] = (uint64_t)__popcll(um);
    }
    if (um) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(um);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(q.handover_count + 32u * sub, (uint32_t)__popcll(um));
        base = (uint32_t)__shfl((int)base, (int)leader, 64);
        if (undecided) {  // the wave-per-pod kernel scans on behind the words looked at here
            const uint32_t slot = sub * q.sub_cap + base + (uint32_t)__popcll(um & ((1ull << lane) - 1ull));
            uint4 *rec = reinterpret_cast<uint4 *>(q.handover_recs) + (size_t)slot * 4u;
            rec[0] = make_uint4(pod, start, r, ((start >> 9) + q.lane_blocks) << 3);  // (the next word: a 64-byte boundary)
            rec[1] = make_uint4((uint32_t)tol, (uint32_t)(tol >> 32), (uint32_t)(uint64_t)req_c, (uint32_t)((uint64_t)req_c >> 32));
            rec[2] = make_uint4(sel[0], sel[1], sel[2], sel[3]);
            rec[3] = make_uint4(sel[4], sel[5], sel[6], sel[7]);
        }
    }
    if (work && !undecided) q.binding[pod] = found;
    KSCHED_BF_STAMP(tr, 4, pod);
}

// Best fit for pods that constrain a LIST key (a high-cardinality label key kept per tile as a sorted list, tile_index.hpp): only
// the few nodes carrying the pod's value can be feasible at all, so instead of scanning best-fit positions the wave walks those
// nodes -- lane = tile: two lower bounds in the tile's list give the range, every node of the range is tested in full from its
// node record (same predicate as k_select_sampled) -- and takes the one with the smallest best-fit position.
struct BestfitListedArgs {
    const uint8_t *lists;          // IndexedSnapshot::d_list: [tiles][nlist][kListBytes]
    const int64_t *nrec;           // node records
    const uint32_t *nlab;          // [nkeys][n]
    const uint32_t *bf_rank, *bf_order;
    const int64_t *pcpu, *pmem;
    const uint32_t *psel;
    const uint64_t *ptol;
    const uint64_t *listed_mask;   // [ceil(p / 64)], from k_pick_bestfit_lanes: wave = pod
    int32_t *binding;
    uint32_t p, n, nkeys, tiles, nlist, list_col[2], do_fit, do_taint;
};
__global__ __launch_bounds__(256) void k_pick_bestfit_listed(const BestfitListedArgs q) {
    kernarg_warm<sizeof(BestfitListedArgs)>();
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= q.p) return;
    if (!((q.listed_mask[wave >> 6] >> (wave & 63u)) & 1ull)) return;
    const uint32_t pod = wave;
    const int64_t rc = q.do_fit ? q.pcpu[pod] : 0, rm = q.do_fit ? q.pmem[pod] : 0;
    const uint64_t tol = (q.do_taint && q.ptol) ? q.ptol[pod] : 0ull;
    // walk the lists of the first list key the pod constrains; the other constraints are checked per candidate
    uint32_t j0 = 0;
    while (j0 + 1u < q.nlist && q.psel[(size_t)q.list_col[j0] * q.p + pod] == 0u) ++j0;
    const uint32_t want = q.psel[(size_t)q.list_col[j0] * q.p + pod];
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t t = lane; t < q.tiles; t += 64u) {
        const uint8_t *L = q.lists + ((size_t)t * q.nlist + j0) * 6144u;  // kListBytes
        const uint32_t *vals = reinterpret_cast<const uint32_t *>(L);
        const uint16_t *nodes = reinterpret_cast<const uint16_t *>(L + 4096u);
        uint32_t lo = 0;  // entries below `want`
        for (uint32_t half = 512u; half >= 1u; half >>= 1) lo += (vals[lo + half - 1u] < want) ? half : 0u;
        lo += (vals[lo] < want) ? 1u : 0u;
        for (uint32_t e = lo; e < 1024u && vals[e] == want; ++e) {
            const uint32_t node = t * 1024u + nodes[e];
            if (node >= q.n) continue;  // (padding carries id 0, never a wanted id; kept for safety)
            bool f = true;
            if (q.do_fit) {
                const i64x2 cm = reinterpret_cast<const i64x2 *>(q.nrec)[(size_t)node * 4u];
                f = rc <= cm.x && rm <= cm.y;  // src/predicates.rs:42
            }
            if (f && q.do_taint) f = ((uint64_t)q.nrec[(size_t)node * kNodeRecWords + 2u] & ~tol) == 0ull;
            for (uint32_t k = 0; f && k < q.nkeys; ++k) {  // src/predicates.rs:48-57, every key incl. the list keys
                const uint32_t s = q.psel[(size_t)k * q.p + pod];
                if (s != 0u && s != q.nlab[(size_t)k * q.n + node]) f = false;
            }
            if (f) best = min(best, q.bf_rank[node]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off, 64));
    if (lane == 0) q.binding[pod] = (best == 0xFFFFFFFFu) ? -1 : (int32_t)q.bf_order[best];
}

}  // namespace ksched
