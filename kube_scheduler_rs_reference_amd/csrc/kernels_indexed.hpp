// kernels_indexed.hpp -- "indexed" mask kernel: per-tile bitmap index resident in LDS, all
// predicate work done 64 nodes (one mask word) at a time.
//
// Why: the output (P x ceil(N/64) words) is the HBM traffic; deciding every bit with its own
// compare (kernels_direct.hpp) costs >= 4 VALU issues per output word per wave64 and lands an
// order of magnitude under the HBM write roofline.  Here every predicate is turned into an AND
// of precomputed node bitmaps, so one VALU op decides 64 x 64 pairs.
//
// Snapshot index (built once per ksched_set_nodes, per tile of kTileNodes = 1024 nodes = 16 words):
//   * fit   -- src/predicates.rs:42  req <= avail.  Sort the tile's avail values; node n gets its
//              position pos[n] in that order (ties broken by node index, so pos is a
//              permutation).  For a pod, r = #values < req (lower bound, binary search in LDS);
//              then  req <= avail[n]  <=>  pos[n] >= r, exactly, for any int64 inputs.
//              pos >= r is evaluated two-level, pos = 32*hi + lo, r = 32*rh + rl:
//                  pos >= r  <=>  hi > rh  ||  (hi == rh && lo >= rl)
//                            <=>  GEH[rh] & (GEH[rh+1] | GEL[rl])          (GEH[h] = {n: hi >= h})
//              3 bitmap rows per resource instead of 1025 rows for a one-level table.
//   * sel   -- src/predicates.rs:45-61.  One bitmap row per (key, value id): nodes carrying that
//              value.  A pod ANDs the rows of the keys it constrains; unconstrained keys cost
//              nothing; KSCHED_SEL_NEVER / unknown ids hit the all-zero row.
//   * taint -- (taints[n] & ~tol[p]) == 0.  Per 4-bit group g of taint bits and per tolerated
//              subset s of that group: row {n : taints_g[n] subset of s}; a pod ANDs one row per
//              group.
// All rows are 16 words (128 B); padding bits (node >= N) are zero in every row, so they are
// zero in every result.
//
// Kernel: a block owns one tile (its rows + the two sorted arrays staged in LDS, <= 160 KiB) and
// a contiguous chunk of pods.  Per batch of blockDim pods: prologue, one lane per pod (binary
// searches, row ids -> LDS parameter slots); main loop, 8 lanes per pod x 2 words per lane:
// ds_read_b128 of the selected rows, AND/OR, one 16-byte store per lane -> each wave store
// instruction emits eight 128-byte row segments.  Blocks of adjacent tiles of the same pod chunk
// are mapped to the same XCD (block id % 8) so partial cache lines at tile seams merge in one L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <climits>
#include <numeric>
#include <vector>

#include "../../include/ksched.h"

namespace ksched {

constexpr int kTileWords = 16;
constexpr int kTileNodes = kTileWords * 64;  // 1024
constexpr int kFitHi = 34;                   // GEH[0..33] (GEH[32], GEH[33] are zero rows)
constexpr int kFitLo = 32;                   // GEL[0..31]
constexpr int kIdxMaxKeys = 32;
constexpr int kIdxMaxGroups = 16;            // 64 taint bits / 4
constexpr uint32_t kLdsBudget = 160u * 1024u;

struct IndexedLayout {
    uint32_t n, W, tiles, rows, nkeys, ngroups;
    uint32_t row_zero, row_valid;
    uint32_t row_cpu_hi, row_cpu_lo, row_mem_hi, row_mem_lo;
    uint32_t row_taint;                 // + 16 * group + subset
    uint32_t lab_base[kIdxMaxKeys];     // row of value id 1 of key k
    uint32_t lab_max[kIdxMaxKeys];      // largest id with a row
};

struct IndexedSnapshot {
    bool built = false;
    IndexedLayout lay{};
    int64_t *d_sorted_cpu = nullptr;  // [tiles][1024], padded with INT64_MAX
    int64_t *d_sorted_mem = nullptr;
    uint64_t *d_tables = nullptr;     // [tiles][rows][16]
    size_t sorted_cap = 0, tables_cap = 0;
};

inline void indexed_release(IndexedSnapshot &s) {
    if (s.d_sorted_cpu) (void)hipFree(s.d_sorted_cpu);
    if (s.d_sorted_mem) (void)hipFree(s.d_sorted_mem);
    if (s.d_tables) (void)hipFree(s.d_tables);
    s = IndexedSnapshot{};
}

// bytes of LDS parameter slot per pod: 4 x u8 fit ranks, u16 label count, u16 x nkeys label rows,
// u16 x ngroups taint rows; rounded up to 8 bytes
inline uint32_t indexed_param_stride(const IndexedLayout &l) {
    const uint32_t b = 4 + 2 + 2 * l.nkeys + 2 * l.ngroups;
    return (b + 7u) & ~7u;
}
inline uint32_t indexed_lds_bytes(const IndexedLayout &l, uint32_t block_threads) {
    return l.rows * 128u + 2u * kTileNodes * 8u + block_threads * indexed_param_stride(l);
}

// Build the per-tile index on the host and upload it.  Leaves s.built == false (and returns
// hipSuccess) when the snapshot is outside what the indexed kernel supports; the caller then
// uses the direct kernel.
inline hipError_t indexed_build(IndexedSnapshot &s, uint32_t n, const int64_t *cpu, const int64_t *mem, const uint32_t *lab,
                                uint32_t nkeys, const uint64_t *taints) {
    s.built = false;
    if (n == 0 || nkeys > kIdxMaxKeys) return hipSuccess;
    IndexedLayout l{};
    l.n = n;
    l.W = (n + 63u) / 64u;
    l.tiles = (n + kTileNodes - 1) / kTileNodes;
    l.nkeys = nkeys;
    uint64_t all_taints = 0;
    if (taints)
        for (uint32_t i = 0; i < n; ++i) all_taints |= taints[i];
    l.ngroups = all_taints ? (uint32_t)((64 - __builtin_clzll(all_taints)) + 3) / 4 : 0;

    uint32_t r = 0;
    l.row_zero = r++;
    l.row_valid = r++;
    l.row_cpu_hi = r; r += kFitHi;
    l.row_cpu_lo = r; r += kFitLo;
    l.row_mem_hi = r; r += kFitHi;
    l.row_mem_lo = r; r += kFitLo;
    l.row_taint = r; r += 16 * l.ngroups;
    uint64_t label_rows = 0;
    for (uint32_t k = 0; k < nkeys; ++k) {
        uint32_t mx = 0;
        for (uint32_t i = 0; i < n; ++i) mx = std::max(mx, lab[(size_t)k * n + i]);
        l.lab_max[k] = mx;
        label_rows += mx;
    }
    // all rows + sorted arrays + parameter slots of a 512-thread block must fit in LDS, and row
    // ids must fit the u16 parameter fields
    if (r + label_rows > 60000) return hipSuccess;
    for (uint32_t k = 0; k < nkeys; ++k) {
        l.lab_base[k] = r;
        r += l.lab_max[k];
    }
    l.rows = r;
    if (indexed_lds_bytes(l, 512) > kLdsBudget) return hipSuccess;

    const size_t tile_words = (size_t)l.rows * kTileWords;
    std::vector<uint64_t> tab((size_t)l.tiles * tile_words, 0ull);
    std::vector<int64_t> scpu((size_t)l.tiles * kTileNodes, INT64_MAX), smem((size_t)l.tiles * kTileNodes, INT64_MAX);
    std::vector<uint32_t> ord(kTileNodes);
    for (uint32_t t = 0; t < l.tiles; ++t) {
        const uint32_t base = t * kTileNodes;
        const uint32_t m = std::min<uint32_t>(kTileNodes, n - base);
        uint64_t *T = tab.data() + (size_t)t * tile_words;
        auto setbit = [&](uint32_t row, uint32_t local) { T[(size_t)row * kTileWords + (local >> 6)] |= 1ull << (local & 63u); };
        for (uint32_t i = 0; i < m; ++i) setbit(l.row_valid, i);
        // fit: positions in the sorted order of each resource
        for (int res = 0; res < 2; ++res) {
            const int64_t *v = res == 0 ? cpu : mem;
            int64_t *sorted = (res == 0 ? scpu.data() : smem.data()) + (size_t)t * kTileNodes;
            const uint32_t row_hi = res == 0 ? l.row_cpu_hi : l.row_mem_hi;
            const uint32_t row_lo = res == 0 ? l.row_cpu_lo : l.row_mem_lo;
            std::iota(ord.begin(), ord.begin() + m, 0u);
            std::stable_sort(ord.begin(), ord.begin() + m, [&](uint32_t a, uint32_t b) { return v[base + a] < v[base + b]; });
            for (uint32_t pos = 0; pos < m; ++pos) {
                const uint32_t local = ord[pos];
                sorted[pos] = v[base + local];
                const uint32_t hi = pos >> 5, lo = pos & 31u;
                for (uint32_t h = 0; h <= hi; ++h) setbit(row_hi + h, local);  // GEH[h] = {hi >= h}
                for (uint32_t q = 0; q <= lo; ++q) setbit(row_lo + q, local);  // GEL[q] = {lo >= q}
            }
        }
        // labels
        for (uint32_t k = 0; k < nkeys; ++k)
            for (uint32_t i = 0; i < m; ++i) {
                const uint32_t id = lab[(size_t)k * n + base + i];
                if (id) setbit(l.lab_base[k] + id - 1, i);
            }
        // taints: row (g, s) = nodes whose taint bits of group g are a subset of s
        for (uint32_t g = 0; g < l.ngroups; ++g)
            for (uint32_t i = 0; i < m; ++i) {
                const uint32_t tg = (uint32_t)((taints[base + i] >> (4 * g)) & 15ull);
                for (uint32_t sub = 0; sub < 16; ++sub)
                    if ((tg & ~sub) == 0) setbit(l.row_taint + 16 * g + sub, i);
            }
    }
    hipError_t e;
    const size_t sorted_elems = (size_t)l.tiles * kTileNodes;
    if (sorted_elems > s.sorted_cap) {
        if (s.d_sorted_cpu) (void)hipFree(s.d_sorted_cpu);
        if (s.d_sorted_mem) (void)hipFree(s.d_sorted_mem);
        s.d_sorted_cpu = s.d_sorted_mem = nullptr;
        s.sorted_cap = 0;
        if ((e = hipMalloc((void **)&s.d_sorted_cpu, sorted_elems * 8)) != hipSuccess) return e;
        if ((e = hipMalloc((void **)&s.d_sorted_mem, sorted_elems * 8)) != hipSuccess) return e;
        s.sorted_cap = sorted_elems;
    }
    if (tab.size() > s.tables_cap) {
        if (s.d_tables) (void)hipFree(s.d_tables);
        s.d_tables = nullptr;
        s.tables_cap = 0;
        if ((e = hipMalloc((void **)&s.d_tables, tab.size() * 8)) != hipSuccess) return e;
        s.tables_cap = tab.size();
    }
    if ((e = hipMemcpy(s.d_sorted_cpu, scpu.data(), sorted_elems * 8, hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(s.d_sorted_mem, smem.data(), sorted_elems * 8, hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(s.d_tables, tab.data(), tab.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return e;
    s.lay = l;
    s.built = true;
    return hipSuccess;
}

inline bool indexed_applicable(const IndexedSnapshot &s, uint32_t /*flags*/, bool /*have_sel*/) { return s.built; }
inline size_t indexed_scratch_bytes(const IndexedSnapshot &, uint32_t) { return 0; }

struct IndexedArgs {
    IndexedLayout lay;
    uint32_t p;
    uint32_t chunks;          // pod chunks; chunk c = rows [c * pods_per_chunk, ...)
    uint32_t pods_per_chunk;
    uint32_t do_fit, do_sel, do_taint;
    uint32_t param_stride;    // bytes
};

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));

// number of sorted values strictly below `req` (sorted[] has kTileNodes entries, padded with INT64_MAX)
__device__ __forceinline__ uint32_t lower_bound_1024(const int64_t *sorted, int64_t req) {
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = kTileNodes / 2; step >= 1; step >>= 1)
        if (sorted[lo + step - 1] < req) lo += step;
    // lo in [0, 1023]; one more probe decides 1023 vs 1024
    if (sorted[lo] < req) lo += 1;
    return lo;
}

template <bool WANT_FIT>
__global__ __launch_bounds__(512) void k_eval_indexed(const int64_t *__restrict__ g_sorted_cpu, const int64_t *__restrict__ g_sorted_mem,
                                                       const uint64_t *__restrict__ g_tables, const int64_t *__restrict__ g_pcpu,
                                                       const int64_t *__restrict__ g_pmem, const uint32_t *__restrict__ g_psel,
                                                       const uint64_t *__restrict__ g_ptol, uint64_t *__restrict__ out_feas,
                                                       uint64_t *__restrict__ out_fit, const IndexedArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const IndexedLayout &L = a.lay;
    // XCD-aware work mapping: block b runs on XCD b % 8 (observed dispatch order; speed only).
    // All tiles of one pod chunk get the same XCD so seam cache lines meet in one L2.
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, i = b >> 3;
    const uint32_t tile = i % L.tiles;
    const uint32_t chunk = (i / L.tiles) * 8u + xcd;
    if (chunk >= a.chunks) return;

    uint64_t *tab = reinterpret_cast<uint64_t *>(smem);                                  // [rows][16]
    int64_t *s_cpu = reinterpret_cast<int64_t *>(smem + (size_t)L.rows * 128u);          // [1024]
    int64_t *s_mem = s_cpu + kTileNodes;                                                 // [1024]
    uint8_t *params = reinterpret_cast<uint8_t *>(s_mem + kTileNodes);                   // [blockDim][stride]

    // ---- stage this tile's index into LDS (16-byte vectors) -------------------------------
    {
        const u64x2 *src = reinterpret_cast<const u64x2 *>(g_tables + (size_t)tile * L.rows * kTileWords);
        u64x2 *dst = reinterpret_cast<u64x2 *>(tab);
        const uint32_t nvec = L.rows * (kTileWords / 2);
        for (uint32_t v = threadIdx.x; v < nvec; v += blockDim.x) dst[v] = src[v];
        if (a.do_fit) {
            const u64x2 *sc = reinterpret_cast<const u64x2 *>(g_sorted_cpu + (size_t)tile * kTileNodes);
            const u64x2 *sm = reinterpret_cast<const u64x2 *>(g_sorted_mem + (size_t)tile * kTileNodes);
            u64x2 *dc = reinterpret_cast<u64x2 *>(s_cpu);
            u64x2 *dm = reinterpret_cast<u64x2 *>(s_mem);
            for (uint32_t v = threadIdx.x; v < kTileNodes / 2; v += blockDim.x) {
                dc[v] = sc[v];
                dm[v] = sm[v];
            }
        }
    }
    __syncthreads();

    const uint32_t pod_lo = chunk * a.pods_per_chunk;
    const uint32_t pod_hi = min(a.p, pod_lo + a.pods_per_chunk);
    const uint32_t wp = threadIdx.x & 7u;             // word pair inside the tile: words 2wp, 2wp+1
    const uint32_t sub = threadIdx.x >> 3;            // pod slot inside one main-loop pass
    const uint32_t pods_per_pass = blockDim.x >> 3;
    const uint32_t w0 = tile * kTileWords + 2u * wp;  // first global word of this lane
    const bool has0 = w0 < L.W, has1 = w0 + 1 < L.W;

    for (uint32_t b0 = pod_lo; b0 < pod_hi; b0 += blockDim.x) {
        // ---- prologue: one lane per pod ------------------------------------------------
        {
            const uint32_t pod = b0 + threadIdx.x;
            if (pod < pod_hi) {
                uint8_t *pp = params + (size_t)threadIdx.x * a.param_stride;
                uint32_t fitw = 0;
                if (a.do_fit) {
                    const uint32_t rc = lower_bound_1024(s_cpu, g_pcpu[pod]);
                    const uint32_t rm = lower_bound_1024(s_mem, g_pmem[pod]);
                    fitw = (rc >> 5) | ((rc & 31u) << 8) | ((rm >> 5) << 16) | ((rm & 31u) << 24);
                }
                *reinterpret_cast<uint32_t *>(pp) = fitw;
                uint16_t *lab = reinterpret_cast<uint16_t *>(pp + 6);
                uint32_t cnt = 0;
                if (a.do_sel) {
                    for (uint32_t k = 0; k < L.nkeys; ++k) {
                        const uint32_t s = g_psel[(size_t)k * a.p + pod];
                        if (s != 0u) {
                            lab[cnt++] = (uint16_t)((s <= L.lab_max[k]) ? (L.lab_base[k] + s - 1u) : L.row_zero);
                        }
                    }
                }
                *reinterpret_cast<uint16_t *>(pp + 4) = (uint16_t)cnt;
                if (a.do_taint) {
                    uint16_t *tn = lab + L.nkeys;
                    const uint64_t tol = g_ptol ? g_ptol[pod] : 0ull;
                    for (uint32_t g = 0; g < L.ngroups; ++g)
                        tn[g] = (uint16_t)(L.row_taint + 16u * g + (uint32_t)((tol >> (4u * g)) & 15ull));
                }
            }
        }
        __syncthreads();
        // ---- main loop: 8 lanes per pod, 2 words per lane --------------------------------
        const uint32_t npods = min(blockDim.x, pod_hi - b0);
        for (uint32_t s0 = 0; s0 < npods; s0 += pods_per_pass) {
            const uint32_t slot = s0 + sub;
            if (slot < npods && has0) {
                const uint8_t *pp = params + (size_t)slot * a.param_stride;
                const u64x2 *T = reinterpret_cast<const u64x2 *>(tab) + wp;  // row r -> T[r * 8]
                u64x2 f;
                if (a.do_fit) {
                    const uint32_t fitw = *reinterpret_cast<const uint32_t *>(pp);
                    const uint32_t ch = fitw & 255u, cl = (fitw >> 8) & 255u, mh = (fitw >> 16) & 255u, ml = fitw >> 24;
                    const u64x2 c = T[(L.row_cpu_hi + ch) * 8u] & (T[(L.row_cpu_hi + ch + 1u) * 8u] | T[(L.row_cpu_lo + cl) * 8u]);
                    const u64x2 m = T[(L.row_mem_hi + mh) * 8u] & (T[(L.row_mem_hi + mh + 1u) * 8u] | T[(L.row_mem_lo + ml) * 8u]);
                    f = c & m;
                } else {
                    f = T[L.row_valid * 8u];
                }
                const size_t o = (size_t)(b0 + slot) * L.W + w0;
                if (WANT_FIT) {
                    if (has1) *reinterpret_cast<u64x2_a8 *>(out_fit + o) = f;
                    else out_fit[o] = f.x;
                }
                if (a.do_sel) {
                    const uint32_t cnt = *reinterpret_cast<const uint16_t *>(pp + 4);
                    const uint16_t *lab = reinterpret_cast<const uint16_t *>(pp + 6);
                    for (uint32_t q = 0; q < cnt; ++q) f &= T[(uint32_t)lab[q] * 8u];
                }
                if (a.do_taint) {
                    const uint16_t *tn = reinterpret_cast<const uint16_t *>(pp + 6) + L.nkeys;
                    for (uint32_t g = 0; g < L.ngroups; ++g) f &= T[(uint32_t)tn[g] * 8u];
                }
                if (out_feas) {
                    if (has1) *reinterpret_cast<u64x2_a8 *>(out_feas + o) = f;
                    else out_feas[o] = f.x;
                }
            }
        }
        __syncthreads();
    }
}

// Launch geometry: tiles x chunks blocks (chunks rounded up to a multiple of 8 for the XCD map).
inline hipError_t run_indexed(const IndexedSnapshot &s, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                              const uint64_t *ptol, uint32_t flags, uint64_t *out_feas, uint64_t *out_fit, uint8_t * /*scratch*/,
                              hipStream_t stream) {
    const IndexedLayout &l = s.lay;
    IndexedArgs a{};
    a.lay = l;
    a.p = p;
    a.do_fit = (flags & KSCHED_FIT) ? 1u : 0u;
    a.do_sel = ((flags & KSCHED_SEL) && psel && l.nkeys) ? 1u : 0u;
    a.do_taint = ((flags & KSCHED_TAINT) && l.ngroups) ? 1u : 0u;
    a.param_stride = indexed_param_stride(l);
    const uint32_t threads = 512;
    const uint32_t lds = indexed_lds_bytes(l, threads);
    const uint32_t blocks_per_cu = std::max(1u, std::min(kLdsBudget / lds, 2048u / threads));
    const uint32_t slots = 256u * blocks_per_cu;
    // chunks: enough blocks to fill the chip a whole number of times, never less than one batch of pods per block
    const uint32_t max_chunks = std::max(1u, (p + threads - 1) / threads);
    uint32_t best = 1;
    double best_eff = -1.0;
    for (uint32_t waves = 1; waves <= 4; ++waves) {
        uint32_t ch = std::max(1u, std::min(max_chunks, (slots * waves) / l.tiles));
        const uint32_t blocks = ch * l.tiles;
        const double eff = (double)blocks / (double)(((blocks + slots - 1) / slots) * slots);
        if (eff > best_eff + 0.02) {
            best_eff = eff;
            best = ch;
        }
    }
    a.chunks = best;
    a.pods_per_chunk = (p + a.chunks - 1) / a.chunks;
    a.chunks = (p + a.pods_per_chunk - 1) / a.pods_per_chunk;
    const uint32_t chunks8 = (a.chunks + 7u) & ~7u;
    const dim3 grid(chunks8 * l.tiles);
    const bool want_fit = (flags & KSCHED_WANT_FIT_MASK) && out_fit;
    hipError_t e;
    if (want_fit) {
        e = hipFuncSetAttribute((const void *)k_eval_indexed<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_eval_indexed<true>, grid, dim3(threads), lds, stream, s.d_sorted_cpu, s.d_sorted_mem, s.d_tables, pcpu,
                           pmem, psel, ptol, out_feas, out_fit, a);
    } else {
        e = hipFuncSetAttribute((const void *)k_eval_indexed<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_eval_indexed<false>, grid, dim3(threads), lds, stream, s.d_sorted_cpu, s.d_sorted_mem, s.d_tables, pcpu,
                           pmem, psel, ptol, out_feas, out_fit, a);
    }
    return hipGetLastError();
}

}  // namespace ksched
