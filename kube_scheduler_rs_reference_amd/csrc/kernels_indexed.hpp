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
// Two kernels per evaluation:
//   k_index_pods  (pre-pass, tiny): grid (tile, pod chunk); stages the tile's two sorted arrays
//       (16 KiB LDS, high occupancy) and, one lane per pod, runs the two branch-free binary
//       searches -> one packed 4-byte word of fit ranks per (tile, pod).  Tile-0 blocks also turn
//       the pod's selector ids / toleration bits into row ids: 16 bytes per pod (4 label rows +
//       4 taint rows; rarely-needed extras go to an overflow array).  Extra HBM traffic: 8 bytes
//       per (pod, tile) written+read against 128 bytes of mask per (pod, tile): ~6 %.
//   k_eval_indexed (main): a block owns one tile (its bitmap rows staged in LDS, <= 160 KiB) and
//       a chunk of pods.  8 lanes per pod x 2 words per lane: two coalesced global loads of the
//       pod's parameters, ds_read_b128 of the selected rows, AND/OR, one 16-byte store per lane,
//       so each wave store instruction emits eight 128-byte row segments.  No barriers after
//       staging; iterations are independent, so loads of later pods overlap the LDS reads of
//       earlier ones.  Blocks of adjacent tiles of the same pod chunk are mapped to the same XCD
//       (block id % 8) so partial cache lines at tile seams merge in one L2.
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

inline uint32_t indexed_lds_bytes(const IndexedLayout &l) { return l.rows * 128u; }
// u16 entries per pod in the overflow array: label rows beyond the 4 inline slots, taint rows beyond 4
inline uint32_t indexed_ext_stride(const IndexedLayout &l) {
    return (l.nkeys > 3 ? l.nkeys - 3 + 1 : 0) + (l.ngroups > 4 ? l.ngroups - 4 : 0);
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
    // all rows of a tile must fit in LDS and row ids must fit 16 bits (0xFFFF is a sentinel)
    if (r + label_rows > 1280) return hipSuccess;
    for (uint32_t k = 0; k < nkeys; ++k) {
        l.lab_base[k] = r;
        r += l.lab_max[k];
    }
    l.rows = r;
    if (indexed_lds_bytes(l) > kLdsBudget) return hipSuccess;

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

// device scratch per evaluation: rec16[p] (16 B) + ext[p][ext_stride] (u16) + fitw[tiles][p] (u32)
inline size_t indexed_scratch_bytes(const IndexedSnapshot &s, uint32_t p) {
    const IndexedLayout &l = s.lay;
    size_t b = (size_t)p * 16u;
    b += ((size_t)p * indexed_ext_stride(l) * 2u + 15u) & ~(size_t)15u;
    b += (size_t)l.tiles * p * 4u;
    return b + 64;
}

struct IndexedArgs {
    IndexedLayout lay;
    uint32_t p;
    uint32_t chunks;          // pod chunks of the main kernel; chunk c = rows [c * pods_per_chunk, ...)
    uint32_t pods_per_chunk;
    uint32_t ext_stride;      // u16 entries per pod in ext
    uint32_t do_fit, do_sel, do_taint;
    uint32_t debug;           // ablation bits (KSCHED_OPT_DEBUG; timing experiments only, results invalid when != 0)
};

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));

// ---- pre-pass: fit ranks per (tile, pod), row ids per pod ------------------------------------------
__global__ __launch_bounds__(256) void k_index_pods(const int64_t *__restrict__ g_sorted_cpu, const int64_t *__restrict__ g_sorted_mem,
                                                     const int64_t *__restrict__ g_pcpu, const int64_t *__restrict__ g_pmem,
                                                     const uint32_t *__restrict__ g_psel, const uint64_t *__restrict__ g_ptol,
                                                     uint4 *__restrict__ rec16, uint16_t *__restrict__ ext, uint32_t *__restrict__ fitw,
                                                     const IndexedArgs a, uint32_t pods_per_block) {
    __shared__ __attribute__((aligned(16))) int64_t s_cpu[kTileNodes];
    __shared__ __attribute__((aligned(16))) int64_t s_mem[kTileNodes];
    const IndexedLayout &L = a.lay;
    const uint32_t tile = blockIdx.x;
    if (a.do_fit) {
        const u64x2 *sc = reinterpret_cast<const u64x2 *>(g_sorted_cpu + (size_t)tile * kTileNodes);
        const u64x2 *sm = reinterpret_cast<const u64x2 *>(g_sorted_mem + (size_t)tile * kTileNodes);
        for (uint32_t v = threadIdx.x; v < kTileNodes / 2; v += blockDim.x) {
            reinterpret_cast<u64x2 *>(s_cpu)[v] = sc[v];
            reinterpret_cast<u64x2 *>(s_mem)[v] = sm[v];
        }
        __syncthreads();
    }
    const uint32_t lo = blockIdx.y * pods_per_block;
    const uint32_t hi = min(a.p, lo + pods_per_block);
    for (uint32_t pod = lo + threadIdx.x; pod < hi; pod += blockDim.x) {
        if (a.do_fit) {
            const int64_t rc = g_pcpu[pod], rm = g_pmem[pod];
            // two interleaved branch-free binary searches: r = #sorted values < req
            uint32_t lc = 0, lm = 0;
#pragma unroll
            for (uint32_t step = kTileNodes / 2; step >= 1; step >>= 1) {
                const int64_t vc = s_cpu[lc + step - 1], vm = s_mem[lm + step - 1];
                lc += (vc < rc) ? step : 0u;
                lm += (vm < rm) ? step : 0u;
            }
            lc += (s_cpu[lc] < rc) ? 1u : 0u;  // 1023 vs 1024
            lm += (s_mem[lm] < rm) ? 1u : 0u;
            fitw[(size_t)tile * a.p + pod] = (lc >> 5) | ((lc & 31u) << 8) | ((lm >> 5) << 16) | ((lm & 31u) << 24);
        }
        if (tile == 0 && (a.do_sel || a.do_taint)) {
            // row ids: 4 inline label rows + 4 inline taint rows; unused slots read the all-valid row
            uint32_t l0 = L.row_valid, l1 = L.row_valid, l2 = L.row_valid, l3 = L.row_valid;
            uint32_t t0 = L.row_valid, t1 = L.row_valid, t2 = L.row_valid, t3 = L.row_valid;
            uint16_t *ex = ext + (size_t)pod * a.ext_stride;
            if (a.do_sel) {
                uint32_t cnt = 0;
                for (uint32_t k = 0; k < L.nkeys; ++k) {
                    const uint32_t sv = g_psel[(size_t)k * a.p + pod];
                    if (sv != 0u) {
                        const uint32_t row = (sv <= L.lab_max[k]) ? (L.lab_base[k] + sv - 1u) : L.row_zero;
                        l0 = (cnt == 0) ? row : l0;
                        l1 = (cnt == 1) ? row : l1;
                        l2 = (cnt == 2) ? row : l2;
                        l3 = (cnt == 3) ? row : l3;
                        if (cnt >= 3) ex[1 + (cnt - 3)] = (uint16_t)row;  // 4th and later rows, used only when cnt > 4
                        ++cnt;
                    }
                }
                if (cnt > 4) {  // sentinel: ex[0] rows follow in ex[1..]
                    l3 = 0xFFFFu;
                    ex[0] = (uint16_t)(cnt - 3);
                }
            }
            if (a.do_taint) {
                const uint64_t tol = g_ptol ? g_ptol[pod] : 0ull;
                uint16_t *ext_t = ex + (L.nkeys > 3 ? L.nkeys - 3 + 1 : 0);
                for (uint32_t g = 0; g < L.ngroups; ++g) {
                    const uint32_t row = L.row_taint + 16u * g + (uint32_t)((tol >> (4u * g)) & 15ull);
                    t0 = (g == 0) ? row : t0;
                    t1 = (g == 1) ? row : t1;
                    t2 = (g == 2) ? row : t2;
                    t3 = (g == 3) ? row : t3;
                    if (g >= 4) ext_t[g - 4] = (uint16_t)row;
                }
            }
            const uint32_t rows[8] = {l0, l1, l2, l3, t0, t1, t2, t3};
            uint4 r;
            r.x = rows[0] | (rows[1] << 16);
            r.y = rows[2] | (rows[3] << 16);
            r.z = rows[4] | (rows[5] << 16);
            r.w = rows[6] | (rows[7] << 16);
            rec16[pod] = r;
        }
    }
}

// ---- main kernel -------------------------------------------------------------------------------------
template <bool FIT, bool SEL, bool TAINT, bool WANT_FIT>
__global__ __launch_bounds__(1024) void k_eval_indexed(const uint64_t *__restrict__ g_tables, const uint4 *__restrict__ rec16,
                                                        const uint16_t *__restrict__ ext, const uint32_t *__restrict__ fitw,
                                                        uint64_t *__restrict__ out_feas, uint64_t *__restrict__ out_fit,
                                                        const IndexedArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const IndexedLayout &L = a.lay;
    // XCD-aware work mapping: block b runs on XCD b % 8 (observed dispatch order; speed only).
    // All tiles of one pod chunk get the same XCD so seam cache lines meet in one L2.
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, i = b >> 3;
    const uint32_t tile = i % L.tiles;
    const uint32_t chunk = (i / L.tiles) * 8u + xcd;
    if (chunk >= a.chunks) return;

    // ---- stage this tile's bitmap rows into LDS (16-byte vectors) ---------------------------
    if (!(a.debug & 8u)) {
        const u64x2 *src = reinterpret_cast<const u64x2 *>(g_tables + (size_t)tile * L.rows * kTileWords);
        u64x2 *dst = reinterpret_cast<u64x2 *>(smem);
        const uint32_t nvec = L.rows * (kTileWords / 2);
        for (uint32_t v = threadIdx.x; v < nvec; v += blockDim.x) dst[v] = src[v];
    }
    __syncthreads();

    const uint32_t pod_lo = chunk * a.pods_per_chunk;
    const uint32_t pod_hi = min(a.p, pod_lo + a.pods_per_chunk);
    const uint32_t wp = threadIdx.x & 7u;             // word pair inside the tile: words 2wp, 2wp+1
    const uint32_t w0 = tile * kTileWords + 2u * wp;  // first global word of this lane
    const bool has0 = w0 < L.W, has1 = w0 + 1 < L.W;
    if (!has0 || (a.debug & 16u)) return;
    const u64x2 *T = reinterpret_cast<const u64x2 *>(smem) + wp;  // row r -> T[r * 8]
    const uint32_t *fw = fitw + (size_t)tile * a.p;
    const uint32_t pods_per_pass = blockDim.x >> 3;

#pragma unroll 2
    for (uint32_t pod = pod_lo + (threadIdx.x >> 3); pod < pod_hi; pod += pods_per_pass) {
        u64x2 f;
        if (FIT) {
            const uint32_t q = fw[pod];
            const uint32_t ch = q & 255u, cl = (q >> 8) & 255u, mh = (q >> 16) & 255u, ml = q >> 24;
            const u64x2 c = T[(L.row_cpu_hi + ch) * 8u] & (T[(L.row_cpu_hi + ch + 1u) * 8u] | T[(L.row_cpu_lo + cl) * 8u]);
            const u64x2 m = T[(L.row_mem_hi + mh) * 8u] & (T[(L.row_mem_hi + mh + 1u) * 8u] | T[(L.row_mem_lo + ml) * 8u]);
            f = c & m;
        } else {
            f = T[L.row_valid * 8u];
        }
        const size_t o = (size_t)pod * L.W + w0;
        if (WANT_FIT) {
            if (has1) *reinterpret_cast<u64x2_a8 *>(out_fit + o) = f;
            else out_fit[o] = f.x;
        }
        if (SEL || TAINT) {
            const uint4 r = rec16[pod];
            if (SEL) {
                const uint32_t r3 = r.y >> 16;
                if (r3 != 0xFFFFu) {
                    f &= (T[(r.x & 0xFFFFu) * 8u] & T[(r.x >> 16) * 8u]) & (T[(r.y & 0xFFFFu) * 8u] & T[r3 * 8u]);
                } else {  // more than four constrained keys (rare): the rest sits in ext
                    f &= (T[(r.x & 0xFFFFu) * 8u] & T[(r.x >> 16) * 8u]) & T[(r.y & 0xFFFFu) * 8u];
                    const uint16_t *ex = ext + (size_t)pod * a.ext_stride;
                    const uint32_t more = ex[0];
                    for (uint32_t j = 0; j < more; ++j) f &= T[(uint32_t)ex[1 + j] * 8u];
                }
            }
            if (TAINT) {
                f &= (T[(r.z & 0xFFFFu) * 8u] & T[(r.z >> 16) * 8u]) & (T[(r.w & 0xFFFFu) * 8u] & T[(r.w >> 16) * 8u]);
                if (L.ngroups > 4u) {
                    const uint16_t *ext_t = ext + (size_t)pod * a.ext_stride + (L.nkeys > 3 ? L.nkeys - 3 + 1 : 0);
                    for (uint32_t g = 4; g < L.ngroups; ++g) f &= T[(uint32_t)ext_t[g - 4] * 8u];
                }
            }
        }
        if (out_feas && (!(a.debug & 1u) || (f.x == 0x123456789abcdefull && f.y == 7ull))) {
            if (has1) *reinterpret_cast<u64x2_a8 *>(out_feas + o) = f;
            else out_feas[o] = f.x;
        }
    }
}

template <bool FIT, bool SEL, bool TAINT>
inline hipError_t launch_indexed_t(bool want_fit, dim3 grid, uint32_t threads, uint32_t lds, hipStream_t stream, const uint64_t *tables,
                                   const uint4 *rec16, const uint16_t *ext, const uint32_t *fitw, uint64_t *out_feas, uint64_t *out_fit,
                                   const IndexedArgs &a) {
    hipError_t e;
    if (want_fit) {
        e = hipFuncSetAttribute((const void *)k_eval_indexed<FIT, SEL, TAINT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_eval_indexed<FIT, SEL, TAINT, true>), grid, dim3(threads), lds, stream, tables, rec16, ext, fitw, out_feas, out_fit, a);
    } else {
        e = hipFuncSetAttribute((const void *)k_eval_indexed<FIT, SEL, TAINT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_eval_indexed<FIT, SEL, TAINT, false>), grid, dim3(threads), lds, stream, tables, rec16, ext, fitw, out_feas, out_fit, a);
    }
    return hipGetLastError();
}

// Launch geometry of the main kernel: tiles x chunks blocks (chunks rounded up to a multiple of 8 for the XCD map).
inline hipError_t run_indexed(const IndexedSnapshot &s, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                              const uint64_t *ptol, uint32_t flags, uint64_t *out_feas, uint64_t *out_fit, uint8_t *scratch,
                              hipStream_t stream, uint32_t debug = 0) {
    const IndexedLayout &l = s.lay;
    IndexedArgs a{};
    a.lay = l;
    a.p = p;
    a.do_fit = (flags & KSCHED_FIT) ? 1u : 0u;
    a.do_sel = ((flags & KSCHED_SEL) && psel && l.nkeys) ? 1u : 0u;
    a.do_taint = ((flags & KSCHED_TAINT) && l.ngroups) ? 1u : 0u;
    a.ext_stride = indexed_ext_stride(l);
    a.debug = debug;
    // carve the scratch buffer
    uint4 *rec16 = reinterpret_cast<uint4 *>(scratch);
    uint16_t *ext = reinterpret_cast<uint16_t *>(scratch + (size_t)p * 16u);
    uint32_t *fitw = reinterpret_cast<uint32_t *>(scratch + (size_t)p * 16u + (((size_t)p * a.ext_stride * 2u + 15u) & ~(size_t)15u));

    // ---- pre-pass ---------------------------------------------------------------------------
    if (a.do_fit || a.do_sel || a.do_taint) {
        const uint32_t ptiles = a.do_fit ? l.tiles : 1u;
        uint32_t ychunks = std::max(1u, std::min((p + 255u) / 256u, (2048u + ptiles - 1u) / ptiles));
        const uint32_t ppb = (p + ychunks - 1u) / ychunks;
        ychunks = (p + ppb - 1u) / ppb;
        hipLaunchKernelGGL(k_index_pods, dim3(ptiles, ychunks), dim3(256), 0, stream, s.d_sorted_cpu, s.d_sorted_mem, pcpu, pmem, psel,
                           ptol, rec16, ext, fitw, a, ppb);
    }

    // ---- main kernel ------------------------------------------------------------------------
    const uint32_t threads = 1024;
    const uint32_t lds = indexed_lds_bytes(l);
    const uint32_t blocks_per_cu = std::max(1u, std::min(kLdsBudget / std::max(lds, 1u), 2048u / threads));
    const uint32_t slots = 256u * blocks_per_cu;
    // chunks: fill the chip a whole number of times; never less than one pass (128 pods) per block
    const uint32_t max_chunks = std::max(1u, (p + 127u) / 128u);
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
    const int sel = a.do_sel ? 1 : 0, tnt = a.do_taint ? 1 : 0, fit = a.do_fit ? 1 : 0;
    switch (fit * 4 + sel * 2 + tnt) {
        case 0: return launch_indexed_t<false, false, false>(want_fit, grid, threads, lds, stream, s.d_tables, rec16, ext, fitw, out_feas, out_fit, a);
        case 1: return launch_indexed_t<false, false, true>(want_fit, grid, threads, lds, stream, s.d_tables, rec16, ext, fitw, out_feas, out_fit, a);
        case 2: return launch_indexed_t<false, true, false>(want_fit, grid, threads, lds, stream, s.d_tables, rec16, ext, fitw, out_feas, out_fit, a);
        case 3: return launch_indexed_t<false, true, true>(want_fit, grid, threads, lds, stream, s.d_tables, rec16, ext, fitw, out_feas, out_fit, a);
        case 4: return launch_indexed_t<true, false, false>(want_fit, grid, threads, lds, stream, s.d_tables, rec16, ext, fitw, out_feas, out_fit, a);
        case 5: return launch_indexed_t<true, false, true>(want_fit, grid, threads, lds, stream, s.d_tables, rec16, ext, fitw, out_feas, out_fit, a);
        case 6: return launch_indexed_t<true, true, false>(want_fit, grid, threads, lds, stream, s.d_tables, rec16, ext, fitw, out_feas, out_fit, a);
        default: return launch_indexed_t<true, true, true>(want_fit, grid, threads, lds, stream, s.d_tables, rec16, ext, fitw, out_feas, out_fit, a);
    }
}

}  // namespace ksched
