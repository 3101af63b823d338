// tile_index.hpp -- the per-tile bitmap index of a node snapshot (built once per ksched_set_nodes)
// that the fused mask kernel (kernels_fused.hpp) evaluates against.
//
// Why an index: the output (P x ceil(N/64) words) is the HBM traffic; deciding every bit with its
// own compare (kernels_direct.hpp) costs >= 4 VALU issues per output word per wave64 and lands an
// order of magnitude under the HBM write roofline.  Here every predicate is turned into an AND
// of precomputed node bitmaps, so one VALU op decides 32 (pod, node) pairs per lane.
//
// Per tile of kTileNodes = 1024 nodes = 16 mask words:
//   * fit   -- src/predicates.rs:42  req <= avail.  Sort the tile's avail values; node n gets its
//              position pos[n] in that order (ties broken by node index, so pos is a
//              permutation).  For a pod, r = #values < req (lower bound; descent of the tile's
//              breadth-first search tree in LDS);
//              then  req <= avail[n]  <=>  pos[n] >= r, exactly, for any int64 inputs.
//              pos >= r is evaluated two-level, pos = 32*hi + lo, r = 32*rh + rl:
//                  pos >= r  <=>  hi > rh  ||  (hi == rh && lo >= rl)
//                            <=>  GEH[rh] & (GEH[rh+1] | GEL[rl])          (GEH[h] = {n: hi >= h})
//              3 bitmap rows per resource instead of 1025 rows for a one-level table.
//   * sel   -- src/predicates.rs:45-61.  One bitmap row per (key, value id): nodes carrying that
//              value.  A pod ANDs the rows of the keys it constrains; unconstrained keys read the
//              all-valid row; KSCHED_SEL_NEVER / unknown ids hit the all-zero row.
//   * taint -- (taints[n] & ~tol[p]) == 0.  Per 4-bit group g of taint bits and per tolerated
//              subset s of that group: row {n : taints_g[n] subset of s}; a pod ANDs one row per
//              group.
// All rows are 16 words (128 B); padding bits (node >= N) are zero in every row, so they are
// zero in every result.
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
    int64_t *d_sorted_cpu = nullptr;  // [tiles][1024] search trees (eytzinger_from_sorted), padded with INT64_MAX
    int64_t *d_sorted_mem = nullptr;
    uint64_t *d_tables = nullptr;     // [tiles][rows][16]
    uint32_t *d_lab_meta = nullptr;   // lab_base[32], lab_max[32], then 8 zero words
    size_t sorted_cap = 0, tables_cap = 0;
    // host images of the three device arrays, kept so that ksched_update_nodes can rebuild the fit part of single tiles
    std::vector<uint64_t> h_tables;
    std::vector<int64_t> h_sorted_cpu, h_sorted_mem;
};

inline void indexed_release(IndexedSnapshot &s) {
    if (s.d_sorted_cpu) (void)hipFree(s.d_sorted_cpu);
    if (s.d_sorted_mem) (void)hipFree(s.d_sorted_mem);
    if (s.d_tables) (void)hipFree(s.d_tables);
    if (s.d_lab_meta) (void)hipFree(s.d_lab_meta);
    s = IndexedSnapshot{};  // also drops the host images
}

inline uint32_t indexed_lds_bytes(const IndexedLayout &l) { return l.rows * 128u; }

// Breadth-first (Eytzinger) image of a sorted array of kTileNodes values: slot k in [1, 1024) is node k of
// the perfect binary search tree over sorted[0..1022] (node k at level L = floor(log2 k), j = k - 2^L, holds
// sorted[(2j + 1) * 2^(9 - L) - 1]); slot 0 holds sorted[1023].  The kernel descends from k = 1 with
// k = 2k + (tree[k] < req); after 10 levels k - 1024 is the number of values among sorted[0..1022] below req.
inline void eytzinger_from_sorted(const int64_t *sorted, int64_t *tree) {
    tree[0] = sorted[kTileNodes - 1];
    for (uint32_t level = 0; level < 10; ++level)
        for (uint32_t j = 0; j < (1u << level); ++j) tree[(1u << level) + j] = sorted[((2u * j + 1u) << (9u - level)) - 1u];
}
// Fit part of one tile: the rows GEH/GEL of both resources and the two search trees, from the tile's current
// `available` values (src/predicates.rs:27-38).  Rewrites those rows from scratch (they are contiguous:
// [row_cpu_hi, row_cpu_hi + 2 * (kFitHi + kFitLo))), so it serves both the full build and ksched_update_nodes.
inline void index_tile_fit(const IndexedLayout &l, uint32_t t, const int64_t *cpu, const int64_t *mem, uint64_t *T, int64_t *tree_cpu,
                           int64_t *tree_mem) {
    const uint32_t base = t * kTileNodes;
    const uint32_t m = std::min<uint32_t>(kTileNodes, l.n - base);
    std::fill(T + (size_t)l.row_cpu_hi * kTileWords, T + (size_t)(l.row_cpu_hi + 2 * (kFitHi + kFitLo)) * kTileWords, 0ull);
    auto setbit = [&](uint32_t row, uint32_t local) { T[(size_t)row * kTileWords + (local >> 6)] |= 1ull << (local & 63u); };
    uint32_t ord[kTileNodes];
    int64_t sorted[kTileNodes];
    for (int res = 0; res < 2; ++res) {
        const int64_t *v = res == 0 ? cpu : mem;
        const uint32_t row_hi = res == 0 ? l.row_cpu_hi : l.row_mem_hi;
        const uint32_t row_lo = res == 0 ? l.row_cpu_lo : l.row_mem_lo;
        std::fill(sorted, sorted + kTileNodes, INT64_MAX);
        std::iota(ord, ord + m, 0u);
        std::stable_sort(ord, ord + m, [&](uint32_t a, uint32_t b) { return v[base + a] < v[base + b]; });
        for (uint32_t pos = 0; pos < m; ++pos) {
            const uint32_t local = ord[pos];
            sorted[pos] = v[base + local];
            const uint32_t hi = pos >> 5, lo = pos & 31u;
            for (uint32_t h = 0; h <= hi; ++h) setbit(row_hi + h, local);  // GEH[h] = {hi >= h}
            for (uint32_t q = 0; q <= lo; ++q) setbit(row_lo + q, local);  // GEL[q] = {lo >= q}
        }
        eytzinger_from_sorted(sorted, res == 0 ? tree_cpu : tree_mem);
    }
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
    for (uint32_t t = 0; t < l.tiles; ++t) {
        const uint32_t base = t * kTileNodes;
        const uint32_t m = std::min<uint32_t>(kTileNodes, n - base);
        uint64_t *T = tab.data() + (size_t)t * tile_words;
        auto setbit = [&](uint32_t row, uint32_t local) { T[(size_t)row * kTileWords + (local >> 6)] |= 1ull << (local & 63u); };
        for (uint32_t i = 0; i < m; ++i) setbit(l.row_valid, i);
        index_tile_fit(l, t, cpu, mem, T, scpu.data() + (size_t)t * kTileNodes, smem.data() + (size_t)t * kTileNodes);
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
    if (!s.d_lab_meta && (e = hipMalloc((void **)&s.d_lab_meta, 72 * sizeof(uint32_t))) != hipSuccess) return e;
    {
        uint32_t meta[72] = {};
        for (int k = 0; k < 32; ++k) {
            meta[k] = l.lab_base[k];
            meta[32 + k] = l.lab_max[k];
        }
        if ((e = hipMemcpy(s.d_lab_meta, meta, sizeof meta, hipMemcpyHostToDevice)) != hipSuccess) return e;
    }
    s.h_tables = std::move(tab);
    s.h_sorted_cpu = std::move(scpu);
    s.h_sorted_mem = std::move(smem);
    s.lay = l;
    s.built = true;
    return hipSuccess;
}

// ksched_update_nodes: `available` changed on some nodes of tile t -> rebuild that tile's fit rows and search
// trees on the host image and re-upload just those (label and taint rows are untouched).
inline hipError_t indexed_update_tile(IndexedSnapshot &s, uint32_t t, const int64_t *cpu, const int64_t *mem) {
    const IndexedLayout &l = s.lay;
    const size_t tile_words = (size_t)l.rows * kTileWords;
    uint64_t *T = s.h_tables.data() + (size_t)t * tile_words;
    int64_t *tc = s.h_sorted_cpu.data() + (size_t)t * kTileNodes, *tm = s.h_sorted_mem.data() + (size_t)t * kTileNodes;
    index_tile_fit(l, t, cpu, mem, T, tc, tm);
    hipError_t e;
    const size_t fit_off = (size_t)l.row_cpu_hi * kTileWords, fit_words = (size_t)2 * (kFitHi + kFitLo) * kTileWords;
    if ((e = hipMemcpy(s.d_tables + (size_t)t * tile_words + fit_off, T + fit_off, fit_words * 8, hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(s.d_sorted_cpu + (size_t)t * kTileNodes, tc, kTileNodes * 8, hipMemcpyHostToDevice)) != hipSuccess) return e;
    return hipMemcpy(s.d_sorted_mem + (size_t)t * kTileNodes, tm, kTileNodes * 8, hipMemcpyHostToDevice);
}

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

}  // namespace ksched
