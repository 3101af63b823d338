// tile_index.hpp -- the per-tile bitmap index of a node snapshot that the fused mask kernel (kernels_fused.hpp) evaluates
// against: its layout, and its SPEC as host code (index_tile_fit, indexed_build_host).  The product builds it on the device
// (kernels_build.hpp); the host functions here define what those kernels must produce and serve the host-only arithmetic test.
//
// Why an index: the output (P x ceil(N/64) words) is the HBM traffic; deciding every bit with its
// own compare (kernels_direct.hpp) costs >= 4 VALU issues per output word per wave64 and lands an
// order of magnitude under the HBM write roofline.  Here every predicate is turned into an AND
// of precomputed node bitmaps, so one VALU op decides 32 (pod, node) pairs per lane.
//
// Per tile of kTileNodes = 1024 nodes = 16 mask words; a tile is 8 sub-tiles of 128 nodes (one
// 16-byte chunk of a row = what one lane of the kernel's phase 2 owns):
//   * fit   -- src/predicates.rs:42  req <= avail.  Sort the tile's avail values; node n gets its
//              position pos[n] in that order (ties broken by node index, so pos is a
//              permutation).  For a pod, r = #values < req (lower bound; descent of the tile's
//              breadth-first search tree in LDS); then
//                  req <= avail[n]  <=>  pos[n] >= r,            exactly, for any int64 inputs.
//              pos >= r is decided per sub-tile with ONE bitmap read per resource:
//                  cnt[r][s] = #{nodes of sub-tile s with pos < r}          (a byte, 0..128)
//                  lr[n]     = #{nodes of n's sub-tile with pos < pos[n]}   (n's local rank)
//                  pos[n] >= r  <=>  lr[n] >= cnt[r][sub-tile of n]
//              because the nodes of a sub-tile with pos < r are exactly its cnt[r][s] lowest-ranked
//              ones.  Row c (0..128) of the resource's block of the table is {n : lr[n] >= c}; the
//              lane owning sub-tile s reads chunk s of row cnt[r][s].  cnt is a table of 1025
//              8-byte entries per resource (one byte per sub-tile), read once per pod and tile.
//   * sel   -- src/predicates.rs:45-61.  One bitmap row per (key, value id): nodes carrying that
//              value, plus one all-zero row after the last id of every key.  A pod ANDs the rows of
//              the keys it constrains; unconstrained keys read the all-valid row; ids no node
//              carries (KSCHED_SEL_NEVER, unknown ids) clamp to the key's all-zero row.
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
constexpr int kSubNodes = 128;               // nodes per sub-tile = bits per 16-byte chunk
constexpr int kSubTiles = kTileNodes / kSubNodes;  // 8
constexpr int kFitRows = kSubNodes + 1;      // rows {lr >= c}, c = 0..128, per resource
constexpr int kCntEntries = kTileNodes + 2;  // cnt[r], r = 0..1024, padded to a multiple of 16 bytes
constexpr int kIdxMaxKeys = 32;
constexpr int kIdxMaxGroups = 16;            // 64 taint bits / 4
constexpr uint32_t kLdsBudget = 160u * 1024u;
// per-tile "aux" block (global image = LDS image): [tree cpu][tree mem][cnt cpu][cnt mem], 8-byte words
constexpr uint32_t kAuxTreeWords = kTileNodes;
constexpr uint32_t kAuxWords = 2u * kAuxTreeWords + 2u * kCntEntries;  // 4100 words = 32800 bytes
// LDS the kernel needs besides the bitmap rows: the aux block and up to 40 bytes of per-pod records for
// 16 waves x 64 pods (kernels_fused.hpp: fused_lds_bytes)
constexpr uint32_t kLdsNonRowBytesMax = kAuxWords * 8u + 1024u * 40u;
// High-cardinality label keys (e.g. kubernetes.io/hostname: one value per node) do not get one bitmap row per value -- 5000
// values x 128 B would not fit any LDS.  Such a key is kept per tile as a LIST instead: the tile's nodes sorted by the key's
// value id (u32 ids, then u16 node numbers).  A pod that constrains the key binary-searches its id (phase 1) and gets a range
// of at most a few nodes, whose bits phase 2 sets one by one.  Up to kMaxListKeys (2) keys per snapshot (each costs a pipelined operand register in the kernel); rows are given to the
// keys with the fewest values first.
constexpr uint32_t kMaxListKeys = 2;
constexpr uint32_t kListBytes = kTileNodes * 4u + kTileNodes * 2u;  // per (tile, list key): 6 KiB
constexpr uint32_t kListRecBytes = 8u;                              // per pod of a round: kMaxListKeys x (first entry u16 | count u16)
constexpr uint32_t kLabList = 0xFFFFFFFFu;                          // lab_base of a list key

struct IndexedLayout {
    uint32_t n, W, tiles, rows, nkeys, ngroups;
    uint32_t row_zero, row_valid;
    uint32_t row_cpu, row_mem;          // first of the kFitRows rows {lr >= c} of each resource
    uint32_t row_taint;                 // + 16 * group + subset
    uint32_t lab_base[kIdxMaxKeys];     // row of value id 1 of key k; row lab_base + lab_max is the key's all-zero row; kLabList = list key
    uint32_t lab_max[kIdxMaxKeys];      // largest id some node carries
    uint32_t nlist;                     // keys kept as sorted lists instead of rows
    uint32_t list_col[kMaxListKeys];    // their label columns
};

struct IndexedSnapshot {
    bool built = false;
    IndexedLayout lay{};
    uint64_t *d_tables = nullptr;     // [tiles][rows][16]
    uint64_t *d_aux = nullptr;        // [tiles][kAuxWords]: search trees (eytzinger_from_sorted, padded with INT64_MAX) + cnt tables
    uint32_t *d_lab_meta = nullptr;   // lab_base[32], lab_max[32], then 8 zero words
    uint8_t *d_list = nullptr;        // [tiles][nlist][kListBytes]: per list key the tile's value ids ascending (u32[1024]), then their nodes (u16[1024])
    size_t aux_cap = 0, tables_cap = 0, list_cap = 0;
};

inline void indexed_release(IndexedSnapshot &s) {
    if (s.d_aux) (void)hipFree(s.d_aux);
    if (s.d_tables) (void)hipFree(s.d_tables);
    if (s.d_lab_meta) (void)hipFree(s.d_lab_meta);
    if (s.d_list) (void)hipFree(s.d_list);
    s = IndexedSnapshot{};
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

// Fit part of one tile: the rows {lr >= c} of both resources, the two search trees and the two cnt tables, from the
// tile's current `available` values (src/predicates.rs:27-38).  Rewrites them from scratch (the rows are contiguous:
// [row_cpu, row_cpu + 2 * kFitRows)), so it serves both the full build and ksched_update_nodes.
//   T   : the tile's rows  [rows][16]
//   aux : the tile's aux block [kAuxWords]
inline void index_tile_fit(const IndexedLayout &l, uint32_t t, const int64_t *cpu, const int64_t *mem, uint64_t *T, uint64_t *aux) {
    const uint32_t base = t * kTileNodes;
    const uint32_t m = std::min<uint32_t>(kTileNodes, l.n - base);
    std::fill(T + (size_t)l.row_cpu * kTileWords, T + (size_t)(l.row_cpu + 2 * kFitRows) * kTileWords, 0ull);
    auto setbit = [&](uint32_t row, uint32_t local) { T[(size_t)row * kTileWords + (local >> 6)] |= 1ull << (local & 63u); };
    uint32_t ord[kTileNodes];
    int64_t sorted[kTileNodes];
    for (int res = 0; res < 2; ++res) {
        const int64_t *v = res == 0 ? cpu : mem;
        const uint32_t row0 = res == 0 ? l.row_cpu : l.row_mem;
        int64_t *tree = reinterpret_cast<int64_t *>(aux) + (size_t)res * kAuxTreeWords;
        uint64_t *cnt = aux + 2u * kAuxTreeWords + (size_t)res * kCntEntries;
        std::fill(sorted, sorted + kTileNodes, INT64_MAX);
        std::iota(ord, ord + m, 0u);
        std::stable_sort(ord, ord + m, [&](uint32_t a, uint32_t b) { return v[base + a] < v[base + b]; });
        // walk the nodes in position order: `seen[s]` = nodes of sub-tile s met so far = the local rank of the next one,
        // and, packed one byte per sub-tile, exactly cnt[pos]
        uint64_t packed = 0;  // byte s = seen[s]; a count of 128 needs the whole byte, never more
        uint32_t seen[kSubTiles] = {};
        for (uint32_t pos = 0; pos < m; ++pos) {
            cnt[pos] = packed;
            const uint32_t local = ord[pos], s = local / kSubNodes;
            sorted[pos] = v[base + local];
            const uint32_t lr = seen[s]++;
            for (uint32_t c = 0; c <= lr; ++c) setbit(row0 + c, local);  // rows {lr >= c}
            packed += 1ull << (8u * s);
        }
        for (uint32_t r = m; r < (uint32_t)kCntEntries; ++r) cnt[r] = packed;  // r = m..1024: every node is below the request
        eytzinger_from_sorted(sorted, tree);
    }
}

// Decide the row layout of a snapshot: which rows exist and where.  `lab_max[k]` = largest value id some node carries for
// key k, `all_taints` = OR of every node's taint bits.  Returns false (with the reason in *why) when the snapshot is outside
// what the fused kernel supports; the caller then uses the direct kernel.
inline bool indexed_plan(IndexedLayout &l, uint32_t n, uint32_t nkeys, const uint32_t *lab_max, uint64_t all_taints, const char **why) {
    if (n == 0) {
        *why = "no nodes";
        return false;
    }
    if (nkeys > kIdxMaxKeys) {
        *why = "more label keys than the bitmap index holds";
        return false;
    }
    l = IndexedLayout{};
    l.n = n;
    l.W = (n + 63u) / 64u;
    l.tiles = (n + kTileNodes - 1) / kTileNodes;
    l.nkeys = nkeys;
    l.ngroups = all_taints ? (uint32_t)((64 - __builtin_clzll(all_taints)) + 3) / 4 : 0;
    // Row order: [zero, valid, taint rows, label rows, fit rows of cpu, fit rows of memory].  The rows a pod names by
    // record (labels, taints, valid, zero) come first so that their BYTE offsets fit 16 bits (kernels_fused.hpp).
    uint32_t r = 0;
    l.row_zero = r++;
    l.row_valid = r++;
    l.row_taint = r; r += 16 * l.ngroups;
    for (uint32_t k = 0; k < nkeys; ++k) l.lab_max[k] = lab_max[k];
    // Rows go to the keys with the fewest values; keys that do not fit become lists (largest first), at most kMaxListKeys.
    // All rows of a tile must fit in LDS next to the aux block, the lists and the per-pod records, and the named rows (the ones
    // a pod's record addresses with 16 bits) below 64 KiB.
    bool is_list[kIdxMaxKeys] = {};
    for (;;) {
        uint64_t label_rows = 0;
        for (uint32_t k = 0; k < nkeys; ++k)
            if (!is_list[k]) label_rows += (uint64_t)lab_max[k] + 1u;  // + the key's all-zero row (ids above the max clamp to it)
        const uint64_t lds = (r + label_rows + 2u * kFitRows) * 128u + kLdsNonRowBytesMax + (uint64_t)l.nlist * kListBytes +
                             (l.nlist ? 1024u * kListRecBytes : 0u);
        if (lds <= kLdsBudget && (r + label_rows) * 128u <= 65536u) break;
        if (l.nlist == kMaxListKeys) {
            *why = "more than two high-cardinality label keys: their (key, value) rows exceed the LDS budget of the fused kernel";
            return false;
        }
        uint32_t big = nkeys;
        for (uint32_t k = 0; k < nkeys; ++k)
            if (!is_list[k] && (big == nkeys || lab_max[k] > lab_max[big])) big = k;
        if (big == nkeys) {
            *why = "the fit rows alone exceed the LDS budget";
            return false;
        }
        is_list[big] = true;
        ++l.nlist;
    }
    {
        uint32_t j = 0;
        for (uint32_t k = 0; k < nkeys; ++k)
            if (is_list[k]) l.list_col[j++] = k;  // ascending column order
    }
    for (uint32_t k = 0; k < nkeys; ++k) {
        if (is_list[k]) {
            l.lab_base[k] = kLabList;
            continue;
        }
        l.lab_base[k] = r;
        r += l.lab_max[k] + 1u;
    }
    l.row_cpu = r; r += kFitRows;
    l.row_mem = r; r += kFitRows;
    l.rows = r;
    return true;
}

inline hipError_t indexed_reserve(IndexedSnapshot &s, const IndexedLayout &l) {
    hipError_t e;
    const size_t aux_words = (size_t)l.tiles * kAuxWords, tab_words = (size_t)l.tiles * l.rows * kTileWords;
    if (aux_words > s.aux_cap) {
        if (s.d_aux) (void)hipFree(s.d_aux);
        s.d_aux = nullptr;
        s.aux_cap = 0;
        if ((e = hipMalloc((void **)&s.d_aux, aux_words * 8)) != hipSuccess) return e;
        s.aux_cap = aux_words;
    }
    if (tab_words > s.tables_cap) {
        if (s.d_tables) (void)hipFree(s.d_tables);
        s.d_tables = nullptr;
        s.tables_cap = 0;
        if ((e = hipMalloc((void **)&s.d_tables, tab_words * 8)) != hipSuccess) return e;
        s.tables_cap = tab_words;
    }
    if (!s.d_lab_meta && (e = hipMalloc((void **)&s.d_lab_meta, 72 * sizeof(uint32_t))) != hipSuccess) return e;
    const size_t list_bytes = (size_t)l.tiles * l.nlist * kListBytes;
    if (list_bytes > s.list_cap) {
        if (s.d_list) (void)hipFree(s.d_list);
        s.d_list = nullptr;
        s.list_cap = 0;
        if ((e = hipMalloc((void **)&s.d_list, list_bytes)) != hipSuccess) return e;
        s.list_cap = list_bytes;
    }
    return hipSuccess;
}

inline void indexed_meta(const IndexedLayout &l, uint32_t meta[72]) {
    for (int k = 0; k < 72; ++k) meta[k] = 0;
    for (int k = 0; k < 32; ++k) {
        meta[k] = l.lab_base[k];
        meta[32 + k] = l.lab_max[k];
    }
}

// The SPEC of the index, on the host (the device build of kernels_build.hpp must reproduce these tables bit for bit;
// KSCHED_OPT_INDEX_BUILD = 1 selects this path, tests/test_gpu_index_build.py compares the two).  Fills the host images
// s.h_tables / s.h_aux for the layout `l` and uploads them.
inline hipError_t indexed_build_host(IndexedSnapshot &s, const IndexedLayout &l, const int64_t *cpu, const int64_t *mem, const uint32_t *lab,
                                     const uint64_t *taints, hipStream_t stream) {
    const uint32_t n = l.n, nkeys = l.nkeys;
    const size_t tile_words = (size_t)l.rows * kTileWords;
    std::vector<uint64_t> tab((size_t)l.tiles * tile_words, 0ull);
    std::vector<uint64_t> aux((size_t)l.tiles * kAuxWords, 0ull);
    std::vector<uint8_t> lists((size_t)l.tiles * l.nlist * kListBytes, 0);
    for (uint32_t t = 0; t < l.tiles; ++t) {
        const uint32_t base = t * kTileNodes;
        const uint32_t m = std::min<uint32_t>(kTileNodes, n - base);
        uint64_t *T = tab.data() + (size_t)t * tile_words;
        auto setbit = [&](uint32_t row, uint32_t local) { T[(size_t)row * kTileWords + (local >> 6)] |= 1ull << (local & 63u); };
        for (uint32_t i = 0; i < m; ++i) setbit(l.row_valid, i);
        index_tile_fit(l, t, cpu, mem, T, aux.data() + (size_t)t * kAuxWords);
        // labels
        for (uint32_t k = 0; k < nkeys; ++k) {
            if (l.lab_base[k] == kLabList) continue;
            for (uint32_t i = 0; i < m; ++i) {
                const uint32_t id = lab[(size_t)k * n + base + i];
                if (id) setbit(l.lab_base[k] + id - 1, i);
            }
        }
        // list keys: the tile's 1024 slots (padding carries id 0 = absent) ascending by (id, node)
        for (uint32_t j = 0; j < l.nlist; ++j) {
            const uint32_t k = l.list_col[j];
            uint32_t ord[kTileNodes], val[kTileNodes];
            for (uint32_t i = 0; i < (uint32_t)kTileNodes; ++i) {
                ord[i] = i;
                val[i] = i < m ? lab[(size_t)k * n + base + i] : 0u;
            }
            std::stable_sort(ord, ord + kTileNodes, [&](uint32_t x, uint32_t y) { return val[x] < val[y]; });
            uint8_t *L = lists.data() + ((size_t)t * l.nlist + j) * kListBytes;
            uint32_t *vals = reinterpret_cast<uint32_t *>(L);
            uint16_t *nodes = reinterpret_cast<uint16_t *>(L + kTileNodes * 4u);
            for (uint32_t i = 0; i < (uint32_t)kTileNodes; ++i) {
                vals[i] = val[ord[i]];
                nodes[i] = (uint16_t)ord[i];
            }
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
    if ((e = hipMemcpyAsync(s.d_aux, aux.data(), aux.size() * 8, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(s.d_tables, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if (!lists.empty() && (e = hipMemcpyAsync(s.d_list, lists.data(), lists.size(), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;  // the vectors go out of scope
    return hipSuccess;
}

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

}  // namespace ksched
