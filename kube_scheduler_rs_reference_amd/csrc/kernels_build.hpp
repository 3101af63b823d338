// kernels_build.hpp -- the snapshot builder on the device (the hot half): the per-tile bitmap index
// of tile_index.hpp and the best-fit structures, built by HIP kernels from the node columns already resident in HBM.
//
// The reference recomputes `available` per evaluation (src/predicates.rs:21-38); here a snapshot precedes every batch, so
// after the mask kernel (26 us per 100k-pod batch) the snapshot build is what a real batch loop waits for.  On the host it cost
// 2.7-17.9 ms per ksched_set_nodes and 0.7-7.2 ms per ksched_update_nodes (profiles/r01_h5_host_costs_snapshot_calls.txt); these
// kernels make it a few launches.  tile_index.hpp's host functions (index_tile_fit, eytzinger_from_sorted) remain the SPEC: the
// device build must produce bit-identical tables (tests/test_gpu_index_build.py compares checksums and results).
//
//   k_build_tile_fit    grid (tiles | dirty tiles, 2 resources) x 1024 threads: thread = node of the tile.
//                         sort  = bitonic network over LDS on (value, node) pairs: the tile's positions
//                         tree  = Eytzinger image of the sorted values                                (search tree of phase 1)
//                         cnt[r]= per-sub-tile counts of nodes with pos < r, one byte each = exclusive prefix sum over positions
//                         lr    = rank of the node inside its 128-node sub-tile = its sub-tile's byte of that prefix
//                         rows  = {n : lr[n] >= c}, c = 0..128: one wave ballot per (row, word)
//   k_build_tile_named  grid tiles x 1024: valid row, taint subset rows (ballots), label (key, value) rows (LDS atomic OR)
//   k_patch_nodes       ksched_update_nodes: scatter the new `available` values into the columns
//   k_bf_*              best-fit order and its row bitmaps (built lazily, on the first PICK_BESTFIT after a change)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_direct.hpp"
#include "tile_index.hpp"

namespace ksched {

struct BuildFitArgs {
    const int64_t *ncpu, *nmem;  // node columns [n]
    uint64_t *tables, *aux;      // IndexedSnapshot::d_tables, d_aux
    const uint32_t *tile_list;   // tiles to (re)build, or nullptr = tile blockIdx.x
    uint32_t n, rows, row_cpu;   // layout: rows per tile, first fit row of cpu (memory's follow kFitRows later)
};

// Sort the tile's kTileNodes (key, node) pairs ascending by (key, node): a bitonic network, one element per thread (1024
// threads).  On return thread i holds the element at POSITION i (kv, ki), and s_key / s_idx hold the sorted keys / their nodes by
// position.  55 compare-exchange stages; the 45 whose partners are lanes of the same wave exchange through lane shuffles (no
// LDS, no barrier), the 10 cross-wave ones through LDS between block barriers.  (Ranking by counting -- 1024 broadcast compares
// per thread -- was measured at 54 us per tile: 8k VALU instructions per thread on one CU; the all-LDS network at 22 us.)
__device__ __forceinline__ int64_t shfl_xor_key(int64_t x, uint32_t j) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, (int)j, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)((uint64_t)x >> 32), (int)j, 64);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint32_t shfl_xor_key(uint32_t x, uint32_t j) { return (uint32_t)__shfl_xor((int)x, (int)j, 64); }

template <class K>
__device__ __forceinline__ void bitonic_sort_tile(K &kv, uint32_t &ki, K *s_key, uint16_t *s_idx, uint32_t i) {
    for (uint32_t k = 2; k <= (uint32_t)kTileNodes; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            K pv;
            uint32_t pi;
            if (j >= 64u) {  // the partner sits in another wave: through LDS, between block barriers
                s_key[i] = kv;
                s_idx[i] = (uint16_t)ki;
                __syncthreads();
                pv = s_key[i ^ j];
                pi = s_idx[i ^ j];
                __syncthreads();  // everybody has read before anybody writes its slot again
            } else {          // a lane of this wave: registers only (45 of the 55 stages)
                pv = shfl_xor_key(kv, j);
                pi = shfl_xor_key(ki, j);
            }
            const bool ascending = (i & k) == 0u, lower = (i & j) == 0u;
            const bool p_less = pv < kv || (pv == kv && pi < ki);
            const bool take = (lower == ascending) ? p_less : !p_less;  // the lower slot of an ascending pair keeps the smaller element
            kv = take ? pv : kv;
            ki = take ? pi : ki;
        }
    }
    s_key[i] = kv;
    s_idx[i] = (uint16_t)ki;
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_build_tile_fit(const BuildFitArgs a) {
    kernarg_warm<sizeof(BuildFitArgs)>();
    __shared__ int64_t s_key[kTileNodes];  // bitonic network: value; afterwards the sorted values (position order)
    __shared__ uint16_t s_idx[kTileNodes];  // ... node of the element; afterwards s_lr: local rank by node
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_rows[kFitRows * kTileWords];
    const uint32_t tile = a.tile_list ? a.tile_list[blockIdx.x] : blockIdx.x;
    const uint32_t res = blockIdx.y;
    const int64_t *col = res == 0 ? a.ncpu : a.nmem;
    const uint32_t base = tile * kTileNodes;
    const uint32_t m = min((uint32_t)kTileNodes, a.n - base);
    const uint32_t i = threadIdx.x, lane = i & 63u, wave = i >> 6;
    // sort the tile's (value, node) pairs: ties go by node index, so the order is total and the positions are a permutation
    // (tile_index.hpp); padding (node >= m) carries INT64_MAX and the largest indices: it sorts last
    int64_t kv = (i < m) ? col[base + i] : INT64_MAX;
    uint32_t ki = i;
    bitonic_sort_tile(kv, ki, s_key, s_idx, i);
    // thread i now holds the element at POSITION i: value kv, node ki; s_key is the sorted column
    uint64_t *aux = a.aux + (size_t)tile * kAuxWords;
    // search tree (eytzinger_from_sorted): slot k of level L = floor(log2 k), j = k - 2^L holds sorted[(2j + 1) * 2^(9 - L) - 1]
    {
        int64_t t;
        if (i == 0) {
            t = s_key[kTileNodes - 1];
        } else {
            const uint32_t L = 31u - (uint32_t)__builtin_clz(i), j = i - (1u << L);
            t = s_key[((2u * j + 1u) << (9u - L)) - 1u];
        }
        reinterpret_cast<int64_t *>(aux)[(size_t)res * kAuxTreeWords + i] = t;
    }
    // cnt[r] = per-sub-tile counts of nodes at positions < r, one byte each = exclusive prefix sum, over positions, of what each
    // position adds (bytes never carry: a sub-tile holds at most 128 nodes).  The same prefix gives the local rank of the node at
    // position r: the number of nodes of ITS sub-tile at earlier positions.
    {
        const bool live = ki < m;
        const uint32_t sub = ki / kSubNodes;
        const uint64_t x = live ? (1ull << (8u * sub)) : 0ull;
        uint64_t incl = x;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), d, 64);
            if (lane >= d) incl += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();  // also: every thread has read its tree entry from s_key / s_idx is no longer needed as the network's
        uint64_t before = 0, total = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            const uint64_t t = s_wave[w];
            before += (w < wave) ? t : 0ull;
            total += t;
        }
        const uint64_t excl = before + incl - x;
        uint64_t *cnt = aux + 2u * kAuxTreeWords + (size_t)res * kCntEntries;
        cnt[i] = excl;
        if (i < (uint32_t)kCntEntries - (uint32_t)kTileNodes) cnt[kTileNodes + i] = total;  // r = 1024, 1025
        // The node at this position has local rank lr = its sub-tile's byte of the prefix: inside a sub-tile the live nodes' local
        // ranks are 0, 1, 2, ... without gaps.  s_idx becomes the inverse, per sub-tile: s_idx[sub * 128 + lr] = the node (0..127 inside
        // the sub-tile) with that local rank; 0xFFFF where no live node has it.
        __syncthreads();  // everybody has read s_idx / s_key as the sort left them (tree entries above)
        s_idx[i] = 0xFFFFu;
        __syncthreads();
        if (live) s_idx[sub * kSubNodes + (uint32_t)((excl >> (8u * sub)) & 0xFFull)] = (uint16_t)(ki & (kSubNodes - 1u));
    }
    __syncthreads();
    // rows {lr >= c}, c = 0..128.  Chunk s of row c (16 bytes = the 128 nodes of sub-tile s) = OR over l >= c of the bit of the node
    // with local rank l: a suffix OR over 128 ranks -- thread (s, c) owns s_rows[c][2s..2s+1], exactly where the finished chunk
    // belongs.  (One wave ballot per (row, word), 129 of them per wave, took 7 us; a 7-step LDS scan with 14 block barriers 5 us.)
    {
        const uint32_t sub = i >> 7, c = i & 127u;
        uint64_t *mine = s_rows + c * kTileWords + 2u * sub;
        const uint32_t node = s_idx[sub * kSubNodes + c];
        uint64_t lo = (node < 64u) ? (1ull << node) : 0ull, hi = (node >= 64u && node < 128u) ? (1ull << (node - 64u)) : 0ull;
        if (i < (uint32_t)kTileWords) s_rows[(kFitRows - 1) * kTileWords + i] = 0ull;  // row 128: no node has a local rank that high
        // suffix OR over the wave's 64 ranks with lane shuffles, then the upper wave of the sub-tile hands its total to the lower one
        auto shfl_down64 = [](uint64_t x, uint32_t d) -> uint64_t {
            const uint32_t l = (uint32_t)__shfl_down((int)(uint32_t)x, d, 64), h = (uint32_t)__shfl_down((int)(uint32_t)(x >> 32), d, 64);
            return ((uint64_t)h << 32) | l;
        };
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint64_t olo = shfl_down64(lo, d), ohi = shfl_down64(hi, d);
            if (lane + d < 64u) {
                lo |= olo;
                hi |= ohi;
            }
        }
        if ((c >> 6) == 1u && lane == 0) {  // the upper wave's total = its suffix at lane 0
            s_wave[2u * sub] = lo;
            s_wave[2u * sub + 1u] = hi;
        }
        __syncthreads();
        if ((c >> 6) == 0u) {
            lo |= s_wave[2u * sub];
            hi |= s_wave[2u * sub + 1u];
        }
        mine[0] = lo;
        mine[1] = hi;
    }
    __syncthreads();
    uint64_t *T = a.tables + ((size_t)tile * a.rows + a.row_cpu + (size_t)res * kFitRows) * kTileWords;
    for (uint32_t k = i; k < (uint32_t)kFitRows * kTileWords; k += 1024u) T[k] = s_rows[k];
}

// List keys (tile_index.hpp): per (tile, list key) the tile's slots ascending by (value id, node); padding carries id 0 (absent).
struct BuildListArgs {
    const uint32_t *nlab;  // [nkeys][n]
    uint8_t *lists;        // IndexedSnapshot::d_list
    uint32_t n, nlist;
    uint32_t list_col[kMaxListKeys];
};
__global__ __launch_bounds__(1024) void k_build_tile_list(const BuildListArgs a) {
    kernarg_warm<sizeof(BuildListArgs)>();
    __shared__ uint32_t s_key[kTileNodes];
    __shared__ uint16_t s_idx[kTileNodes];
    const uint32_t tile = blockIdx.x, j = blockIdx.y, i = threadIdx.x;
    const uint32_t base = tile * kTileNodes;
    const uint32_t m = min((uint32_t)kTileNodes, a.n - base);
    uint32_t kv = (i < m) ? a.nlab[(size_t)a.list_col[j] * a.n + base + i] : 0u;
    uint32_t ki = i;
    bitonic_sort_tile(kv, ki, s_key, s_idx, i);
    uint8_t *L = a.lists + ((size_t)tile * a.nlist + j) * kListBytes;
    reinterpret_cast<uint32_t *>(L)[i] = kv;
    reinterpret_cast<uint16_t *>(L + kTileNodes * 4u)[i] = (uint16_t)ki;
}

struct BuildNamedArgs {
    const uint32_t *nlab;      // [nkeys][n]
    const uint64_t *ntaint;    // [n] or nullptr
    uint64_t *tables;
    const uint32_t *lab_meta;  // lab_base[32] (kLabList = the key has no rows), lab_max[32]
    uint32_t n, rows, nkeys, ngroups, row_valid, row_taint, named_rows;  // named_rows = row_cpu: rows [0, named_rows) are built here
};


// dynamic LDS: named_rows * 128 bytes
__global__ __launch_bounds__(1024) void k_build_tile_named(const BuildNamedArgs a) {
    kernarg_warm<sizeof(BuildNamedArgs)>();
    extern __shared__ __attribute__((aligned(16))) uint64_t s_named[];
    const uint32_t tile = blockIdx.x;
    const uint32_t base = tile * kTileNodes;
    const uint32_t m = min((uint32_t)kTileNodes, a.n - base);
    const uint32_t i = threadIdx.x, lane = i & 63u, wave = i >> 6;
    const bool live = i < m;
    const uint32_t total = a.named_rows * kTileWords;
    for (uint32_t k = i; k < total; k += 1024u) s_named[k] = 0ull;
    __syncthreads();
    {
        const uint64_t word = __ballot(live);
        if (lane == 0) s_named[a.row_valid * kTileWords + wave] = word;
    }
    if (a.ngroups) {  // row (g, s) = nodes whose taint bits of group g are a subset of s
        const uint64_t t = (live && a.ntaint) ? a.ntaint[base + i] : 0ull;
        for (uint32_t g = 0; g < a.ngroups; ++g) {
            const uint32_t tg = (uint32_t)((t >> (4u * g)) & 15ull);
            for (uint32_t sset = 0; sset < 16; ++sset) {
                const uint64_t word = __ballot(live && (tg & ~sset) == 0u);
                if (lane == 0) s_named[(a.row_taint + 16u * g + sset) * kTileWords + wave] = word;
            }
        }
    }
    for (uint32_t k = 0; k < a.nkeys; ++k) {
        const uint32_t lb = a.lab_meta[k];
        if (lb == kLabList) continue;
        const uint32_t id = live ? a.nlab[(size_t)k * a.n + base + i] : 0u;
        if (id) atomicOr((unsigned long long *)&s_named[(size_t)(lb + id - 1u) * kTileWords + wave], 1ull << lane);
    }
    __syncthreads();
    uint64_t *T = a.tables + (size_t)tile * a.rows * kTileWords;
    for (uint32_t k = i; k < total; k += 1024u) T[k] = s_named[k];
}

// ---- ksched_update_nodes ------------------------------------------------------------------------------------------------
constexpr uint32_t kPatchInline = 16;  // updates of up to this many nodes travel in the kernel arguments (no copy, no staging)
struct PatchArgs {
    int64_t *ncpu, *nmem, *nrec;  // columns; nrec = [n][kNodeRecWords] node records (kernels_direct.hpp): words 0, 1 = cpu, mem
    const uint32_t *idx;         // device arrays for count > kPatchInline, else nullptr
    const int64_t *cpu, *mem;
    uint32_t count;
    uint32_t idx_in[kPatchInline];
    int64_t cpu_in[kPatchInline], mem_in[kPatchInline];
    // small updates: the list of touched tiles for the re-index kernel that follows rides along (no separate launch, no copy)
    uint32_t *tile_out;
    uint32_t ntiles;  // <= count: every touched tile holds at least one updated node
    uint32_t tiles_in[kPatchInline];
};
__global__ __launch_bounds__(256) void k_patch_nodes(const PatchArgs a) {
    kernarg_warm<sizeof(PatchArgs)>();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.count) return;
    if (a.tile_out && t < a.ntiles) a.tile_out[t] = a.tiles_in[t];
    uint32_t node;
    int64_t c, m;
    if (a.idx) {
        node = a.idx[t];
        c = a.cpu[t];
        m = a.mem[t];
    } else {
        node = a.idx_in[t];
        c = a.cpu_in[t];
        m = a.mem_in[t];
    }
    a.ncpu[node] = c;
    a.nmem[node] = m;
    a.nrec[(size_t)kNodeRecWords * node] = c;
    a.nrec[(size_t)kNodeRecWords * node + 1] = m;
}

// node records (kernels_direct.hpp "Node records"): one 64-byte line per node for the candidate-testing picks
__global__ __launch_bounds__(256) void k_build_nrec(const int64_t *__restrict__ cpu, const int64_t *__restrict__ mem, const uint64_t *__restrict__ taints,
                                                    const uint32_t *__restrict__ lab, uint32_t nkeys, int64_t *__restrict__ rec, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t l[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) l[k] = k < nkeys ? lab[(size_t)k * n + i] : 0u;
    int64_t *r = rec + (size_t)kNodeRecWords * i;
    r[0] = cpu[i];
    r[1] = mem[i];
    r[2] = taints ? (int64_t)taints[i] : 0;
    r[3] = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) r[4 + k] = (int64_t)(((uint64_t)l[2 * k + 1] << 32) | l[2 * k]);
}

// ---- best-fit structures, after the two device sorts ------------------------------------------------------
// by_cpu[r]   : node with cpu rank r (ascending (cpu, node))           -- sort 1
// bf_order[i] : node at best-fit position i (ascending (mem, cpu, node)) -- sort 2 (stable, by mem, of by_cpu)
struct BfGatherArgs {
    const int64_t *ncpu, *nmem;
    const uint32_t *bf_order, *by_cpu;
    uint32_t *bf_rank, *cpurank;  // bf_rank[node] = position, cpurank[node] = cpu rank
    int64_t *bf_mem, *bf_cpu, *cpu_sorted;
    int64_t *samples;  // [mem s1 (n1)][mem s2 (n2)][cpu s1][cpu s2]: last element of every block of 64 / 4096
    uint32_t n, n1, n2;
};
__global__ __launch_bounds__(256) void k_bf_gather(const BfGatherArgs a) {
    kernarg_warm<sizeof(BfGatherArgs)>();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint32_t node = a.bf_order[i], cn = a.by_cpu[i];
    a.bf_rank[node] = i;
    a.cpurank[cn] = i;
    const int64_t bm = a.nmem[node], cs = a.ncpu[cn];
    a.bf_mem[i] = bm;
    a.bf_cpu[i] = a.ncpu[node];
    a.cpu_sorted[i] = cs;
    const bool last = i + 1u == a.n;
    if ((i & 63u) == 63u || last) {
        a.samples[i >> 6] = bm;
        a.samples[(size_t)a.n1 + a.n2 + (i >> 6)] = cs;
    }
    if ((i & 4095u) == 4095u || last) {
        a.samples[(size_t)a.n1 + (i >> 12)] = bm;
        a.samples[(size_t)2 * a.n1 + a.n2 + (i >> 12)] = cs;
    }
}

struct BfRowsArgs {
    const uint32_t *nlab;
    const uint64_t *ntaint;       // nullptr: every taint row is the all-valid row
    const uint32_t *bf_order, *cpurank;
    const uint32_t *lab_meta;
    uint64_t *rows;               // [rows][Wbf], zero-filled before the launch
    uint32_t n, Wbf, nkeys, ngroups, row_valid, row_taint, row_cpu0, q, levels;
};
// one wave per 64 best-fit positions (one word of every row)
__global__ __launch_bounds__(256) void k_bf_rows(const BfRowsArgs a) {
    kernarg_warm<sizeof(BfRowsArgs)>();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= a.Wbf) return;
    const uint32_t i = w * 64u + lane;
    const bool live = i < a.n;
    const uint32_t node = live ? a.bf_order[i] : 0u;
    {
        const uint64_t word = __ballot(live);
        if (lane == 0) a.rows[(size_t)a.row_valid * a.Wbf + w] = word;
    }
    for (uint32_t k = 0; k < a.nkeys; ++k) {
        if (a.lab_meta[k] == kLabList) continue;  // list keys have no rows: pods that constrain them take k_pick_bestfit_listed
        const uint32_t id = live ? a.nlab[(size_t)k * a.n + node] : 0u;
        // lanes of one wave own one word of each row: combine per distinct id with a ballot-free OR (atomics on distinct
        // words are rare collisions only inside the wave)
        if (id) atomicOr((unsigned long long *)&a.rows[(size_t)(a.lab_meta[k] + id - 1u) * a.Wbf + w], 1ull << lane);
    }
    if (a.ngroups) {
        const uint64_t t = (live && a.ntaint) ? a.ntaint[node] : 0ull;
        for (uint32_t g = 0; g < a.ngroups; ++g) {
            const uint32_t tg = (uint32_t)((t >> (4u * g)) & 15ull);
            for (uint32_t sset = 0; sset < 16; ++sset) {
                const uint64_t word = __ballot(live && (tg & ~sset) == 0u);
                if (lane == 0) a.rows[(size_t)(a.row_taint + 16u * g + sset) * a.Wbf + w] = word;
            }
        }
    }
    // cpu threshold rows: row[t] = {i : cpurank >= t * q}, t = 0..levels
    const uint32_t cr = live ? a.cpurank[node] : 0u;
    for (uint32_t t = 0; t <= a.levels; ++t) {
        const uint64_t word = __ballot(live && (uint64_t)cr >= (uint64_t)t * a.q);
        if (lane == 0) a.rows[(size_t)(a.row_cpu0 + t) * a.Wbf + w] = word;
    }
}

// 8-ary level arrays of the two sorted columns (k_pick_bestfit_lanes): level k, entry j = last element of block j of 8^k entries
struct BfLevelsArgs {
    const int64_t *bf_mem, *cpu_sorted;
    int64_t *lvl;
    uint32_t n, nlev, lvl_half, lvl_off[6];
};
__global__ __launch_bounds__(256) void k_bf_levels(const BfLevelsArgs a) {
    kernarg_warm<sizeof(BfLevelsArgs)>();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t span = 1;
    uint32_t nk = a.n;
    for (uint32_t k = 1; k <= a.nlev; ++k) {
        span *= 8u;
        nk = (nk + 7u) / 8u;
        if (t < nk) {
            const uint32_t src = (uint32_t)min((uint64_t)a.n, (uint64_t)(t + 1u) * span) - 1u;
            a.lvl[a.lvl_off[k - 1u] + t] = a.bf_mem[src];
            a.lvl[a.lvl_half + a.lvl_off[k - 1u] + t] = a.cpu_sorted[src];
        }
    }
}

// ---- the two orders of the best-fit structures: a merge sort written for them ----------------------------------------------------
// Records (k0, k1, idx) are ordered lexicographically; idx is unique, so the order is total and there is nothing to keep stable:
//   ascending (cpu, node)       -> k0 = cpu,  k1 absent (reads as 0)
//   ascending (mem, cpu, node)  -> k0 = mem,  k1 = cpu
// k_sort_runs sorts every run of 1024 consecutive records with a bitonic network over LDS (slots past n hold +infinity and are not
// written back); k_merge_pass merges neighbouring runs of `run` records into runs of 2 * run BY RANKING: a record's place in the merged
// run is its index in its own run plus the number of records of the sibling run that come before it (one binary search per record;
// the order being total, both sides count strictly smaller records).  ceil(log2(n / 1024)) passes, each a launch of n independent
// threads; the structures are rebuilt lazily, by the first PICK_BESTFIT request after the snapshot changed (n = 50 000: 6 passes).
// (Rounds 1 - 3 called rocPRIM's radix sort here: the one place a library did device work.)
struct SortArgs {
    const int64_t *k0_in, *k1_in;  // k1_in == nullptr: the records have no second key
    const uint32_t *idx_in;        // nullptr: idx = position (the first pass sorts the columns as they are)
    int64_t *k0_out, *k1_out;      // k1_out == nullptr with k1_in == nullptr
    uint32_t *idx_out;
    uint32_t n, run;
};
__device__ __forceinline__ bool rec_less(int64_t a0, int64_t a1, uint32_t ai, int64_t b0, int64_t b1, uint32_t bi) {
    if (a0 != b0) return a0 < b0;
    if (a1 != b1) return a1 < b1;
    return ai < bi;
}
__global__ __launch_bounds__(1024) void k_sort_runs(const SortArgs a) {
    __shared__ int64_t s_k0[1024], s_k1[1024];
    __shared__ uint32_t s_i[1024];
    const uint32_t t = threadIdx.x, g = blockIdx.x * 1024u + t;
    const bool live = g < a.n;
    int64_t k0 = live ? a.k0_in[g] : INT64_MAX, k1 = live ? (a.k1_in ? a.k1_in[g] : 0) : INT64_MAX;
    uint32_t ki = live ? (a.idx_in ? a.idx_in[g] : g) : 0xFFFFFFFFu;
    for (uint32_t k = 2; k <= 1024u; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            s_k0[t] = k0;
            s_k1[t] = k1;
            s_i[t] = ki;
            __syncthreads();
            const int64_t p0 = s_k0[t ^ j], p1 = s_k1[t ^ j];
            const uint32_t pi = s_i[t ^ j];
            __syncthreads();  // everybody has read before anybody writes its slot again
            const bool ascending = (t & k) == 0u, lower = (t & j) == 0u;
            const bool p_less = rec_less(p0, p1, pi, k0, k1, ki);
            const bool take = (lower == ascending) ? p_less : !p_less;  // the lower slot of an ascending pair keeps the smaller record
            k0 = take ? p0 : k0;
            k1 = take ? p1 : k1;
            ki = take ? pi : ki;
        }
    }
    if (live) {  // (the +infinity records end up behind the n live ones of the last run: positions >= n)
        a.k0_out[g] = k0;
        if (a.k1_out) a.k1_out[g] = k1;
        a.idx_out[g] = ki;
    }
}
__global__ __launch_bounds__(256) void k_merge_pass(const SortArgs a) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n) return;
    // (64-bit span arithmetic: 2 * run and base + span must not wrap for n beyond 2^31 -- ADVICE r4; the host side also refuses such an n)
    const uint64_t span = 2ull * a.run;
    const uint32_t base = (uint32_t)(((uint64_t)g / span) * span), off = g - base;
    const bool in_a = off < a.run;
    const uint32_t b_lo = (uint32_t)min((uint64_t)a.n, (uint64_t)base + a.run), b_hi = (uint32_t)min((uint64_t)a.n, (uint64_t)base + span);
    const uint32_t s_lo = in_a ? b_lo : base, s_hi = in_a ? b_hi : b_lo;  // the sibling run
    const int64_t m0 = a.k0_in[g], m1 = a.k1_in ? a.k1_in[g] : 0;
    const uint32_t mi = a.idx_in[g];
    uint32_t lo = s_lo, hi = s_hi;
    while (lo < hi) {  // records of the sibling run that come before this one
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (rec_less(a.k0_in[mid], a.k1_in ? a.k1_in[mid] : 0, a.idx_in[mid], m0, m1, mi)) lo = mid + 1u;
        else hi = mid;
    }
    const uint32_t out = base + (in_a ? off : off - a.run) + (lo - s_lo);
    a.k0_out[out] = m0;
    if (a.k1_out) a.k1_out[out] = m1;
    a.idx_out[out] = mi;
}

}  // namespace ksched
