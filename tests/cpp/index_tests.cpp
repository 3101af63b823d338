// index_tests.cpp -- host-side check of the fit part of the per-tile bitmap index (csrc/tile_index.hpp), no GPU:
// emulates, in scalar code, exactly what the fused kernel does with the index -- the Eytzinger descent for the rank,
// the cnt[rank] lookup, one row chunk per sub-tile -- and compares every (request, node) bit with `req <= avail`
// (src/predicates.rs:42), for full and partial tiles, ties, negative values and the int64 extremes.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../kube_scheduler_rs_reference_amd/csrc/tile_index.hpp"

using namespace ksched;

static int g_fail = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            if (++g_fail < 20) std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                                    \
    } while (0)

// the kernel's phase-1 rank search (kernels_fused.hpp): #values < req
static uint32_t rank_of(const int64_t *tree, int64_t req) {
    uint32_t k = 1;
    for (int level = 0; level < 10; ++level) k = 2u * k + ((tree[k] < req) ? 1u : 0u);
    uint32_t r = k - (uint32_t)kTileNodes;
    if (r == (uint32_t)kTileNodes - 1u && tree[0] < req) ++r;
    return r;
}

static void check_tile(uint32_t n_nodes, uint32_t t, const std::vector<int64_t> &cpu, const std::vector<int64_t> &mem,
                       const std::vector<int64_t> &reqs) {
    IndexedLayout l{};
    l.n = n_nodes;
    l.tiles = (n_nodes + kTileNodes - 1) / kTileNodes;
    l.row_zero = 0;
    l.row_valid = 1;
    l.row_cpu = 5;  // after zero, valid and three other rows
    l.row_mem = 5 + kFitRows;
    l.rows = 5 + 2 * kFitRows + 3;
    std::vector<uint64_t> T((size_t)l.rows * kTileWords, 0xFFFFFFFFFFFFFFFFull);  // garbage: index_tile_fit must clear its rows
    std::vector<uint64_t> aux(kAuxWords, 0x5555555555555555ull);
    index_tile_fit(l, t, cpu.data(), mem.data(), T.data(), aux.data());
    const uint32_t base = t * kTileNodes, m = std::min<uint32_t>(kTileNodes, n_nodes - base);
    for (int res = 0; res < 2; ++res) {
        const std::vector<int64_t> &v = res ? mem : cpu;
        const int64_t *tree = reinterpret_cast<const int64_t *>(aux.data()) + (size_t)res * kAuxTreeWords;
        const uint64_t *cnt = aux.data() + 2u * kAuxTreeWords + (size_t)res * kCntEntries;
        const uint32_t row0 = res ? l.row_mem : l.row_cpu;
        for (int64_t req : reqs) {
            const uint32_t r = rank_of(tree, req);
            uint32_t below = 0;
            for (uint32_t i = 0; i < m; ++i) below += v[base + i] < req;
            CHECK(r == below);
            const uint64_t c8 = cnt[r];
            for (uint32_t local = 0; local < (uint32_t)kTileNodes; ++local) {
                const uint32_t s = local / kSubNodes, c = (uint32_t)((c8 >> (8u * s)) & 0xFFu);
                CHECK(c <= (uint32_t)kSubNodes);
                const bool bit = (T[(size_t)(row0 + c) * kTileWords + (local >> 6)] >> (local & 63u)) & 1ull;
                const bool want = local < m && req <= v[base + local];
                CHECK(bit == want);
            }
        }
    }
    // the rows that are not the fit's were left alone
    CHECK(T[0] == 0xFFFFFFFFFFFFFFFFull && T[(size_t)4 * kTileWords + 15] == 0xFFFFFFFFFFFFFFFFull && T[(size_t)(l.rows - 1) * kTileWords] == 0xFFFFFFFFFFFFFFFFull);
}

int main() {
    std::mt19937_64 rng(12345);
    const int64_t lo = INT64_MIN, hi = INT64_MAX;
    for (uint32_t n_nodes : {1u, 127u, 128u, 129u, 1000u, 1024u, 1025u, 2500u}) {
        std::vector<int64_t> cpu(n_nodes), mem(n_nodes);
        for (uint32_t i = 0; i < n_nodes; ++i) {
            cpu[i] = (int64_t)(rng() % 40) * 250 - 1000;                       // many ties, some negative
            mem[i] = (i % 7 == 0) ? hi : (i % 11 == 0) ? lo : (int64_t)(rng() >> 1) - (int64_t)(rng() >> 2);  // extremes
        }
        std::vector<int64_t> reqs = {lo, lo + 1, -1001, -1000, -999, 0, 1, 249, 250, 251, 8749, 8750, 8751, hi - 1, hi};
        for (int i = 0; i < 40; ++i) reqs.push_back((i & 1) ? cpu[rng() % n_nodes] : mem[rng() % n_nodes]);  // exact hits
        for (int i = 0; i < 20; ++i) reqs.push_back((int64_t)rng());
        for (uint32_t t = 0; t < (n_nodes + kTileNodes - 1) / kTileNodes; ++t) check_tile(n_nodes, t, cpu, mem, reqs);
    }
    std::printf("index_tests: %d failed check(s)\n", g_fail);
    return g_fail ? 1 : 0;
}
