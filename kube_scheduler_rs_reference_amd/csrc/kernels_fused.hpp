// kernels_fused.hpp -- "fused" mask kernel: ONE launch per evaluation over the per-tile bitmap
// index of tile_index.hpp (same snapshot structures, same arithmetic; see that file for why
//   req <= avail[n]  <=>  pos[n] >= rank(req)            (src/predicates.rs:42)
// is exact for any int64 inputs, and how selector / taint predicates become ANDs of bitmap rows).
//
// A block owns one tile (kTileNodes = 1024 nodes = 16 mask words) and a contiguous range of pods.
// LDS holds the tile's bitmap rows, its two sorted resource arrays and a small per-wave record
// area.  After the staging barrier the 16 waves of a block are independent; each walks its own
// pods in rounds of up to 64 through two phases:
//   phase 1 (lane = pod): load the pod's requests / selector ids / tolerations (coalesced), run the
//       two branch-free descents of the breadth-first (Eytzinger) search trees in LDS (the rank lookups), turn selector
//       ids into bitmap row offsets (src/predicates.rs:45-61), park a 16-byte record in the wave's
//       LDS area.
//   phase 2 (8 lanes per pod, 16 bytes = 2 mask words per lane): read the record (LDS broadcast),
//       AND the rows it names (ds_read_b128, v_bitop3), store.  One wave store instruction emits
//       eight 128-byte row segments; there is no global load in this phase.
// The loads of the next round are issued before phase 2 of the current one.
//
// Work split (host side, run_fused): the unit is 8 pods (one phase-2 instruction).  Units are cut
// evenly into `chunks` pod ranges; the (chunk, tile) blocks are dealt to the 8 XCDs in contiguous
// chunk-major runs (block id % 8 = XCD, observed dispatch order; speed only), so the 128-byte
// segments of one pod row that adjacent tiles write meet in ONE L2 and leave it as whole cache
// lines (measured with tools/ubench2: 4.5-5.4 TB/s grouped vs 3.8 TB/s ungrouped when rows are
// not line-aligned).  Inside a block the chunk's units are cut evenly over the 16 waves.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "tile_index.hpp"

// Build-time variants of the fused kernel (tools/build_variants.sh builds one library per setting; the shipped
// library uses the defaults below, chosen from the measurements under profiles/):
//   KSCHED_SPLIT_STAGE  1 = stage the search trees first, run the first round's rank searches while the bitmap
//                           rows are still landing (two barriers); 0 = stage everything, one barrier
//   KSCHED_STORE_POLICY 0 = plain stores (write-back L2), 1 = nt, 2 = sc1 (write-through), 3 = sc0 sc1
#ifndef KSCHED_SPLIT_STAGE
#define KSCHED_SPLIT_STAGE 1
#endif
#ifndef KSCHED_STORE_POLICY
#define KSCHED_STORE_POLICY 0
#endif
//   KSCHED_STAGGER      N > 0: the upper half of a block's waves sleeps 64*N cycles before its first rank search, so
//                           the lower half's first stores start while the upper half searches (experiment)
#ifndef KSCHED_STAGGER
#define KSCHED_STAGGER 0
#endif
#ifndef KSCHED_FUSED_THREADS
#define KSCHED_FUSED_THREADS 1024
#endif

namespace ksched {

constexpr uint32_t kFusedThreads = KSCHED_FUSED_THREADS;
constexpr uint32_t kFusedWaves = kFusedThreads / 64;

// Kernel arguments: plain scalars only (they live in SGPRs; keep this small).
struct FusedArgs {
    uint32_t W, pitch, tiles, rows, nkeys, ngroups;
    uint32_t row_zero, row_valid, row_cpu_hi, row_cpu_lo, row_mem_hi, row_mem_lo, row_taint;
    uint32_t lab_base[8], lab_max[8];  // first eight label keys; further keys go through lab_meta
    const uint32_t *lab_meta;          // device copy of IndexedLayout::lab_base[32], lab_max[32]
    const uint64_t *zero64;            // eight zero bytes in device memory
    uint32_t p, units, chunks, run;    // units = ceil(p / 8); run = (chunk, tile) pairs per XCD
    uint32_t unit_q, unit_rem, tiles_rcp;  // units / chunks, units % chunks, floor(2^32 / tiles)
    uint32_t off_sorted, off_rec, off_rec2, off_trow;  // LDS byte offsets of the regions after the bitmap rows
    uint32_t debug;
    uint64_t *trace;  // diagnostics: per-block phase timestamps (100 MHz), or nullptr
};

// LDS carve-up: [rows * 128 : bitmap rows][2 * 8 KiB : sorted cpu, mem][16 waves * 64 * 16 B : records]
//                [16 * 64 * 8 B : label rows 5..8][16 * 64 * 8 B : taint rows]
inline uint32_t fused_lds_bytes(const IndexedLayout &l, bool fit, bool sel, bool taint, FusedArgs *a = nullptr) {
    uint32_t off = l.rows * 128u;
    const uint32_t off_sorted = off;
    if (fit) off += 2u * kTileNodes * 8u;
    const uint32_t off_rec = off;
    off += kFusedWaves * 64u * 16u;
    const uint32_t off_rec2 = off;
    if (sel) off += kFusedWaves * 64u * 8u;
    const uint32_t off_trow = off;
    if (taint) off += kFusedWaves * 64u * 8u;
    if (a) {
        a->off_sorted = off_sorted;
        a->off_rec = off_rec;
        a->off_rec2 = off_rec2;
        a->off_trow = off_trow;
    }
    return off;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));

template <bool FIT, bool SEL, bool TAINT, bool WANT_FIT, bool WIDE>
__global__ __launch_bounds__(kFusedThreads) void k_eval_fused(
    const uint64_t *__restrict__ g_tables, const int64_t *__restrict__ g_sorted_cpu, const int64_t *__restrict__ g_sorted_mem,
    const int64_t *__restrict__ g_pcpu, const int64_t *__restrict__ g_pmem, const uint32_t *__restrict__ g_psel,
    const uint64_t *__restrict__ g_ptol, uint64_t *__restrict__ out_feas, uint64_t *__restrict__ out_fit, const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t b = blockIdx.x;
    // (chunk, tile) pairs in chunk-major order are dealt to the XCDs in contiguous runs: XCD x = block id % 8
    // (observed dispatch order; speed only) takes pairs [x * run, (x + 1) * run), so the tile-blocks of one
    // pod range sit on one XCD (at most one seam per XCD boundary) and share its L2.
    uint32_t tile, chunk;
    if (!(a.debug & 32u)) {
        const uint32_t l = (b & 7u) * a.run + (b >> 3);
        if ((b >> 3) >= a.run) return;
        chunk = __umulhi(l, a.tiles_rcp);  // l / tiles by a host-made reciprocal (no division sequence in the prologue)
        tile = l - chunk * a.tiles;
        if (tile >= a.tiles) {  // the reciprocal can be one short
            tile -= a.tiles;
            ++chunk;
        }
    } else {  // experiment: plain round-robin of (tile, chunk) pairs
        tile = b % a.tiles;
        chunk = b / a.tiles;
    }
    if (chunk >= a.chunks) return;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform: scalar control flow below
    const bool tracer = a.trace && threadIdx.x == 0;
    auto stamp = [&](uint32_t i) {
        if (tracer) a.trace[(size_t)b * 8u + i] = wall_clock64();
    };
    stamp(0);
    // this wave's units [u, u_hi): chunk range cut evenly over the waves
    // balanced split of `units` over `chunks`: the first unit_rem chunks get unit_q + 1 units
    const uint32_t c_lo = chunk * a.unit_q + min(chunk, a.unit_rem);
    const uint32_t c_n = a.unit_q + (chunk < a.unit_rem ? 1u : 0u);
    uint32_t u = c_lo + (wave * c_n) / kFusedWaves;
    const uint32_t u_hi = c_lo + ((wave + 1u) * c_n) / kFusedWaves;

    // ---- pod operands ---------------------------------------------------------------------------
    // Loaded with inline-asm global loads that the compiler's s_waitcnt bookkeeping does not see, and
    // awaited by hand (wait_ops<N>): gfx950 has ONE in-order counter (vmcnt) for loads and stores, and
    // the compiler would wait vmcnt(0) for these operands at every round, i.e. drain the wave's mask
    // stores before the next round may start.  Counting by hand (N = stores issued after the loads)
    // lets the stores of round r stay in flight while round r+1 is prepared.  The loaded registers
    // are only named by the load statement and the wait statement (cdna_hip_programming.md 5.7 (ii)).
    int64_t rc = 0, rm = 0;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    uint64_t tol = 0;
    auto issue_ops = [&](uint32_t pod) {
        const uint32_t pc = min(pod, a.p - 1u);  // clamp: lanes past the end read a valid row and are masked later
        if (FIT) {
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rc) : "v"(g_pcpu + pc) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rm) : "v"(g_pmem + pc) : "memory");
        }
        if (SEL) {
            const uint32_t kl = a.nkeys - 1u;
#define KSCHED_LOAD_SEL(K, DST) asm volatile("global_load_dword %0, %1, off" : "=v"(DST) : "v"(g_psel + (size_t)min((uint32_t)K, kl) * a.p + pc) : "memory")
            KSCHED_LOAD_SEL(0, s0);
            KSCHED_LOAD_SEL(1, s1);
            KSCHED_LOAD_SEL(2, s2);
            KSCHED_LOAD_SEL(3, s3);
            KSCHED_LOAD_SEL(4, s4);
            KSCHED_LOAD_SEL(5, s5);
            KSCHED_LOAD_SEL(6, s6);
            KSCHED_LOAD_SEL(7, s7);
#undef KSCHED_LOAD_SEL
        }
        if (TAINT) {  // no tolerations given = tolerate nothing: every lane reads one zero word (the select is on the address,
                      // never on the in-flight destination register)
            const uint64_t *tp = g_ptol ? g_ptol + pc : a.zero64;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(tol) : "v"(tp) : "memory");
        }
    };
    // wait until at most N vector-memory operations issued after the operand loads are still outstanding
    // (a macro, not a lambda: asm operands cannot be lambda captures)
#define KSCHED_WAIT_OPS(N)                                                                                                            \
    asm volatile("s_waitcnt vmcnt(%c11)"                                                                                              \
                 : "+v"(rc), "+v"(rm), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7), "+v"(tol)  \
                 : "n"(N)                                                                                                             \
                 : "memory")
    const int64_t *s_cpu = reinterpret_cast<const int64_t *>(smem + a.off_sorted);
    const int64_t *s_mem = s_cpu + kTileNodes;
    uint4 *s_rec = reinterpret_cast<uint4 *>(smem + a.off_rec) + wave * 64u;
    uint2 *s_rec2 = reinterpret_cast<uint2 *>(smem + a.off_rec2) + wave * 64u;  // label rows 5..8 of a pod (selector keys 5..8)
    uint2 *s_trow = reinterpret_cast<uint2 *>(smem + a.off_trow) + wave * 64u;  // four taint rows per pod

    const uint32_t wp = lane & 7u, sub = lane >> 3;
    const uint32_t w0 = tile * kTileWords + 2u * wp;  // first mask word of this lane in phase 2
    // a lane may store 16 bytes when both words lie inside the row pitch (words in [W, pitch) are padding)
    const bool has0 = w0 < a.pitch, has1 = w0 + 1u < a.pitch;
    const bool tile_full = (tile + 1u) * kTileWords <= a.pitch;  // block-uniform
    const bool taint_inline = a.ngroups <= 4u;
    // Records hold row numbers scaled by RS: byte offsets (RS = 128; one SDWA add per address) when the
    // table is below 64 KiB, else 16-byte units (RS = 8; extract + shift-add).
    constexpr uint32_t RS = WIDE ? 8u : 128u;
    const uint8_t *Tb = smem + wp * 16u;
    auto ldrow = [&](uint32_t field) -> u32x4 { return *reinterpret_cast<const u32x4 *>(Tb + (WIDE ? field * 16u : field)); };
    auto lo16 = [](uint32_t x) { return x & 0xFFFFu; };
    auto hi16 = [](uint32_t x) { return x >> 16; };

    // Row loads of one pod-row of phase 2 (issued together, consumed later: two iterations are
    // interleaved by hand so that ~20 LDS reads are in flight per wave).
    struct Rows {
        u32x4 c0, c1, c2, m0, m1, m2, l0, l1, l2, l3, t0, t1, t2, t3, x0, x1, x2, x3;
    };
    auto load_extra = [&](const uint2 r2, Rows &R) {  // label rows 5..8 (rounds where some pod constrains more than four keys)
        R.x0 = ldrow(lo16(r2.x));
        R.x1 = ldrow(hi16(r2.x));
        R.x2 = ldrow(lo16(r2.y));
        R.x3 = ldrow(hi16(r2.y));
    };
    auto load_rows = [&](const uint4 r, const uint2 tr, Rows &R) {
        if (FIT) {
            R.c0 = ldrow(lo16(r.x));
            R.c1 = ldrow(lo16(r.x) + RS);
            R.c2 = ldrow(hi16(r.x));
            R.m0 = ldrow(lo16(r.y));
            R.m1 = ldrow(lo16(r.y) + RS);
            R.m2 = ldrow(hi16(r.y));
        } else {
            R.c0 = ldrow(a.row_valid * RS);
        }
        if (SEL) {
            R.l0 = ldrow(lo16(r.z));
            R.l1 = ldrow(hi16(r.z));
            R.l2 = ldrow(lo16(r.w));
            R.l3 = ldrow(hi16(r.w));
        }
        if (TAINT) {
            R.t0 = ldrow(lo16(tr.x));
            R.t1 = ldrow(hi16(tr.x));
            R.t2 = ldrow(lo16(tr.y));
            R.t3 = ldrow(hi16(tr.y));
        }
    };
    auto fit_of = [&](const Rows &R) -> u32x4 {
        if (FIT) return (R.c0 & (R.c1 | R.c2)) & (R.m0 & (R.m1 | R.m2));  // pos >= rank, both resources
        return R.c0;
    };
    // The asm forms are opaque to the compiler's hazard recognizer: a VALU write of the data registers right after a
    // store of more than 64 bits needs 2 wait states on gfx950 (the compiler inserts them for its own stores), hence
    // the s_nop 1 inside the statement.
    auto store16 = [&](uint64_t *dst, size_t o, const u32x4 f) {
#if KSCHED_STORE_POLICY == 1
        __builtin_nontemporal_store(f, reinterpret_cast<u32x4_a8 *>(dst + o));
#elif KSCHED_STORE_POLICY == 2
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + o), "v"(f) : "memory");
#elif KSCHED_STORE_POLICY == 3
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst + o), "v"(f) : "memory");
#elif KSCHED_STORE_POLICY == 4
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(dst + o), "v"(f) : "memory");
#elif KSCHED_STORE_POLICY == 5
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(dst + o), "v"(f) : "memory");
#elif KSCHED_STORE_POLICY == 6
        asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(dst + o), "v"(f) : "memory");
#else
        *reinterpret_cast<u32x4_a8 *>(dst + o) = f;
#endif
    };
    auto combine_store = [&](uint32_t pod, const Rows &R, bool extra) {
        const size_t o = (size_t)pod * a.pitch + w0;
        u32x4 f = fit_of(R);
        if (WANT_FIT) store16(out_fit, o, f);
        if (SEL) f = ((f & R.l0) & R.l1) & (R.l2 & R.l3);
        if (SEL && extra) f = ((f & R.x0) & R.x1) & (R.x2 & R.x3);
        if (TAINT) f = ((f & R.t0) & R.t1) & (R.t2 & R.t3);
        store16(out_feas, o, f);  // out_feas is never null here (the API supplies a scratch mask when the caller gives none)
    };

    // Checked form of one pod-row: end-of-range / partial-tile predicates and the overflow walks
    // (more than eight constrained keys, more than four taint groups).  Rare, not unrolled.
    auto emit_checked = [&](uint32_t pod, const uint4 r, const uint2 r2, const uint2 tr, bool over) {
        const bool live = pod < a.p && has0;
        Rows R;
        load_rows(r, tr, R);
        if (SEL) load_extra(r2, R);
        u32x4 f = fit_of(R);
        const size_t o = (size_t)pod * a.pitch + w0;
        auto store = [&](uint64_t *dst) {
            if (has1) store16(dst, o, f);
            else dst[o] = ((uint64_t)f.y << 32) | f.x;
        };
        if (WANT_FIT && live) store(out_fit);
        if (SEL) {
            if (!over) {
                f = ((f & R.l0) & R.l1) & (R.l2 & R.l3);
                f = ((f & R.x0) & R.x1) & (R.x2 & R.x3);
            } else if (live) {  // more than eight constrained keys: walk every key of this pod
                for (uint32_t k = 0; k < a.nkeys; ++k) {
                    const uint32_t s = g_psel[(size_t)k * a.p + pod];
                    if (s != 0u) f &= ldrow(((s <= a.lab_meta[32u + k]) ? (a.lab_meta[k] + s - 1u) : a.row_zero) * RS);
                }
            }
        }
        if (TAINT) {
            if (taint_inline) {
                f = ((f & R.t0) & R.t1) & (R.t2 & R.t3);
            } else if (live) {
                const uint64_t t = g_ptol ? g_ptol[pod] : 0ull;
                for (uint32_t g = 0; g < a.ngroups; ++g) f &= ldrow((a.row_taint + 16u * g + (uint32_t)((t >> (4u * g)) & 15ull)) * RS);
            }
        }
        if (live && out_feas && !(a.debug & 1u)) store(out_feas);
    };

    // ---- phase 1: lane = pod pod0 + lane (branch-free); returns the overflow ballot ---------------
    bool extra_any = false;
    auto phase1 = [&](uint32_t pod0) -> uint64_t {
        uint4 rec;
        rec.x = rec.y = 0;
        if (FIT) {
            // r = #sorted values < req: two interleaved descents of the tile's implicit search trees.  The sorted
            // arrays are stored in breadth-first (Eytzinger) order (tile_index.hpp): the candidates of one level are
            // contiguous, so the 64 lanes of a step hit distinct LDS banks (levels 0..5: conflict-free; deeper levels:
            // random 2-3-way) instead of the 16-32-way conflicts of power-of-two strides in a plain sorted array.
            uint32_t lc = 0, lm = 0;
            if (!(a.debug & 2u)) {
                uint32_t kc = 1, km = 1;
#pragma unroll
                for (uint32_t level = 0; level < 10; ++level) {
                    const int64_t vc = s_cpu[kc], vm = s_mem[km];
                    kc = 2u * kc + ((vc < rc) ? 1u : 0u);  // right child when the node's value is below the request
                    km = 2u * km + ((vm < rm) ? 1u : 0u);
                }
                lc = kc - (uint32_t)kTileNodes;  // the gap reached = #values < req among sorted[0..1022]
                lm = km - (uint32_t)kTileNodes;
                lc += (lc == (uint32_t)kTileNodes - 1u && s_cpu[0] < rc) ? 1u : 0u;  // slot 0 holds sorted[1023]: 1023 -> 1024
                lm += (lm == (uint32_t)kTileNodes - 1u && s_mem[0] < rm) ? 1u : 0u;
            }
            rec.x = ((a.row_cpu_hi + (lc >> 5)) * RS) | (((a.row_cpu_lo + (lc & 31u)) * RS) << 16);
            rec.y = ((a.row_mem_hi + (lm >> 5)) * RS) | (((a.row_mem_lo + (lm & 31u)) * RS) << 16);
        }
        const uint32_t rv = a.row_valid * RS;
        uint32_t lr[8] = {rv, rv, rv, rv, rv, rv, rv, rv};
        uint32_t cnt = 0;
        if (SEL) {
            const uint32_t sv[8] = {s0, s1, s2, s3, s4, s5, s6, s7};
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                const uint32_t s = (k < a.nkeys) ? sv[k] : 0u;
                const bool on = s != 0u;
                const uint32_t row = ((s <= a.lab_max[k]) ? (a.lab_base[k] + s - 1u) : a.row_zero) * RS;
#pragma unroll
                for (uint32_t j = 0; j <= k; ++j) lr[j] = (on && cnt == j) ? row : lr[j];  // the (cnt+1)-th constrained key goes to slot cnt
                cnt += on ? 1u : 0u;
            }
            if (a.nkeys > 8u) {  // keys 9.. : any constraint there sends the pod down the overflow walk
                const uint32_t pc = min(pod0 + lane, a.p - 1u);
                for (uint32_t k = 8; k < a.nkeys; ++k) cnt += (g_psel[(size_t)k * a.p + pc] != 0u) ? 9u : 0u;
            }
            s_rec2[lane] = make_uint2(lr[4] | (lr[5] << 16), lr[6] | (lr[7] << 16));
        }
        rec.z = lr[0] | (lr[1] << 16);
        rec.w = lr[2] | (lr[3] << 16);
        s_rec[lane] = rec;
        if (TAINT) {
            uint32_t t[4];
#pragma unroll
            for (uint32_t g = 0; g < 4; ++g)
                t[g] = (g < a.ngroups) ? (a.row_taint + 16u * g + (uint32_t)((tol >> (4u * g)) & 15ull)) * RS : rv;
            s_trow[lane] = make_uint2(t[0] | (t[1] << 16), t[2] | (t[3] << 16));
        }
        extra_any = __ballot(cnt > 4u) != 0ull;  // some pod of the round needs label rows 5..8
        const uint64_t over = __ballot(cnt > 8u);  // only possible with more than eight label keys
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return over;
    };

    // vector-memory operations the unchecked phase 2 issues per round (what KSCHED_WAIT_OPS may leave in flight)
    constexpr int kFastStores = 8 * (WANT_FIT ? 2 : 1);

    // ---- one software-pipelined loop: the operand loads of round r+1 are issued before the stores of
    // round r and awaited after them, from ONE load site and ONE wait site (no register copies can be
    // scheduled between a load and its wait).  The tile is staged inside the first trip, after the
    // first operand loads have been issued.
    bool more = u < u_hi, have_prev = false, first = true, stamped4 = false;
    uint32_t prev_u = 0, prev_nu = 0;
    uint64_t prev_over = 0;
    bool prev_extra = false;
    while (true) {
        if (more) issue_ops(u * 8u + lane);
        if ((a.debug & 128u) && have_prev && !stamped4) stamp(1);  // experiment: after the 2nd round's operand loads were issued
        if (first) {
#if KSCHED_SPLIT_STAGE
            // Stage the tile global -> LDS without a VGPR round trip (global_load_lds_dwordx4: per-lane global address,
            // LDS destination = M0 + lane*16), in two steps so that the first round's rank searches overlap the
            // landing of the bitmap rows: (1) the two search trees (16 KiB), wait, barrier; (2) the bitmap rows are
            // issued and left in flight -- phase 1 only reads the trees; the rows are awaited (vmcnt(0) + barrier)
            // right after the first phase 1, before any phase 2.  The DMA is inline asm: the compiler's own
            // LDS-DMA tracking would drain it (vmcnt(0)) at the first LDS read.
            auto stage = [&](const void *gsrc, uint32_t lds_off, uint32_t bytes) {
                const uint8_t *g = static_cast<const uint8_t *>(gsrc);
                for (uint32_t off = wave * 1024u; off < bytes; off += kFusedWaves * 1024u) {
                    const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
                        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(smem + lds_off + off));
                    if (off + lane * 16u < bytes) {
                        uint32_t keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep)
                                     : "v"(g + off + lane * 16u), "s"(lds_dst)
                                     : "memory");
                    }
                }
            };
            if (FIT) {
                stage(g_sorted_cpu + (size_t)tile * kTileNodes, a.off_sorted, kTileNodes * 8u);
                stage(g_sorted_mem + (size_t)tile * kTileNodes, a.off_sorted + kTileNodes * 8u, kTileNodes * 8u);
            }
            if (!(a.debug & 128u)) stamp(1);
            // Drains this wave's tree DMAs AND its first operand loads (issued earlier).  Needed even without FIT: the
            // counted wait below assumes that only stores are younger than the operand loads; in the first trip the row
            // DMAs are, so the operands must have landed before those are issued.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (FIT) __syncthreads();
            if (!(a.debug & 128u)) stamp(2);
            if (!(a.debug & 8u)) stage(g_tables + (size_t)tile * a.rows * kTileWords, 0u, a.rows * 128u);
#else
            // stage the tile: bitmap rows + sorted arrays, global -> LDS without a VGPR round trip
            // (global_load_lds_dwordx4: per-lane global address, LDS destination = wave-uniform base + lane*16)
            auto stage = [&](const void *gsrc, uint32_t lds_off, uint32_t bytes) {
                const uint8_t *g = static_cast<const uint8_t *>(gsrc);
                for (uint32_t off = wave * 1024u; off < bytes; off += kFusedWaves * 1024u) {
                    if (off + lane * 16u < bytes)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + off + lane * 16u),
                                                         (__attribute__((address_space(3))) void *)(smem + lds_off + off), 16, 0, 0);
                }
            };
            if (!(a.debug & 8u)) stage(g_tables + (size_t)tile * a.rows * kTileWords, 0u, a.rows * 128u);
            if (FIT) {
                stage(g_sorted_cpu + (size_t)tile * kTileNodes, a.off_sorted, kTileNodes * 8u);
                stage(g_sorted_mem + (size_t)tile * kTileNodes, a.off_sorted + kTileNodes * 8u, kTileNodes * 8u);
            }
            if (!(a.debug & 128u)) stamp(1);
            __syncthreads();
            if (!(a.debug & 128u)) stamp(2);
#endif
        }
        if (have_prev) {
            // ============ phase 2 of the previous round: 8 lanes per pod, 2 words per lane ===========
            const uint32_t pod0 = prev_u * 8u;
            // A short round (prev_nu < 8) is always the wave's last one: no operand loads are in flight behind
            // it, so the counted wait does not depend on how many stores it issues.
            const bool fast = (prev_nu == 8u || !more) && prev_over == 0ull && pod0 + prev_nu * 8u <= a.p && tile_full &&
                              (!TAINT || taint_inline) && !(a.debug & 16u);
            if (fast && prev_nu == 8u) {
                // STEP pod rows per step (two when the row registers allow: ~20 LDS reads in flight per
                // wave); the records of the next step are fetched while this step's rows are combined.
                constexpr uint32_t STEP = TAINT ? 1u : 2u;
                uint4 rn[STEP];
                uint2 tn[STEP];
#pragma unroll
                for (uint32_t j = 0; j < STEP; ++j) {
                    rn[j] = s_rec[j * 8u + sub];
                    tn[j] = TAINT ? s_trow[j * 8u + sub] : make_uint2(0u, 0u);
                }
                if (SEL && prev_extra) {  // wave-uniform: some pod constrains 5..8 keys, all pods read eight label rows
#pragma unroll
                    for (uint32_t it = 0; it < 8; ++it) {
                        Rows A;
                        const uint4 r = s_rec[it * 8u + sub];
                        load_rows(r, TAINT ? s_trow[it * 8u + sub] : make_uint2(0u, 0u), A);
                        load_extra(s_rec2[it * 8u + sub], A);
                        combine_store(pod0 + it * 8u + sub, A, true);
                    }
                } else {
#pragma unroll
                    for (uint32_t it = 0; it < 8; it += STEP) {
                        Rows R[STEP];
#pragma unroll
                        for (uint32_t j = 0; j < STEP; ++j) load_rows(rn[j], tn[j], R[j]);
                        if (it + STEP < 8u) {
#pragma unroll
                            for (uint32_t j = 0; j < STEP; ++j) {
                                rn[j] = s_rec[(it + STEP + j) * 8u + sub];
                                tn[j] = TAINT ? s_trow[(it + STEP + j) * 8u + sub] : make_uint2(0u, 0u);
                            }
                        }
#pragma unroll
                        for (uint32_t j = 0; j < STEP; ++j) combine_store(pod0 + (it + j) * 8u + sub, R[j], false);
                    }
                }
            } else if (fast) {
                // short last round of the wave: same unchecked rows, one at a time
#pragma unroll 1
                for (uint32_t it = 0; it < prev_nu; ++it) {
                    Rows A;
                    load_rows(s_rec[it * 8u + sub], TAINT ? s_trow[it * 8u + sub] : make_uint2(0u, 0u), A);
                    if (SEL) load_extra(s_rec2[it * 8u + sub], A);
                    combine_store(pod0 + it * 8u + sub, A, true);
                }
            } else {
                if (!(a.debug & 16u)) {
#pragma unroll 1
                    for (uint32_t it = 0; it < prev_nu; ++it) {
                        const uint32_t pl = it * 8u + sub;
                        emit_checked(pod0 + pl, s_rec[pl], SEL ? s_rec2[pl] : make_uint2(0u, 0u), TAINT ? s_trow[pl] : make_uint2(0u, 0u),
                                     (prev_over >> pl) & 1ull);
                    }
                }
                // an unknown number of stores (and overflow-walk loads) went out: drain, so that the
                // counted wait below still covers the operand loads
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (!stamped4) stamp(4);
            stamped4 = true;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const bool had_more = more;
        if (more) {
#if KSCHED_STAGGER
            if (first && wave >= kFusedWaves / 2u) __builtin_amdgcn_s_sleep(KSCHED_STAGGER);
#endif
            KSCHED_WAIT_OPS(kFastStores);  // operands of round u have landed; up to kFastStores younger stores may be in flight
            prev_over = phase1(u * 8u);
            prev_extra = extra_any;
            if (!have_prev) stamp(3);
            prev_u = u;
            prev_nu = min(8u, u_hi - u);
            have_prev = true;
            u += 8u;
            more = u < u_hi;
        }
#if KSCHED_SPLIT_STAGE
        if (first) {  // every wave's first trip (block-uniform): the bitmap rows have landed before any phase 2 reads them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no operand loads of the next round are in flight yet
            __syncthreads();
        }
#endif
        first = false;
        if (!had_more) break;
    }
#undef KSCHED_WAIT_OPS
    if (a.trace && lane == 0) {  // every wave: latest loop end / drain of the block
        atomicMax((unsigned long long *)&a.trace[(size_t)b * 8u + 5u], (unsigned long long)wall_clock64());
        __builtin_amdgcn_s_waitcnt(0);  // all counters to zero: this wave's stores have been acknowledged
        atomicMax((unsigned long long *)&a.trace[(size_t)b * 8u + 6u], (unsigned long long)wall_clock64());
    }
    if (tracer) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.trace[(size_t)b * 8u + 7u] = xcc;
    }
}

struct FusedLaunch {
    dim3 grid;
    uint32_t lds;
    hipStream_t stream;
    const IndexedSnapshot *snap;
    const int64_t *pcpu, *pmem;
    const uint32_t *psel;
    const uint64_t *ptol;
    uint64_t *out_feas, *out_fit;
    hipEvent_t ev_start, ev_stop;  // optional: attached to the dispatch itself (exact kernel duration)
};

template <bool FIT, bool SEL, bool TAINT, bool WANT_FIT, bool WIDE>
inline hipError_t launch_fused_k(const FusedLaunch &q, const FusedArgs &a) {
    auto kern = k_eval_fused<FIT, SEL, TAINT, WANT_FIT, WIDE>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds);
    if (e != hipSuccess) return e;
    const IndexedSnapshot &s = *q.snap;
    if (q.ev_start || q.ev_stop)
        hipExtLaunchKernelGGL(kern, q.grid, dim3(kFusedThreads), q.lds, q.stream, q.ev_start, q.ev_stop, 0, s.d_tables, s.d_sorted_cpu,
                              s.d_sorted_mem, q.pcpu, q.pmem, q.psel, q.ptol, q.out_feas, q.out_fit, a);
    else
        hipLaunchKernelGGL(kern, q.grid, dim3(kFusedThreads), q.lds, q.stream, s.d_tables, s.d_sorted_cpu, s.d_sorted_mem, q.pcpu, q.pmem,
                           q.psel, q.ptol, q.out_feas, q.out_fit, a);
    return hipGetLastError();
}

template <bool FIT, bool SEL, bool TAINT>
inline hipError_t launch_fused_t(bool want_fit, bool wide, const FusedLaunch &q, const FusedArgs &a) {
    if (want_fit) return wide ? launch_fused_k<FIT, SEL, TAINT, true, true>(q, a) : launch_fused_k<FIT, SEL, TAINT, true, false>(q, a);
    return wide ? launch_fused_k<FIT, SEL, TAINT, false, true>(q, a) : launch_fused_k<FIT, SEL, TAINT, false, false>(q, a);
}

inline bool fused_applicable(const IndexedSnapshot &s, uint32_t flags) {
    if (!s.built) return false;
    return fused_lds_bytes(s.lay, flags & KSCHED_FIT, (flags & KSCHED_SEL) && s.lay.nkeys, (flags & KSCHED_TAINT) && s.lay.ngroups) <= kLdsBudget;
}

// pitch = words between consecutive pod rows of the output masks (>= W; W = packed).
inline hipError_t run_fused(const IndexedSnapshot &s, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                            const uint64_t *ptol, uint32_t flags, uint64_t *out_feas, uint64_t *out_fit, uint32_t pitch, hipStream_t stream,
                            uint32_t debug = 0, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, uint64_t *trace = nullptr,
                            uint32_t trace_blocks = 0) {
    const IndexedLayout &l = s.lay;
    FusedArgs a{};
    a.W = l.W;
    a.pitch = pitch;
    a.tiles = l.tiles;
    a.rows = l.rows;
    a.nkeys = l.nkeys;
    a.ngroups = l.ngroups;
    a.row_zero = l.row_zero;
    a.row_valid = l.row_valid;
    a.row_cpu_hi = l.row_cpu_hi;
    a.row_cpu_lo = l.row_cpu_lo;
    a.row_mem_hi = l.row_mem_hi;
    a.row_mem_lo = l.row_mem_lo;
    a.row_taint = l.row_taint;
    for (int k = 0; k < 8; ++k) {
        a.lab_base[k] = l.lab_base[k];
        a.lab_max[k] = l.lab_max[k];
    }
    a.lab_meta = s.d_lab_meta;
    a.zero64 = reinterpret_cast<const uint64_t *>(s.d_lab_meta + 64);
    a.p = p;
    a.units = (p + 7u) / 8u;
    a.debug = debug;
    const bool do_fit = flags & KSCHED_FIT;
    const bool do_sel = (flags & KSCHED_SEL) && psel && l.nkeys;
    const bool do_taint = (flags & KSCHED_TAINT) && l.ngroups;
    const uint32_t lds = fused_lds_bytes(l, do_fit, do_sel, do_taint, &a);

    // chunks: as many pod ranges as keep every block resident at once (256 CUs x blocks per CU), but no
    // more than one round (64 pods) per wave needs.
    const uint32_t blocks_per_cu = std::max(1u, std::min(kLdsBudget / lds, 2048u / kFusedThreads));
    const uint32_t rounds = (a.units + 7u) / 8u;
    const uint32_t want = (rounds + kFusedWaves - 1u) / kFusedWaves;  // chunks that give every wave one round
    a.chunks = std::max(1u, std::min((256u * blocks_per_cu) / l.tiles, want));
    a.unit_q = a.units / a.chunks;
    a.unit_rem = a.units % a.chunks;
    a.tiles_rcp = (uint32_t)std::min<uint64_t>((1ull << 32) / l.tiles, 0xFFFFFFFFull);
    const uint32_t total = a.chunks * l.tiles;
    a.run = (total + 7u) / 8u;
    const dim3 grid((debug & 32u) ? total : a.run * 8u);
    a.trace = (trace && grid.x <= trace_blocks) ? trace : nullptr;
    const bool want_fit = (flags & KSCHED_WANT_FIT_MASK) && out_fit;
    const int sel = do_sel ? 1 : 0, tnt = do_taint ? 1 : 0, fit = do_fit ? 1 : 0;
    const bool wide = l.rows * 128u > 65536u - 128u;  // row byte offsets no longer fit 16 bits
    const FusedLaunch q{grid, lds, stream, &s, pcpu, pmem, psel, ptol, out_feas, out_fit, ev_start, ev_stop};
#define KSCHED_FUSED_CASE(F, S, T) return launch_fused_t<F, S, T>(want_fit, wide, q, a)
    switch (fit * 4 + sel * 2 + tnt) {
        case 0: KSCHED_FUSED_CASE(false, false, false);
        case 1: KSCHED_FUSED_CASE(false, false, true);
        case 2: KSCHED_FUSED_CASE(false, true, false);
        case 3: KSCHED_FUSED_CASE(false, true, true);
        case 4: KSCHED_FUSED_CASE(true, false, false);
        case 5: KSCHED_FUSED_CASE(true, false, true);
        case 6: KSCHED_FUSED_CASE(true, true, false);
        default: KSCHED_FUSED_CASE(true, true, true);
    }
#undef KSCHED_FUSED_CASE
}

}  // namespace ksched
