// kernels_fused.hpp -- "fused" mask kernel: ONE launch per evaluation over the per-tile bitmap
// index of tile_index.hpp (same snapshot structures, same arithmetic; see that file for why
//   req <= avail[n]  <=>  pos[n] >= rank(req)  <=>  lr[n] >= cnt[rank][sub-tile of n]     (src/predicates.rs:42)
// is exact for any int64 inputs, and how selector / taint predicates become ANDs of bitmap rows).
//
// A block owns one tile (kTileNodes = 1024 nodes = 16 mask words) and a contiguous range of pods.
// LDS holds the tile's bitmap rows, its aux block (two search trees, two cnt tables) and a small
// per-wave record area.  After the staging barrier the 16 waves of a block are independent; each
// walks its own pods in rounds of up to 64 through two phases:
//   phase 1 (lane = pod): load the pod's requests / selector ids / tolerations (coalesced), run the
//       two branch-free descents of the breadth-first (Eytzinger) search trees in LDS (the rank lookups),
//       fetch cnt[rank] of each resource (8 bytes: one row number per sub-tile), turn selector ids into
//       bitmap row offsets (src/predicates.rs:45-61; the constrained keys are scattered into the pod's
//       record with one 2-byte LDS store each), park the records in the wave's LDS area.
//   phase 2 (8 lanes per pod, 16 bytes = 2 mask words = one sub-tile per lane): read the records, AND
//       the row chunks they name (ds_read_b128: ONE per resource for the fit, one per constrained-key
//       slot, one per taint group), store.  One wave store instruction emits eight 128-byte row
//       segments; there is no global load in this phase.
// The loads of the next round are issued before phase 2 of the current one.
//
// The kernel is bound by instruction issue (VALU ~55 %, LDS ~40 % of the cycles at C4, rocprofv3 SQ counters in
// profiles/), not by LDS or HBM bandwidth alone, so both phases are written to minimise VALU work: records hold
// ready-made LDS byte offsets (one add per row address), the fit's row is one byte read + one shift-add per
// resource, and the mask stores use an SGPR base + 32-bit lane offset (one add per store instead of a 64-bit
// multiply-add).
//
// Phase-2 lane layout: ds_read_b128 is served in four fixed groups of 16 lanes ({0-3,12-15,20-27},
// {4-11,16-19,28-31} and the same +32; MI355X_MICROARCH.md "LDS"), and only lanes of one group can
// conflict.  The 8 lanes of a pod are therefore placed inside ONE group (a pod's 8 chunks of a row
// are 128 contiguous bytes = 32 distinct banks), two pods per group: a group conflicts only when its
// two pods read different rows of the same bank half.
//
// Work split (host side, run_fused): the unit is 8 pods (one phase-2 instruction).  Units are cut
// evenly into `chunks` pod ranges; the (chunk, tile) blocks are dealt to the 8 XCDs in contiguous
// chunk-major runs (block id % 8 = XCD, observed dispatch order; speed only), so the 128-byte
// segments of one pod row that adjacent tiles write meet in ONE L2 (tools/ubench3.hip: the pattern
// streams at the flat-store rate).  Inside a block the chunk's units are cut evenly over the 16 waves.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernarg.hpp"
#include "kernels_direct.hpp"  // SelectArgs, select_one_pod: the sampled pick that rides in this kernel's fill (PICK)
#include "tile_index.hpp"

// Build-time variants (tools/build_variants.sh builds one library per setting for A/B timing; the shipped library
// uses the defaults below, chosen from the measurements under profiles/):
//   KSCHED_STORE_POLICY  mask stores: 0 = plain (write-back L2), 1 = nt, 2 = sc1 (write-through: no dirty lines are
//                        left for the end-of-kernel L2 flush), 3 = sc0 sc1, 4 = sc1 nt, 5 = sc0 sc1 nt
//                        Shipped: 5.  Rounds 1 - 4 shipped 2, chosen with ONE mask buffer rewritten in place (the Infinity Cache absorbing part of
//                        the stream).  With the outputs rotated over more than the cache AND fresh inputs every step (round 5: what a scheduler does)
//                        the non-temporal write-through forms win everywhere -- the mask is never read again by this kernel, and lines that do not
//                        linger in the caches leave them to the operands and the tile index: C3 20.1 -> 18.4 us per step, the C4 shard 42.5 -> 38.1,
//                        the C5 shard 220 -> 196 (sessions r5g, r5h: profiles/r05_r5h_store_policy.txt; 4 and 5 within noise of each other except at
//                        the C5 shard, where 5 is 2 - 3 % ahead); in place nothing changes (19.9 - 20.5 us either way).
//                        6 = MIXED (shipped since the end of round 6): policy 5, except that ONE of a round's eight pod-row steps (KSCHED_STORE_ALT, default: the last)
//                        is stored write-through WITHOUT the nt hint (sc1).  One such store in eight is enough: the C5 shard's mask kernel 140 - 151 us -> 137 - 138 us
//                        in every one of twelve fresh processes (the spread over allocations that round 5 found and round 6 could only probe around is gone with it),
//                        its step 197 - 209 -> 193 us; the C4 shard 37.2 -> 36.9 us, C3 unchanged (17.4 us, steadier).  More of them loses again (every other step:
//                        = policy 5; three in four: C3 + 0.9 us, the C5 shard + 10 us), and the odd store has to be write-through and temporal (sc1 or sc0 sc1; plain
//                        write-back + 1.5 us at the C5 shard, nt alone nothing): sessions r8g - r8j, profiles/r06_r8g_r8j_mixed_store_policy.txt.
//   KSCHED_FUSED_THREADS threads per block (waves x 64)
#ifndef KSCHED_STORE_POLICY
#define KSCHED_STORE_POLICY 6
#endif
#ifndef KSCHED_STORE_ALT  // policy 6: which pod-row step of a round (0 .. 7) is stored without the nt hint
#define KSCHED_STORE_ALT(it) ((it) == 7u)
#endif
#ifndef KSCHED_FUSED_THREADS
#define KSCHED_FUSED_THREADS 1024
#endif
//   KSCHED_PROFILE       1 = diagnostics build (tools/trace_fused.py --profile): wave 0 of every block accumulates the core
//                        cycles it spends in phase 2, in the operand wait and in phase 1, and leaves them in trace words 1..4
#ifndef KSCHED_PROFILE
#define KSCHED_PROFILE 0
#endif
//   KSCHED_FUSED_WPE     experiment (tools/build_variants.sh): waves per SIMD the compiler must leave room for (5 = at most 96 VGPRs instead of 128:
//                        a 16-wave block then leaves a CU's SIMDs 128 registers each for the waves of ANOTHER kernel)
#if defined(KSCHED_FUSED_NUM_VGPR)
#define KSCHED_FUSED_WPE_ATTR __attribute__((amdgpu_num_vgpr(KSCHED_FUSED_NUM_VGPR)))
#elif defined(KSCHED_FUSED_WPE)
#define KSCHED_FUSED_WPE_ATTR __attribute__((amdgpu_waves_per_eu(KSCHED_FUSED_WPE, KSCHED_FUSED_WPE)))
#else
#define KSCHED_FUSED_WPE_ATTR
#endif

namespace ksched {

constexpr uint32_t kFusedThreads = KSCHED_FUSED_THREADS;
constexpr uint32_t kFusedWaves = kFusedThreads / 64;

// Kernel arguments: plain scalars only (they live in SGPRs; keep this small).
struct FusedArgs {
    uint32_t W, pitch, tiles, rows, nkeys, ngroups;
    uint32_t row_zero, row_valid, row_cpu, row_mem, row_taint;
    uint32_t lab_off[8], lab_mx1[8];   // first eight label keys: byte offset of the row before id 1's, and lab_max + 1 (the id
                                       // of the key's all-zero row); further keys go through lab_meta
    const uint32_t *lab_meta;          // device copy of IndexedLayout::lab_base[32], lab_max[32]
    const uint64_t *zero64;            // eight zero bytes in device memory
    uint32_t p, units, chunks, run;    // units = ceil(p / 8); run = (chunk, tile) pairs per XCD
    uint32_t unit_q, unit_rem, tiles_rcp;  // units / chunks, units % chunks, floor(2^32 / tiles)
    uint32_t wave_major;               // interleaved order: 1 = stream index wave * chunks + chunk, 0 = chunk * waves + wave
    uint32_t u_stride;                 // units between a wave's consecutive rounds: 8 = every wave owns a contiguous pod range (blocked); chunks * waves * 8 =
                                       // the launch's rounds are dealt round-robin over its (chunk, wave) streams (interleaved: the chip writes ONE moving window)
    uint32_t off_aux, off_fit, off_lab, off_trow;  // LDS byte offsets of the regions after the bitmap rows
    uint32_t nlist, list_mask8, off_list, off_lrec;  // list keys (tile_index.hpp): count, which of the first eight columns are lists, LDS offsets
    uint32_t list_col[kMaxListKeys];                 // their label columns
    uint32_t debug;
    uint32_t has_tol;  // tolerations were given (g_ptol is not null)
    uint32_t pick_ppb, pick_waves;  // PICK == 1: pods whose sampled pick one block carries, and how many of its waves carry them (the others stage)
    uint64_t *pick_acc;             // PICK == 2: [ceil(p / 8)] accumulators, one per unit of eight pods (count << 40 | 8 x 5 feasible-draw bits), all zero between launches
    uint32_t off_park;              // PICK == 2: LDS byte offset of the per-wave park of the round's draws
    uint32_t off_pf;                // LDS byte offset of the 256-byte dump area of the operand prefetch, or 0xFFFFFFFF: no room, no prefetch
    uint64_t *trace;  // diagnostics: per-block phase timestamps (100 MHz), or nullptr
};

// LDS carve-up: [rows * 128 : bitmap rows][aux block: 2 search trees + 2 cnt tables (FIT)]
//               per wave x 64 pods: [16 B fit record (FIT)][16 B label rows 1..8 (SEL)][8 B taint rows (TAINT)]
//               [nlist * 6 KiB: the tile's list keys][per wave x 64 pods: 16 B list record]   (snapshots with list keys only)
constexpr uint32_t kPickAttempts = 5;  // draws per pod the tile-test pick (PICK == 2) handles: ATTEMPTS of src/main.rs:49
constexpr uint32_t kPickParkBytes = kFusedWaves * kPickAttempts * 64u * 4u;
constexpr uint32_t kPrefetchDumpBytes = 256u;  // one dword per lane: where the operand prefetch's LDS-DMA loads land (shared by the block's waves: nobody reads it)
inline uint32_t fused_lds_bytes(const IndexedLayout &l, bool fit, bool sel, bool taint, FusedArgs *a = nullptr, bool park = false) {
    uint32_t off = l.rows * 128u;
    const uint32_t off_aux = off;
    if (fit) off += kAuxWords * 8u;
    const uint32_t off_fit = off;
    if (fit) off += kFusedWaves * 64u * 16u;
    const uint32_t off_lab = off;
    if (sel) off += kFusedWaves * 64u * 16u;
    const uint32_t off_trow = off;
    if (taint) off += kFusedWaves * 64u * 8u;
    const uint32_t off_list = off;
    if (sel && l.nlist) off += l.nlist * kListBytes;
    const uint32_t off_lrec = off;
    if (sel && l.nlist) off += kFusedWaves * 64u * kListRecBytes;
    const uint32_t off_park = off;
    if (park) off += kPickParkBytes;
    // the operand prefetch's dump area: only where it fits (it never decides whether the fused kernel applies)
    uint32_t off_pf = 0xFFFFFFFFu;
    if (off + kPrefetchDumpBytes <= kLdsBudget) {
        off_pf = off;
        off += kPrefetchDumpBytes;
    }
    if (a) {
        a->off_pf = off_pf;
        a->off_park = off_park;
        a->off_list = off_list;
        a->off_lrec = off_lrec;
        a->off_aux = off_aux;
        a->off_fit = off_fit;
        a->off_lab = off_lab;
        a->off_trow = off_trow;
    }
    return off;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));

// LIST: the snapshot keeps some label keys as per-tile sorted lists (high-cardinality keys, tile_index.hpp).  A separate
// instantiation, so that snapshots without such keys run exactly the code they ran before.
// PICK: the sampled pick of select_node_for_pod (src/main.rs:51-71) rides in this launch: the (chunk, tile) blocks share the
// batch's pods evenly, and while a block's tile is being staged its first `pick_waves` waves run select_one_pod
// (kernels_direct.hpp: the drawn candidates tested from the 64-byte node records, exactly what k_select_sampled does) for the
// block's pods instead of issuing staging pieces -- the pick's chain of dependent memory round trips overlaps the fill, which
// every block has to sit through anyway, and a step is ONE kernel.  The pick does not read the mask and the mask code below is
// the same with and without it.
//
// PICK == 2, the tile-test pick: a drawn candidate that lies in the block's tile is tested against the bitmap rows the block has in
// LDS anyway -- the (pod, node) bit the block is about to write into the mask, read straight from the rows phase 1 has just named --
// so no node record is fetched and no wave is taken off the staging.  Every (chunk, tile) block sees all five draws of its
// chunk's pods and tests the ones that fall into its tile (lane = pod, phase 1); what it found goes into a 64-bit accumulator per
// unit of eight pods with ONE returning atomic add per unit and block: five bits per pod = "draw i is feasible" (a draw lies in
// exactly one tile, so the blocks' bit sets are disjoint and the add is an OR), bits 40.. = how many blocks have contributed.  The
// block whose add returns tiles - 1 is the unit's last contributor: it holds every bit, takes each pod's lowest set one (first
// feasible draw wins, src/main.rs:61-65), writes the bindings -- the drawn nodes, from the draws it parked in LDS -- and zeroes the
// accumulator for the next launch.  The atomic is issued at the top of the NEXT trip and awaited by that trip's counted wait, like the operand loads: its
// return registers are in flight inside one trip only.
template <bool FIT, bool SEL, bool TAINT, bool WANT_FIT, bool LIST = false, int PICK = 0>
__global__ __launch_bounds__(kFusedThreads) KSCHED_FUSED_WPE_ATTR void k_eval_fused(
    const uint64_t *__restrict__ g_tables, const uint64_t *__restrict__ g_aux, const int64_t *__restrict__ g_pcpu,
    const int64_t *__restrict__ g_pmem, const uint32_t *__restrict__ g_psel, const uint64_t *__restrict__ g_ptol,
    uint64_t *__restrict__ out_feas, uint64_t *__restrict__ out_fit, const uint8_t *__restrict__ g_list, const FusedArgs a,
    const SelectArgs sa) {
    static_assert(!LIST || SEL, "list keys only exist with the selector predicate");
    static_assert(!PICK || (!LIST && !WANT_FIT), "the pick rides only in the plain mask variants");
    static_assert(PICK != 2 || !TAINT, "the tile-test pick is built for the reference's two predicates");
    kernarg_warm<9 * 8 + sizeof(FusedArgs) + (PICK ? sizeof(SelectArgs) : 0)>();  // the prologue makes four dependent groups of argument loads (kernarg.hpp)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t b = blockIdx.x;
    // (chunk, tile) pairs in chunk-major order are dealt to the XCDs in contiguous runs: XCD x = block id % 8
    // (observed dispatch order; speed only) takes pairs [x * run, (x + 1) * run), so the tile-blocks of one
    // pod range sit on one XCD (at most one seam per XCD boundary) and share its L2.
    uint32_t tile, chunk, lin;  // lin = chunk * tiles + tile: the block's number among the launch's live blocks
    if (!(a.debug & 32u)) {
        const uint32_t l = (b & 7u) * a.run + (b >> 3);
        if ((b >> 3) >= a.run) return;
        chunk = __umulhi(l, a.tiles_rcp);  // l / tiles by a host-made reciprocal (no division sequence in the prologue)
        tile = l - chunk * a.tiles;
        if (tile >= a.tiles) {  // the reciprocal can be one short
            tile -= a.tiles;
            ++chunk;
        }
        lin = l;
    } else {  // experiment: plain round-robin of (tile, chunk) pairs
        tile = b % a.tiles;
        chunk = b / a.tiles;
        lin = chunk * a.tiles + tile;
    }
    if (chunk >= a.chunks) return;

    // PICK: which waves of the block carry the pick (wave-uniform).  Default: the first `pick_waves` (they are launched first, so
    // the pick's chain of round trips gets a head start); debug bit 0x10000 gives it to the last ones instead (A/B).
    const uint32_t pick_waves = PICK == 1 ? a.pick_waves : 0u;
    uint32_t tid = threadIdx.x;
    bool pick_wave = false;
    if constexpr (PICK == 1) {
        // The pick runs HERE, ahead of everything the mask code keeps in registers: select_one_pod holds up to five 48-byte
        // candidates per lane, and next to the main loop's per-lane invariants that would not fit the 128 VGPRs of a 1024-thread
        // block (it spilled when it sat at the loop's first trip).  The `tid` the rest of the kernel derives its per-lane values
        // from passes through an asm statement placed after the pick, so none of them can be computed (and live) before it.
        const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const uint32_t pick_rank = (a.debug & 0x10000u) ? (kFusedWaves - 1u - wave0) : wave0;
        pick_wave = pick_rank < pick_waves;
        if (pick_wave) {
            const uint64_t t_in = (a.trace && threadIdx.x == 0u) ? wall_clock64() : 0ull;  // diagnostics (KSCHED_OPT_TRACE)
            // this block's share of the batch's pods: [lin * ppb, (lin + 1) * ppb), 64 at a time over the pick waves
            const uint32_t base = lin * a.pick_ppb;
            const uint32_t end = min(a.p, base + a.pick_ppb);
            const uint32_t eager = (a.debug >> 8) & 3u;  // A/B of the number of eagerly fetched draws (same bits as the stand-alone kernel)
            for (uint32_t pod = base + pick_rank * 64u + (threadIdx.x & 63u); pod < end; pod += pick_waves * 64u) {
                // src/main.rs:53-66: first feasible draw wins, none -> -1
                int32_t bnd;
                if (eager == 1u) bnd = select_one_pod<5, 3>(sa, pod);
                else if (eager == 2u) bnd = select_one_pod<5, 5>(sa, pod);
                else if (eager == 3u) bnd = select_one_pod<5, 2>(sa, pod);
                else bnd = select_one_pod<5, 1>(sa, pod);
                sa.binding[pod] = bnd;
            }
            if (a.trace && threadIdx.x == 0u)  // trace word 7, bits 8..: wave 0's entry -> its picks issued, in 10 ns ticks (bits 0..7: XCC id)
                atomicOr((unsigned long long *)&a.trace[(size_t)b * 8u + 7u], (unsigned long long)((wall_clock64() - t_in) << 8));
        }
        asm volatile("" : "+v"(tid) : : "memory");
    }
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar control flow below
    const uint32_t stage_rank = (PICK == 1 && !(a.debug & 0x10000u)) ? wave - pick_waves : wave;  // of a staging wave among the staging waves
    const uint32_t stage_waves = kFusedWaves - pick_waves;
    const bool tracer = a.trace && tid == (PICK == 1 ? (kFusedWaves - 1u) * 64u : 0u);  // (PICK == 1: wave 0 carries picks; trace a staging wave)
    auto stamp = [&](uint32_t i) {
        if (tracer) a.trace[(size_t)b * 8u + i] = wall_clock64();
    };
    stamp(0);
    // this wave's units [u, u_hi): chunk range cut evenly over the waves
    // balanced split of `units` over `chunks`: the first unit_rem chunks get unit_q + 1 units
    const uint32_t c_lo = chunk * a.unit_q + min(chunk, a.unit_rem);
    const uint32_t c_n = a.unit_q + (chunk < a.unit_rem ? 1u : 0u);
    uint32_t u = c_lo + (wave * c_n) / kFusedWaves;
    uint32_t u_hi = c_lo + ((wave + 1u) * c_n) / kFusedWaves;
    const uint32_t u_stride = a.u_stride;  // (a kernel argument: uniform; the back end may keep it or load it again)
    if (u_stride != 8u) {
        // interleaved (the default): round g of the launch (64 pods) belongs to stream g mod (chunks * waves); stream = wave * chunks + chunk
        // (wave-major: neighbouring rounds are written by different blocks, hence different compute units and mostly different XCDs) or
        // chunk * waves + wave (chunk-major: a block's sixteen waves write sixteen neighbouring rounds).  The chip then writes ONE moving window
        // of chunks * 16 rounds instead of chunks * 16 streams that each sweep a range of their own megabytes apart: session r6j / r6k, same box,
        // blocked -> chunk-major -> wave-major: C3 19.66 -> 18.73 -> 18.0 us per step, C4 shard 40.9 -> 38.8 -> 37.5, the C5 shard's mask kernel
        // 173 -> 165 -> 152 us (profiles/r06_round_order.md).  Every tile-block of a chunk still sees the same pods in the same units (the
        // tile-test pick's accumulators count on it); only the round at the very end of the batch can be short, and it is its wave's last.
        u = (a.wave_major ? wave * a.chunks + chunk : chunk * kFusedWaves + wave) * 8u;
        u_hi = a.units;
    }

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
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0;  // PICK == 2: the pod's five draws
    uint64_t ar = 0;                                  // PICK == 2: what the pod's accumulator held before this block's add
    auto issue_ops = [&](uint32_t pod) {
        const uint32_t pc = min(pod, a.p - 1u);  // clamp: lanes past the end read a valid row and are masked later
        if constexpr (PICK == 2) {
            static_assert(kPickAttempts == 5, "five draw registers");
            // The round's draws -- 64 pods x 5 = 320 consecutive dwords -- as five COALESCED loads: lane l takes dwords l, 64 + l, ... of the
            // block (a load per draw, lane = pod, touched twenty lines per instruction: the draws were two thirds of the round's line requests,
            // 3 us per 100 k pods once a launch is long enough to be bound by its rounds); phase 1 transposes them through the park.
            const size_t last = (size_t)a.p * kPickAttempts - 1u;
            const size_t f0 = (size_t)(pod - lane) * kPickAttempts + lane;  // (pod - lane = the round's first pod)
            asm volatile("global_load_dword %0, %1, off" : "=v"(d0) : "v"(sa.samples + min(f0, last)) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(d1) : "v"(sa.samples + min(f0 + 64u, last)) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(d2) : "v"(sa.samples + min(f0 + 128u, last)) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(d3) : "v"(sa.samples + min(f0 + 192u, last)) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(d4) : "v"(sa.samples + min(f0 + 256u, last)) : "memory");
        }
        if (FIT) {
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rc) : "v"(g_pcpu + pc) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rm) : "v"(g_pmem + pc) : "memory");
        }
        if (SEL) {
            // columns the snapshot does not have (K >= nkeys) read the zero word: "unconstrained" without a mask later
#define KSCHED_LOAD_SEL(K, DST)                                                                                                   \
    asm volatile("global_load_dword %0, %1, off"                                                                                  \
                 : "=v"(DST)                                                                                                      \
                 : "v"((uint32_t)K < a.nkeys ? g_psel + (size_t)K * a.p + pc : reinterpret_cast<const uint32_t *>(a.zero64)) \
                 : "memory")
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
            const uint64_t *tp = a.has_tol ? g_ptol + pc : a.zero64;
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
    // the same with the tile-test pick's in-flight registers: the five draws and the accumulator's returned value
#define KSCHED_WAIT_OPS_PICK(N)                                                                                                       \
    asm volatile("s_waitcnt vmcnt(%c17)"                                                                                              \
                 : "+v"(rc), "+v"(rm), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7), "+v"(tol), \
                   "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(ar)                                                     \
                 : "n"(N)                                                                                                             \
                 : "memory")
    // aux block of the tile: [tree cpu 1024][tree mem 1024][cnt cpu kCntEntries][cnt mem kCntEntries] (8-byte words)
    const int64_t *s_cpu = reinterpret_cast<const int64_t *>(smem + a.off_aux);
    const int64_t *s_mem = s_cpu + kAuxTreeWords;
    const uint2 *s_cnt_cpu = reinterpret_cast<const uint2 *>(s_cpu + 2u * kAuxTreeWords);
    const uint2 *s_cnt_mem = s_cnt_cpu + kCntEntries;
    // per-wave records of the current round (one entry per pod of the round)
    // (row "offsets" are LDS byte offsets of chunk 0 of the row: row * 128; the rows a record can name sit below 64 KiB)
    uint4 *s_fit = reinterpret_cast<uint4 *>(smem + a.off_fit) + wave * 64u;   // cnt[rank]: 8 bytes of cpu (x, y), 8 of memory (z, w)
    uint4 *s_lab = reinterpret_cast<uint4 *>(smem + a.off_lab) + wave * 64u;   // row offsets of the pod's constrained keys 1..8 (8 x 16 bit)
    uint2 *s_trow = reinterpret_cast<uint2 *>(smem + a.off_trow) + wave * 64u;  // four taint row offsets per pod
    // (the list code addresses LDS through an explicit local-address-space pointer: with generic pointers one TAINT + LIST
    // instantiation ran into a compiler back-end error, "Illegal instruction ... V_CMP_NE_U32 0, $src_shared_base")
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    lds_u8 *const lds = (lds_u8 *)smem;
    uint2 *s_lrec = reinterpret_cast<uint2 *>(smem + a.off_lrec) + wave * 64u;  // LIST: per list key (first entry | count << 16); count 0xFFFF = unconstrained
    constexpr uint32_t kListCap = 8u;  // the unchecked phase 2 walks at most this many entries per list key; longer ranges take the checked path

    // phase-2 lane layout (file header): quad q = (lane & 31) >> 2 -> pod 0,2,2,0,3,1,1,3 of the half, chunk base
    // 0,0,4,4,0,0,4,4; `sub` = pod of the 8-pod step, `wp` = chunk (sub-tile) of the row
    const uint32_t quad = (lane & 31u) >> 2;
    const uint32_t wp = ((0xCCu >> quad) & 1u) * 4u + (lane & 3u);
    const uint32_t sub = (lane >> 5) * 4u + ((0x31130220u >> (4u * quad)) & 15u);
    const uint32_t w0 = tile * kTileWords + 2u * wp;  // first mask word of this lane in phase 2
    // a lane may store 16 bytes when both words lie inside the row pitch (words in [W, pitch) are padding)
    const bool has0 = w0 < a.pitch, has1 = w0 + 1u < a.pitch;
    const bool tile_full = (tile + 1u) * kTileWords <= a.pitch;  // block-uniform
    const bool taint_inline = a.ngroups <= 4u;
    const uint8_t *Tb = smem + wp * 16u;  // this lane's chunk of row 0
    const uint8_t *Tb_cpu = Tb + a.row_cpu * 128u;  // ... of the first fit row of cpu; memory's rows follow at a constant distance (tile_index.hpp)
    auto ldoff = [&](uint32_t off) -> u32x4 { return *reinterpret_cast<const u32x4 *>(Tb + off); };  // off = row * 128
    auto ldrow = [&](uint32_t row) -> u32x4 { return ldoff(row * 128u); };
    auto lo16 = [](uint32_t x) { return x & 0xFFFFu; };
    auto hi16 = [](uint32_t x) { return x >> 16; };
    // This lane's view of the records of pod `sub` of step 0; step `it` is a constant distance further (it * 8 pods), which
    // folds into the LDS instructions' offset fields (pointer + constant, never index arithmetic on the lane part).
    const uint8_t *fit_lane = reinterpret_cast<const uint8_t *>(s_fit + sub) + wp;  // [0] = cnt byte of cpu, [8] = of memory (one per sub-tile)
    const uint2 *lab_lane = reinterpret_cast<const uint2 *>(s_lab + sub);           // [0] = row offsets of keys 1..4, [1] = keys 5..8
    const uint2 *trow_lane = s_trow + sub;
    // LIST: the nodes of this tile that carry the pod's value of a list key sit at entries [first, first + count) of the key's
    // sorted list; this lane owns sub-tile `wp` (128 nodes): set the bits of the entries that fall into it.  `bound` is the
    // (wave-uniform) number of entries to look at; entries past the pod's own count are masked.
    // (The list code derives its per-lane addresses from opaque copies of `sub` / `wp` at the point of use: as loop invariants
    // of the main loop they would be hoisted into registers that live through every round, and the widest instantiation
    // -- FIT, SEL, TAINT, WANT_FIT, LIST -- would spill; tools/audit_asm.py allows no spills.)
    auto opaque = [](uint32_t x) -> uint32_t {
        asm volatile("" : "+v"(x));
        return x;
    };
    auto lrec_of = [&](uint32_t it) -> uint2 { return (s_lrec + opaque(sub) + it * 8u)[0]; };
    auto list_mask = [&](uint32_t j, uint32_t rec, uint32_t bound, uint32_t wp) -> u32x4 {
        const uint32_t noff = a.off_list + j * kListBytes + kTileNodes * 4u;  // LDS byte offset of the key's node numbers
        const uint32_t first = rec & 0xFFFFu, count = rec >> 16;
        u32x4 m = {0u, 0u, 0u, 0u};
        const uint32_t n_e = (bound != 0xFFFFFFFFu) ? bound : ((count == 0xFFFFu) ? 0u : count);  // 0xFFFFFFFF: this lane's own count
        for (uint32_t e = 0; e < n_e; ++e) {
            const uint32_t node = *(const __attribute__((address_space(3))) uint16_t *)(lds + noff + min(first + e, (uint32_t)kTileNodes - 1u) * 2u);
            const bool hit = e < count && (node >> 7) == wp;
            const uint32_t bit = hit ? (1u << (node & 31u)) : 0u, w = (node >> 5) & 3u;
            m.x |= (w == 0u) ? bit : 0u;
            m.y |= (w == 1u) ? bit : 0u;
            m.z |= (w == 2u) ? bit : 0u;
            m.w |= (w == 3u) ? bit : 0u;
        }
        const uint32_t all = (count == 0xFFFFu) ? 0xFFFFFFFFu : 0u;  // the pod does not constrain this key
        m.x |= all, m.y |= all, m.z |= all, m.w |= all;
        return m;
    };
    auto apply_lists = [&](u32x4 f, const uint2 lr, uint32_t bound) -> u32x4 {
        static_assert(kMaxListKeys == 2 && kListRecBytes == 8, "one 32-bit record slot per list key");
        const uint32_t r[2] = {lr.x, lr.y};
        const uint32_t wp_o = opaque(wp);
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j)
            if (j < a.nlist) f &= list_mask(j, r[j], bound, wp_o);
        return f;
    };

    // Row loads of one pod-row of phase 2 (issued together, consumed later: two iterations are
    // interleaved by hand so that many LDS reads are in flight per wave).
    struct Rows {
        u32x4 c0, m0, l0, l1, l2, l3, t0, t1, t2, t3, x0, x1, x2, x3;
    };
    auto load_extra = [&](const uint2 r2, Rows &R) {  // label rows 5..8 (rounds where some pod constrains more than four keys)
        R.x0 = ldoff(lo16(r2.x));
        R.x1 = ldoff(hi16(r2.x));
        R.x2 = ldoff(lo16(r2.y));
        R.x3 = ldoff(hi16(r2.y));
    };
    // cc, cm: this lane's cnt bytes (the row {lr >= cnt[rank][wp]} of each resource); lb, tr: record halves
    auto load_rows = [&](const uint32_t cc, const uint32_t cm, const uint2 lb, const uint2 tr, Rows &R) {
        if (FIT) {
            R.c0 = *reinterpret_cast<const u32x4 *>(Tb_cpu + cc * 128u);
            R.m0 = *reinterpret_cast<const u32x4 *>(Tb_cpu + cm * 128u + (uint32_t)kFitRows * 128u);  // the distance folds into the read's offset field
        } else {
            R.c0 = ldrow(a.row_valid);
        }
        if (SEL) {
            R.l0 = ldoff(lo16(lb.x));
            R.l1 = ldoff(hi16(lb.x));
            R.l2 = ldoff(lo16(lb.y));
            R.l3 = ldoff(hi16(lb.y));
        }
        if (TAINT) {
            R.t0 = ldoff(lo16(tr.x));
            R.t1 = ldoff(hi16(tr.x));
            R.t2 = ldoff(lo16(tr.y));
            R.t3 = ldoff(hi16(tr.y));
        }
    };
    // records of pod row `it * 8 + sub` of the round, as this lane needs them
    auto rec_cc = [&](uint32_t it) -> uint32_t { return FIT ? (uint32_t)(fit_lane + it * 128u)[0] : 0u; };
    auto rec_cm = [&](uint32_t it) -> uint32_t { return FIT ? (uint32_t)(fit_lane + it * 128u)[8] : 0u; };
    auto rec_lb = [&](uint32_t it) -> uint2 { return SEL ? (lab_lane + it * 16u)[0] : make_uint2(0u, 0u); };
    auto rec_lx = [&](uint32_t it) -> uint2 { return (lab_lane + it * 16u)[1]; };
    auto rec_tr = [&](uint32_t it) -> uint2 { return TAINT ? (trow_lane + it * 8u)[0] : make_uint2(0u, 0u); };
    auto fit_of = [&](const Rows &R) -> u32x4 {
        if (FIT) return R.c0 & R.m0;  // pos >= rank, both resources
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
#elif KSCHED_STORE_POLICY == 5 || KSCHED_STORE_POLICY == 6
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(dst + o), "v"(f) : "memory");
#else
        *reinterpret_cast<u32x4_a8 *>(dst + o) = f;
#endif
    };
    // Unchecked stores of a round: wave-uniform base of the step's first pod row (SGPR pair, scalar arithmetic) + this
    // lane's constant 32-bit byte offset: no per-lane address arithmetic at all (it was a 64-bit multiply-add per store).
    const uint32_t lane_off = (sub * a.pitch + w0) * 8u;  // of pod `sub` within a step
    const uint32_t step_bytes = a.pitch * 64u;            // 8 pod rows
    auto store_rel = [&](uint64_t base, uint32_t voff, const u32x4 f, bool alt = false) {
#if KSCHED_STORE_POLICY == 6  // the two write-through forms mixed: `alt` = this pod-row step goes without the nt hint (file header)
        if (alt) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(f), "s"(base) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt\n\ts_nop 1" ::"v"(voff), "v"(f), "s"(base) : "memory");
#elif KSCHED_STORE_POLICY == 2
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(f), "s"(base) : "memory");
#elif KSCHED_STORE_POLICY == 3
        asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(f), "s"(base) : "memory");
#elif KSCHED_STORE_POLICY == 4
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt\n\ts_nop 1" ::"v"(voff), "v"(f), "s"(base) : "memory");
#elif KSCHED_STORE_POLICY == 5
        asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt\n\ts_nop 1" ::"v"(voff), "v"(f), "s"(base) : "memory");
#elif KSCHED_STORE_POLICY == 1
        __builtin_nontemporal_store(f, reinterpret_cast<u32x4_a8 *>(reinterpret_cast<uint8_t *>(base) + voff));
#else
        *reinterpret_cast<u32x4_a8 *>(reinterpret_cast<uint8_t *>(base) + voff) = f;
#endif
    };
    auto uniform64 = [](uint64_t x) -> uint64_t {  // provably wave-uniform (an SGPR pair for the asm operand)
        // (the builtin returns a signed int: widen through uint32_t, or the low half sign-extends into the high one)
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)), lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
        return ((uint64_t)hi << 32) | (uint64_t)lo;
    };
    uint64_t rb_feas = 0, rb_fit = 0;  // bases of the round in flight (set at the top of phase 2)
    auto combine_store = [&](uint32_t it, const Rows &R, bool extra) {  // pod row `it * 8 + sub` of the round
        const uint64_t step = (uint64_t)it * step_bytes;  // scalar: the step moves the SGPR base, the lane offset stays put
        u32x4 f = fit_of(R);
        if (WANT_FIT) store_rel(rb_fit + step, lane_off, f);
        if (SEL) f = ((f & R.l0) & R.l1) & (R.l2 & R.l3);
        if (SEL && extra) f = ((f & R.x0) & R.x1) & (R.x2 & R.x3);
        if (TAINT) f = ((f & R.t0) & R.t1) & (R.t2 & R.t3);
        store_rel(rb_feas + step, lane_off, f, KSCHED_STORE_ALT(it));  // out_feas is never null here (the API supplies a scratch mask when the caller gives none)
    };

    // Checked form of one pod-row: end-of-range / partial-tile predicates and the overflow walks
    // (more than eight constrained keys, more than four taint groups).  Rare, not unrolled.
    auto emit_checked = [&](uint32_t pod, uint32_t it, bool over) {
        const bool live = pod < a.p && has0;
        Rows R;
        if (LIST) {  // record addresses from opaque copies (see `opaque` below): no loop-invariant registers in this variant
            uint32_t sub_o = sub, wp_o = wp;
            asm volatile("" : "+v"(sub_o), "+v"(wp_o));
            const uint8_t *fit_o = reinterpret_cast<const uint8_t *>(s_fit + sub_o) + wp_o;
            const uint2 *lab_o = reinterpret_cast<const uint2 *>(s_lab + sub_o);
            load_rows(FIT ? (uint32_t)(fit_o + it * 128u)[0] : 0u, FIT ? (uint32_t)(fit_o + it * 128u)[8] : 0u, (lab_o + it * 16u)[0],
                      TAINT ? (s_trow + sub_o + it * 8u)[0] : make_uint2(0u, 0u), R);
            load_extra((lab_o + it * 16u)[1], R);
        } else {
            load_rows(rec_cc(it), rec_cm(it), rec_lb(it), rec_tr(it), R);
            if (SEL) load_extra(rec_lx(it), R);
        }
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
                    if (LIST && a.lab_meta[k] == kLabList) continue;  // list keys have no rows (applied below)
                    const uint32_t s = g_psel[(size_t)k * a.p + pod];
                    if (s != 0u) f &= ldrow((s <= a.lab_meta[32u + k]) ? (a.lab_meta[k] + s - 1u) : a.row_zero);
                }
            }
        }
        if (TAINT) {
            if (taint_inline) {
                f = ((f & R.t0) & R.t1) & (R.t2 & R.t3);
            } else if (live) {
                const uint64_t t = a.has_tol ? g_ptol[pod] : 0ull;
                for (uint32_t g = 0; g < a.ngroups; ++g) f &= ldrow(a.row_taint + 16u * g + (uint32_t)((t >> (4u * g)) & 15ull));
            }
        }
        if (LIST) {
            __builtin_amdgcn_sched_barrier(0);  // the row registers are dead by now: keeps this variant inside the VGPR budget
            f = apply_lists(f, lrec_of(it), 0xFFFFFFFFu);  // every entry of this pod's ranges, however long
        }
        if (live && out_feas && !(a.debug & 1u)) store(out_feas);
    };

    // ---- phase 1: lane = pod pod0 + lane (branch-free); returns the overflow ballot ---------------
    bool extra_any = false, list_any = false;
    uint32_t list_bound = 0;
    uint32_t pick_bits = 0;  // PICK == 2: which of the pod's draws this block found feasible in its tile (the round that was just prepared)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32 *const s_park = (lds_u32 *)(lds + a.off_park) + wave * (kPickAttempts * 64u);  // the round's draws, [draw][lane]
    auto phase1 = [&](uint32_t pod0) -> uint64_t {
        uint2 cnt_c = make_uint2(0u, 0u), cnt_m = make_uint2(0u, 0u);  // cnt[rank] of cpu / memory: one row number per sub-tile
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
            // cnt[rank]: for each sub-tile, how many of its nodes sit below the request = the row {lr >= cnt} to read
            const uint2 cc = s_cnt_cpu[lc], cm = s_cnt_mem[lm];
            s_fit[lane] = make_uint4(cc.x, cc.y, cm.x, cm.y);
            cnt_c = cc;
            cnt_m = cm;
        }
        const uint32_t rv = a.row_valid * 128u;  // offsets, not row numbers, from here on
        uint32_t cnt = 0;
        if (SEL) {
            // The record starts as eight times the all-valid row; the pod's j-th constrained key then overwrites slot j
            // with one 2-byte LDS store (LDS operations of a wave execute in order), instead of a register compaction
            // that costs a select per (key, slot) pair.
            const uint32_t rv2 = rv | (rv << 16);
            s_lab[lane] = make_uint4(rv2, rv2, rv2, rv2);
            uint16_t *const slots = reinterpret_cast<uint16_t *>(s_lab + lane);
            uint16_t *slot = slots;  // next free slot of this pod's record
            const uint32_t sv[8] = {s0, s1, s2, s3, s4, s5, s6, s7};
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                const uint32_t s = sv[k];
                if (s != 0u && !(LIST && ((a.list_mask8 >> k) & 1u))) {  // (list keys have no rows: handled below)
                    // value id s of key k -> its row; ids no node carries (KSCHED_SEL_NEVER, unknown) clamp to the key's all-zero row
                    *slot++ = (uint16_t)(min(s, a.lab_mx1[k]) * 128u + a.lab_off[k]);
                }
            }
            cnt = (uint32_t)(slot - slots);
            if (a.nkeys > 8u) {  // keys 9.. : any constraint there sends the pod down the overflow walk
                const uint32_t pc = min(pod0 + lane, a.p - 1u);
                for (uint32_t k = 8; k < a.nkeys; ++k) {
                    if (LIST && a.lab_meta[k] == kLabList) continue;
                    cnt += (g_psel[(size_t)k * a.p + pc] != 0u) ? 9u : 0u;
                }
            }
        }
        uint32_t lmax = 0;
        if (LIST) {
            // list keys: the pod's id -> the range of the tile's sorted list that carries it (two branch-free lower bounds over
            // 1024 entries in LDS); only lanes that constrain the key search
            // The pod's ids of the list keys: a list key among the first eight columns is already in the pipelined operand
            // registers (picked by the wave-uniform column number); one beyond them is read here (a compiler-issued load: it waits
            // for the wave's outstanding stores -- only snapshots with more than eight label keys AND a list key among the later ones)
            uint32_t lv[kMaxListKeys] = {0u, 0u};
#pragma unroll
            for (uint32_t j = 0; j < kMaxListKeys; ++j) {
                if (j >= a.nlist) continue;
                const uint32_t col = a.list_col[j];
                if (col < 8u) {
                    uint32_t v = s0;
                    v = (col == 1u) ? s1 : v;
                    v = (col == 2u) ? s2 : v;
                    v = (col == 3u) ? s3 : v;
                    v = (col == 4u) ? s4 : v;
                    v = (col == 5u) ? s5 : v;
                    v = (col == 6u) ? s6 : v;
                    v = (col == 7u) ? s7 : v;
                    lv[j] = v;
                } else {
                    lv[j] = g_psel[(size_t)col * a.p + min(pod0 + lane, a.p - 1u)];
                }
            }
            uint32_t rec[kMaxListKeys] = {0xFFFF0000u, 0xFFFF0000u};
            bool any = false;
#pragma unroll
            for (uint32_t j = 0; j < kMaxListKeys; ++j) {
                rec[j] = 0xFFFF0000u;  // unconstrained
                const uint32_t s = lv[j];
                if (j < a.nlist && s != 0u) {
                    const uint32_t voff = a.off_list + j * kListBytes;  // LDS byte offset of the key's sorted ids
                    auto val_at = [&](uint32_t e) -> uint32_t { return *(const __attribute__((address_space(3))) uint32_t *)(lds + voff + e * 4u); };
                    auto lower = [&](uint32_t key) -> uint32_t {  // number of entries below `key`
                        uint32_t base = 0;
#pragma unroll
                        for (uint32_t half = (uint32_t)kTileNodes / 2u; half >= 1u; half >>= 1) base += (val_at(base + half - 1u) < key) ? half : 0u;
                        return base + ((val_at(base) < key) ? 1u : 0u);
                    };
                    const uint32_t lo = lower(s);
                    // the length of the run of `s` from there: the next kListCap + 1 entries are read at once (independent reads,
                    // one latency) instead of a second dependent search; only a longer run pays for the second search
                    uint32_t cnt = 0;
                    bool run = true;
#pragma unroll
                    for (uint32_t e = 0; e <= kListCap; ++e) {
                        run = run && lo + e < (uint32_t)kTileNodes && val_at(min(lo + e, (uint32_t)kTileNodes - 1u)) == s;
                        cnt += run ? 1u : 0u;
                    }
                    if (cnt > kListCap) cnt = lower(s + 1u) - lo;  // (ids are < KSCHED_SEL_NEVER here: SEL_NEVER itself is carried by no node, cnt = 0)
                    rec[j] = lo | (cnt << 16);
                    lmax = max(lmax, cnt);
                    any = true;
                }
            }
            s_lrec[lane] = make_uint2(rec[0], rec[1]);
            list_any = __ballot(any) != 0ull;
#pragma unroll
            for (uint32_t d = 32; d >= 1; d >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, d, 64));
            list_bound = (uint32_t)__builtin_amdgcn_readfirstlane((int)lmax);
        }
        if (TAINT) {
            uint32_t t[4];
#pragma unroll
            for (uint32_t g = 0; g < 4; ++g)
                t[g] = (g < a.ngroups) ? (a.row_taint + 16u * g + (uint32_t)((tol >> (4u * g)) & 15ull)) * 128u : rv;
            s_trow[lane] = make_uint2(t[0] | (t[1] << 16), t[2] | (t[3] << 16));
        }
        if constexpr (PICK == 2) {
            // The tile-test pick: check_node_validity(pod, candidate) (src/predicates.rs:63-77) for the draws that fall into THIS
            // tile, read from the rows phase 2 is about to AND for the same pod: bit (l % 32) of word (l / 32) of every row the pod's
            // records name -- the fit rows of both resources by the candidate's sub-tile, the eight selector slots (unconstrained
            // slots name the all-valid row) -- is exactly the feasible bit this block writes for (pod, candidate).
            const uint4 slots = SEL ? s_lab[lane] : make_uint4(0u, 0u, 0u, 0u);  // (read back after this lane's own scatter stores: LDS operations of a wave execute in order)
            // the draws arrived as the round's block in linear order (issue_ops): through the park -- [pod of the round][draw], where the pod's
            // last contributor reads the winning draw's node one trip later -- every lane gets its own pod's five (stride 5 dwords: no bank conflict;
            // LDS operations of a wave execute in order)
            s_park[lane] = d0;
            s_park[64u + lane] = d1;
            s_park[128u + lane] = d2;
            s_park[192u + lane] = d3;
            s_park[256u + lane] = d4;
            uint32_t dv[kPickAttempts];
#pragma unroll
            for (uint32_t i = 0; i < kPickAttempts; ++i) dv[i] = s_park[lane * kPickAttempts + i];
            uint32_t bits = 0;
#pragma unroll
            for (uint32_t i = 0; i < kPickAttempts; ++i) {
                const uint32_t l = dv[i] - tile * (uint32_t)kTileNodes;  // in this tile <=> l < 1024 (unsigned; a draw >= n lies in no tile, or on padding bits, which are zero in every row)
                if (l < (uint32_t)kTileNodes && !(a.debug & 0x8000000u)) {  // (debug bit 27: no tests -- what the tests cost; results invalid)
                    const uint32_t wofs = (l >> 5) << 2, sub = l >> 7;
                    auto word = [&](uint32_t off) -> uint32_t { return *(lds_u32 *)(lds + off + wofs); };
                    uint32_t v;
                    if (FIT) {
                        const uint32_t sh = (sub & 3u) * 8u;
                        const uint32_t bc = ((sub < 4u ? cnt_c.x : cnt_c.y) >> sh) & 0xFFu, bm = ((sub < 4u ? cnt_m.x : cnt_m.y) >> sh) & 0xFFu;
                        v = word((a.row_cpu + bc) * 128u) & word((a.row_cpu + (uint32_t)kFitRows + bm) * 128u);  // src/predicates.rs:42, both resources
                    } else {
                        v = word(rv);
                    }
                    if (SEL)  // src/predicates.rs:45-61
                        v &= word(slots.x & 0xFFFFu) & word(slots.x >> 16) & word(slots.y & 0xFFFFu) & word(slots.y >> 16) & word(slots.z & 0xFFFFu) &
                             word(slots.z >> 16) & word(slots.w & 0xFFFFu) & word(slots.w >> 16);
                    bits |= ((v >> (l & 31u)) & 1u) << i;
                }
            }
            pick_bits = bits;
        }
        extra_any = __ballot(cnt > 4u) != 0ull;  // some pod of the round needs label rows 5..8
        const uint64_t over = __ballot(cnt > 8u || (LIST && lmax > kListCap));  // more than eight row keys constrained, or a long list range
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
    // ---- operand prefetch ------------------------------------------------------------------------------------------
    // A scheduler never evaluates the same batch twice, so a round's operand lines (64 pods x 68 bytes at C3: 34 lines of 128 bytes
    // over ten columns) come from HBM, not from the L2 an earlier launch left warm -- and the pipeline below issues a round's loads
    // only ONE trip ahead, behind a queue of mask stores: with the bench cycling over different resident batches the C3 step read
    // 24.2 us against 19.7 us with one batch evaluated over and over (session r5c).  So right behind the staging barrier -- not ahead
    // of it: the address arithmetic and the cold translations of these loads would sit in front of the staging pieces every wave
    // waits for -- the wave touches the lines of the rounds AFTER its first one: ONE load instruction per round, lane l fetching a
    // dword of line l of the round, which brings them into the XCD's L2 while rounds 0 and 1 are computed; the real loads of those
    // rounds then hit.  The touched values are never used and take no register: the loads are LDS-DMA loads (global_load_lds_dword)
    // into a 256-byte dump area of their own (FusedArgs::off_pf; no room for it in LDS = no prefetch), issued by inline asm so that
    // the compiler's wait bookkeeping does not see them; they are older than the next trip's operand loads, whose counted wait
    // therefore covers them (vector-memory operations return in order).  At most kPrefetchRounds rounds ahead; waves with more rounds
    // than the prefetch reaches skip it (a launch that long is bound by its stores, not by this latency).
    // (session r5d, inputs rotated: C3 23.9 -> 20.2 us per step, the C4 shard 54.8 -> 42.3; the C5 shard, 24 rounds per wave, 222.5 -> 224.7 with it)
    constexpr uint32_t kPrefetchRounds = 6;
    auto prefetch_rounds = [&]() {
        if (!(FIT || SEL || TAINT) || a.off_pf == 0xFFFFFFFFu || u + u_stride * (kPrefetchRounds + 2u) < u_hi || (a.debug & 0x20000000u)) return;
        const uint32_t dump = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.off_pf);
        // lane -> line of the round: [0,4) cpu, [4,8) memory, [8,24) the eight selector columns (two lines each), [24,34) the draws, [34,38) tolerations
        const uint32_t li = opaque(lane);
#pragma unroll 1
        for (uint32_t r = 1; r <= kPrefetchRounds; ++r) {
            const uint32_t pu = u + u_stride * r;
            if (pu >= u_hi) break;  // wave-uniform
            const uint32_t pod0 = pu * 8u;
            const uint32_t pe = a.p - 1u;
            // integer addresses throughout (no selects between pointers), the lane's index made opaque first: the compiler cannot fold the
            // selection into something wave-uniform that it would hand to the "v" operand as a scalar pair
            const uint32_t e8 = min(pod0 + (li & 3u) * 16u, pe);  // element of an 8-byte column: line (li & 3) of the round
            uint64_t a64 = reinterpret_cast<uint64_t>(a.zero64);  // lanes without a line of their own touch the zero word
            if (FIT && li < 4u) a64 = reinterpret_cast<uint64_t>(g_pcpu + e8);
            if (FIT && li >= 4u && li < 8u) a64 = reinterpret_cast<uint64_t>(g_pmem + e8);
            if (SEL && a.nkeys && li >= 8u && li < 24u) {
                const uint32_t k = min((li - 8u) >> 1, min(a.nkeys, 8u) - 1u);
                a64 = reinterpret_cast<uint64_t>(g_psel + (size_t)k * a.p + min(pod0 + (li & 1u) * 32u, pe));
            }
            if (PICK == 2 && li >= 24u && li < 34u) {
                a64 = reinterpret_cast<uint64_t>(sa.samples + min((size_t)pod0 * kPickAttempts + (li - 24u) * 32u, (size_t)a.p * kPickAttempts - 1u));
            }
            if (TAINT && a.has_tol && li >= 34u && li < 38u) {
                a64 = reinterpret_cast<uint64_t>(g_ptol + e8);
            }
            // (M0 = the LDS destination of an LDS-DMA load; the compiler manages M0 itself -- the staging -- so the statement
            // puts back what it found.  gfx9 wants one wait state between an SALU write of M0 and an LDS-DMA instruction that reads it, the
            // hardware does not interlock and the hazard recogniser does not look inside inline asm: the s_nop on either side of the load
            // is that wait state, for the statement's own M0 write and for the load against the restore -- ADVICE r5; tools/audit_asm.py checks it.)
            uint32_t m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_nop 0\n\ts_mov_b32 m0, %0\n\ts_nop 0" : "=&s"(m0_saved) : "v"(a64), "s"(dump) : "memory");
        }
    };
    bool more = u < u_hi, have_prev = false, first = true, stamped4 = false;
    uint32_t prev_u = 0, prev_nu = 0;
    uint64_t prev_over = 0;
    bool prev_extra = false, prev_list = false;
    uint32_t prev_bound = 0;
#if KSCHED_PROFILE
    uint64_t prof_p2 = 0, prof_wait = 0, prof_p1 = 0, prof_rounds = 0, prof_t = 0;
#define KSCHED_PROF(ACC)                                       \
    do {                                                       \
        const uint64_t now_ = __builtin_readcyclecounter();    \
        ACC += now_ - prof_t;                                  \
        prof_t = now_;                                         \
    } while (0)
#else
#define KSCHED_PROF(ACC) \
    do {                 \
    } while (0)
#endif
    // PICK == 2: what this block found for the pods of the round prepared in the previous trip goes into their accumulators.  ONE
    // 64-bit word per UNIT of eight pods -- bits [5j, 5j + 5) = pod j's feasible draws, bits 40.. = how many blocks have contributed --
    // and one returning atomic add per unit, issued by the unit's first lane with the eight lanes' fields ORed together: the tile
    // blocks of a chunk cut their pods into the same units, so a unit's pods always arrive together.  (One atomic per POD cost 1.5 us
    // per launch at C3 and 5.6 us at the C4 shard, session r3h: the returning device-scope atomics were most of the pick's price.)
    // `nu` = units of that round: its pods are lanes [0, 8 nu); phase 1 prepares 64 lanes whatever the round holds, and the lanes past
    // a short round's end are pods of the NEXT wave's range -- they contribute there, not here.
    auto pick_contribute = [&](uint32_t pod0, uint32_t nu) {
        const uint32_t n_live = min(nu * 8u, a.p - pod0);  // wave-uniform: the round's pods that exist
        const uint32_t j = lane & 7u;
        uint32_t lo = 0, hi = 0;  // this pod's five bits at [5j, 5j + 5) of a 40-bit field
        if (lane < n_live) {
            const uint64_t f = (uint64_t)pick_bits << (5u * j);
            lo = (uint32_t)f;
            hi = (uint32_t)(f >> 32);
        }
#pragma unroll
        for (uint32_t d = 1; d <= 4; d <<= 1) {  // OR over the unit's eight lanes
            lo |= (uint32_t)__shfl_xor((int)lo, (int)d, 64);
            hi |= (uint32_t)__shfl_xor((int)hi, (int)d, 64);
        }
        const uint64_t delta = (1ull << 40) | ((uint64_t)hi << 32) | (uint64_t)lo;
        uint64_t *const slot = a.pick_acc + ((pod0 >> 3) + (lane >> 3));
        // the units' first lanes only (EXEC is narrowed inside the statement: the compiler sees one unconditional definition of `ar`)
        const uint64_t lead = 0x0101010101010101ull & (n_live >= 64u ? ~0ull : ((1ull << n_live) - 1ull));
        uint64_t saved;
        // (device scope -- no sc1 --: what HIP's atomicAdd is; coherent across the XCDs and 1.2 us per launch cheaper at C3 than system scope)
        asm volatile("s_and_saveexec_b64 %1, %4\n\tglobal_atomic_add_x2 %0, %2, %3, off sc0\n\ts_mov_b64 exec, %1"
                     : "=v"(ar), "=&s"(saved)
                     : "v"(slot), "v"(delta), "s"(lead)
                     : "memory", "scc");
    };
    auto pick_decide = [&](uint32_t pod0, uint32_t nu) {
        const uint32_t n_live = min(nu * 8u, a.p - pod0);
        const uint32_t j = lane & 7u, first = lane & ~7u;
        // the unit's returned word, from its first lane
        const uint32_t wlo = (uint32_t)__shfl((int)(uint32_t)ar, (int)first, 64), whi = (uint32_t)__shfl((int)(uint32_t)(ar >> 32), (int)first, 64);
        const uint64_t word = ((uint64_t)whi << 32) | (uint64_t)wlo;
        if (lane < n_live && (uint32_t)(word >> 40) == a.tiles - 1u) {  // every tile's block has contributed: this block decides the unit's pods
            const uint32_t all = ((uint32_t)(word >> (5u * j)) | pick_bits) & ((1u << kPickAttempts) - 1u);
            int32_t bnd = -1;  // no drawn candidate is feasible: None -> NoNodeFound (src/main.rs:70,117)
            if (all) bnd = (int32_t)s_park[lane * kPickAttempts + (uint32_t)__builtin_ctz(all)];  // first feasible draw wins (src/main.rs:61-65)
            sa.binding[pod0 + lane] = bnd;
            if (j == 0u) a.pick_acc[(pod0 >> 3) + (lane >> 3)] = 0ull;  // ready for the next launch (nobody else touches the word any more in this one)
        }
    };
    while (true) {
        if (PICK == 2 && have_prev && !(a.debug & 0x10000000u)) pick_contribute(prev_u * 8u, prev_nu);  // (debug bit 28: no atomics -- what they cost; results invalid)
        if (more) issue_ops(u * 8u + lane);
        if ((a.debug & 128u) && have_prev && !stamped4) stamp(1);  // experiment: after the 2nd round's operand loads were issued
        if (first) {
            // stage the tile: bitmap rows + aux block, global -> LDS without a VGPR round trip
            // (global_load_lds_dwordx4: per-lane global address, LDS destination = wave-uniform base + lane*16)
            auto stage = [&](const void *gsrc, uint32_t lds_off, uint32_t bytes) {
                const uint8_t *g = static_cast<const uint8_t *>(gsrc);
                for (uint32_t off = stage_rank * 1024u; off < bytes; off += stage_waves * 1024u) {
                    if (off + lane * 16u < bytes)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + off + lane * 16u),
                                                         (__attribute__((address_space(3))) void *)(lds + lds_off + off), 16, 0, 0);
                }
            };
            if (!pick_wave) {
                if (FIT) stage(g_aux + (size_t)tile * kAuxWords, a.off_aux, kAuxWords * 8u);  // needed first (phase 1)
                if (LIST) stage(g_list + (size_t)tile * a.nlist * kListBytes, a.off_list, a.nlist * kListBytes);
                if (!(a.debug & 8u)) stage(g_tables + (size_t)tile * a.rows * kTileWords, 0u, a.rows * 128u);
            }
            if (!(a.debug & 128u)) stamp(1);
            // everything this wave has in flight lands before the barrier: its staging pieces and the first round's operand loads
            // (a pick wave issues no staging piece whose wait would cover them)
            if (PICK == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            prefetch_rounds();  // (behind the barrier: see "operand prefetch" above)
            if (!(a.debug & 128u)) stamp(2);
        }
        if (have_prev) {
            // ============ phase 2 of the previous round: 8 lanes per pod, 2 words per lane ===========
#if KSCHED_PROFILE
            prof_t = __builtin_readcyclecounter();
#endif
            const uint32_t pod0 = prev_u * 8u;
            rb_feas = uniform64(reinterpret_cast<uint64_t>(out_feas + (size_t)pod0 * a.pitch));
            if (WANT_FIT) rb_fit = uniform64(reinterpret_cast<uint64_t>(out_fit + (size_t)pod0 * a.pitch));
            // A short round (prev_nu < 8) is always the wave's last one: no operand loads are in flight behind
            // it, so the counted wait does not depend on how many stores it issues.
            const bool fast = (prev_nu == 8u || !more) && prev_over == 0ull && pod0 + prev_nu * 8u <= a.p && tile_full &&
                              (!TAINT || taint_inline) && !(a.debug & 16u);
            if (LIST && fast && (prev_list || prev_nu != 8u)) {  // (also the wave's short last round: one rolled loop less in this variant)
                // Some pod of the round constrains a list key: same unchecked rows, one at a time, plus the list bits
                // (LDS only: the round issues exactly the stores of the plain path, so the counted wait still holds).
                // (record addresses re-derived here from opaque copies, see `opaque` above: no loop-invariant registers)
                const uint32_t sub_o = opaque(sub), wp_o = opaque(wp);
                const uint8_t *fit_o = reinterpret_cast<const uint8_t *>(s_fit + sub_o) + wp_o;
                const uint2 *lab_o = reinterpret_cast<const uint2 *>(s_lab + sub_o);
                const uint2 *trow_o = s_trow + sub_o;
                // software pipeline by hand: the records and rows of pod row it + 1 are fetched before row `it` is combined, so the
                // LDS latency of one row hides under the list arithmetic of the previous one
                Rows A;
                uint2 lr = (s_lrec + sub_o)[0];
                load_rows(FIT ? (uint32_t)fit_o[0] : 0u, FIT ? (uint32_t)fit_o[8] : 0u, lab_o[0], TAINT ? trow_o[0] : make_uint2(0u, 0u), A);
#pragma unroll 1
                for (uint32_t it = 0; it < prev_nu; ++it) {
                    const uint64_t step = (uint64_t)it * step_bytes;
                    u32x4 f = fit_of(A);
                    if (WANT_FIT) store_rel(rb_fit + step, lane_off, f);
                    f = ((f & A.l0) & A.l1) & (A.l2 & A.l3);
                    if (TAINT) f = ((f & A.t0) & A.t1) & (A.t2 & A.t3);
                    if (prev_extra) {  // wave-uniform: some pod of the round constrains five to eight row keys
                        load_extra((lab_o + it * 16u)[1], A);
                        f = ((f & A.x0) & A.x1) & (A.x2 & A.x3);
                    }
                    const uint2 lr_now = lr;
                    const uint32_t nx = min(it + 1u, prev_nu - 1u);  // (the last trip re-reads its own row: harmless)
                    __builtin_amdgcn_sched_barrier(0);
                    lr = (s_lrec + sub_o + nx * 8u)[0];
                    load_rows(FIT ? (uint32_t)(fit_o + nx * 128u)[0] : 0u, FIT ? (uint32_t)(fit_o + nx * 128u)[8] : 0u, (lab_o + nx * 16u)[0],
                              TAINT ? (trow_o + nx * 8u)[0] : make_uint2(0u, 0u), A);
                    f = apply_lists(f, lr_now, prev_bound);
                    store_rel(rb_feas + step, lane_off, f, KSCHED_STORE_ALT(it));
                }
            } else if (fast && prev_nu == 8u) {
                // STEP pod rows per step; the records of the next step are fetched while this step's rows are combined.
                constexpr uint32_t STEP = TAINT ? 1u : 2u;  // as many as the row registers allow (no spills: tools/audit_asm.py)
                uint32_t cn[STEP], mn[STEP];
                uint2 ln[STEP], tn[STEP];
#pragma unroll
                for (uint32_t j = 0; j < STEP; ++j) {
                    cn[j] = rec_cc(j);
                    mn[j] = rec_cm(j);
                    ln[j] = rec_lb(j);
                    tn[j] = rec_tr(j);
                }
                if (SEL && prev_extra) {  // wave-uniform: some pod constrains 5..8 keys, all pods read eight label rows
#pragma unroll
                    for (uint32_t it = 0; it < 8; ++it) {
                        // in two halves (rows 5..8 are loaded after the first ten have been combined): bounds the live
                        // row registers of the widest instantiation below the 128-VGPR budget of a 1024-thread block
                        Rows A;
                        const uint64_t step = (uint64_t)it * step_bytes;
                        load_rows(rec_cc(it), rec_cm(it), rec_lb(it), rec_tr(it), A);
                        u32x4 f = fit_of(A);
                        if (WANT_FIT) store_rel(rb_fit + step, lane_off, f);
                        f = ((f & A.l0) & A.l1) & (A.l2 & A.l3);
                        if (TAINT) f = ((f & A.t0) & A.t1) & (A.t2 & A.t3);
                        __builtin_amdgcn_sched_barrier(0);
                        load_extra(rec_lx(it), A);
                        f = ((f & A.x0) & A.x1) & (A.x2 & A.x3);
                        store_rel(rb_feas + step, lane_off, f, KSCHED_STORE_ALT(it));
                    }
                } else {
#pragma unroll
                    for (uint32_t it = 0; it < 8; it += STEP) {
                        Rows R[STEP];
#pragma unroll
                        for (uint32_t j = 0; j < STEP; ++j) load_rows(cn[j], mn[j], ln[j], tn[j], R[j]);
                        if (!TAINT && it + STEP < 8u) {  // the next step's records, fetched ahead of this step's combine
#pragma unroll
                            for (uint32_t j = 0; j < STEP; ++j) {
                                cn[j] = rec_cc(it + STEP + j);
                                mn[j] = rec_cm(it + STEP + j);
                                ln[j] = rec_lb(it + STEP + j);
                                tn[j] = rec_tr(it + STEP + j);
                            }
                        }
#pragma unroll
                        for (uint32_t j = 0; j < STEP; ++j) combine_store(it + j, R[j], false);
                        if (TAINT && it + STEP < 8u) {  // (with taint rows the registers do not allow the early fetch)
#pragma unroll
                            for (uint32_t j = 0; j < STEP; ++j) {
                                cn[j] = rec_cc(it + STEP + j);
                                mn[j] = rec_cm(it + STEP + j);
                                ln[j] = rec_lb(it + STEP + j);
                                tn[j] = rec_tr(it + STEP + j);
                            }
                        }
                    }
                }
            } else if (!LIST && fast) {
                // short last round of the wave: same unchecked rows, one at a time
#pragma unroll 1
                for (uint32_t it = 0; it < prev_nu; ++it) {
                    Rows A;
                    load_rows(rec_cc(it), rec_cm(it), rec_lb(it), rec_tr(it), A);
                    if (SEL) load_extra(rec_lx(it), A);
                    combine_store(it, A, true);
                }
            } else {
                if (!(a.debug & 16u)) {
#pragma unroll 1
                    for (uint32_t it = 0; it < prev_nu; ++it) {
                        const uint32_t pl = it * 8u + sub;
                        emit_checked(pod0 + pl, it, (prev_over >> pl) & 1ull);
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
            KSCHED_PROF(prof_p2);
        }
        const bool had_more = more;
        if (more) {
#if KSCHED_PROFILE
            prof_t = __builtin_readcyclecounter();
            ++prof_rounds;
#endif
            // operands of round u have landed; up to kFastStores younger stores may be in flight
            if constexpr (PICK == 2) {
                KSCHED_WAIT_OPS_PICK(kFastStores);
                if (have_prev) pick_decide(prev_u * 8u, prev_nu);  // (before phase 1 overwrites the park and pick_bits)
            } else {
                KSCHED_WAIT_OPS(kFastStores);
            }
            KSCHED_PROF(prof_wait);
            prev_over = phase1(u * 8u);
            KSCHED_PROF(prof_p1);
            prev_extra = extra_any;
            prev_list = list_any;
            prev_bound = list_bound;
            if (!have_prev) stamp(3);
            prev_u = u;
            prev_nu = min(8u, u_hi - u);
            have_prev = true;
            u += u_stride;
            more = u < u_hi;
        }
        if (PICK == 2 && !had_more && have_prev) {  // the wave's last round: nothing else is in flight behind its atomic
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ar) : : "memory");
            pick_decide(prev_u * 8u, prev_nu);
        }
        first = false;
        if (!had_more) break;
    }
#undef KSCHED_WAIT_OPS
#undef KSCHED_WAIT_OPS_PICK
    if (a.trace && lane == 0) {  // every wave: latest loop end / drain of the block
        atomicMax((unsigned long long *)&a.trace[(size_t)b * 8u + 5u], (unsigned long long)wall_clock64());
        __builtin_amdgcn_s_waitcnt(0);  // all counters to zero: this wave's stores have been acknowledged
        atomicMax((unsigned long long *)&a.trace[(size_t)b * 8u + 6u], (unsigned long long)wall_clock64());
    }
#if KSCHED_PROFILE
    if (tracer) {
        a.trace[(size_t)b * 8u + 1u] = prof_p2;
        a.trace[(size_t)b * 8u + 2u] = prof_wait;
        a.trace[(size_t)b * 8u + 3u] = prof_p1;
        a.trace[(size_t)b * 8u + 4u] = prof_rounds;
    }
#endif
#undef KSCHED_PROF
    if (tracer) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicOr((unsigned long long *)&a.trace[(size_t)b * 8u + 7u], (unsigned long long)(xcc & 0xFFu));
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
    const SelectArgs *pick;        // the sampled pick that rides in the launch (PICK), or nullptr
    int pick_form;                 // 1 = by waves of the fill (select_one_pod), 2 = tile tests in phase 1
};

template <bool FIT, bool SEL, bool TAINT, bool WANT_FIT, bool LIST, int PICK = 0>
inline hipError_t launch_fused_k(const FusedLaunch &q, const FusedArgs &a) {
    auto kern = k_eval_fused<FIT, SEL, TAINT, WANT_FIT, LIST, PICK>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds);
    if (e != hipSuccess) return e;
    const IndexedSnapshot &s = *q.snap;
    const SelectArgs sa = q.pick ? *q.pick : SelectArgs{};
    if (q.ev_start || q.ev_stop)
        hipExtLaunchKernelGGL(kern, q.grid, dim3(kFusedThreads), q.lds, q.stream, q.ev_start, q.ev_stop, 0, s.d_tables, s.d_aux, q.pcpu,
                              q.pmem, q.psel, q.ptol, q.out_feas, q.out_fit, (const uint8_t *)s.d_list, a, sa);
    else
        hipLaunchKernelGGL(kern, q.grid, dim3(kFusedThreads), q.lds, q.stream, s.d_tables, s.d_aux, q.pcpu, q.pmem, q.psel, q.ptol,
                           q.out_feas, q.out_fit, (const uint8_t *)s.d_list, a, sa);
    return hipGetLastError();
}

template <bool FIT, bool SEL, bool TAINT>
inline hipError_t launch_fused_t(bool want_fit, bool list, const FusedLaunch &q, const FusedArgs &a) {
    if constexpr (SEL) {
        if (list) return want_fit ? launch_fused_k<FIT, SEL, TAINT, true, true>(q, a) : launch_fused_k<FIT, SEL, TAINT, false, true>(q, a);
    }
    if (q.pick && !want_fit) {
        if constexpr (!TAINT) {
            if (q.pick_form == 2) return launch_fused_k<FIT, SEL, TAINT, false, false, 2>(q, a);
        }
        return launch_fused_k<FIT, SEL, TAINT, false, false, 1>(q, a);
    }
    return want_fit ? launch_fused_k<FIT, SEL, TAINT, true, false>(q, a) : launch_fused_k<FIT, SEL, TAINT, false, false>(q, a);
}

// can this request's sampled pick ride in the fused launch?  (plain variants only: no second mask, no list keys)
inline bool fused_pick_applicable(const IndexedSnapshot &s, uint32_t flags, bool want_fit, uint32_t p) {
    const bool list = (flags & KSCHED_SEL) && s.lay.nkeys && s.lay.nlist > 0;
    return s.built && !want_fit && !list && p < (1u << 31);
}
// ... and as the tile-test form (PICK == 2)?  ATTEMPTS draws per pod, the reference's two predicates, at most eight label keys
// (a pod's selector then fits the eight slots of its record), and room in LDS for the per-wave park of the draws
inline bool fused_tile_pick_applicable(const IndexedSnapshot &s, uint32_t flags, uint32_t attempts, bool have_psel) {
    const IndexedLayout &l = s.lay;
    if (attempts != kPickAttempts || ((flags & KSCHED_TAINT) && l.ngroups)) return false;
    const bool sel = (flags & KSCHED_SEL) && have_psel && l.nkeys;
    if (sel && l.nkeys > 8u) return false;
    return fused_lds_bytes(l, flags & KSCHED_FIT, sel, false, nullptr, true) <= kLdsBudget;
}

inline bool fused_applicable(const IndexedSnapshot &s, uint32_t flags) {
    if (!s.built) return false;
    return fused_lds_bytes(s.lay, flags & KSCHED_FIT, (flags & KSCHED_SEL) && s.lay.nkeys, (flags & KSCHED_TAINT) && s.lay.ngroups) <= kLdsBudget;
}

// pitch = words between consecutive pod rows of the output masks (>= W; W = packed).
inline hipError_t run_fused(const IndexedSnapshot &s, uint32_t p, const int64_t *pcpu, const int64_t *pmem, const uint32_t *psel,
                            const uint64_t *ptol, uint32_t flags, uint64_t *out_feas, uint64_t *out_fit, uint32_t pitch, hipStream_t stream,
                            uint32_t debug = 0, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, uint64_t *trace = nullptr,
                            uint32_t trace_blocks = 0, const SelectArgs *pick = nullptr, int pick_form = 1, uint64_t *pick_acc = nullptr,
                            uint32_t grid_cus = 0, int round_order = 0) {
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
    a.row_cpu = l.row_cpu;
    a.row_mem = l.row_mem;
    a.row_taint = l.row_taint;
    for (int k = 0; k < 8; ++k) {
        const bool is_list = l.lab_base[k] == kLabList;  // no rows: phase 1 skips the column (list_mask8)
        a.lab_off[k] = is_list ? 0u : (l.lab_base[k] - 1u) * 128u;  // id s -> row lab_base + s - 1 (ids start at 1; keys without rows never match s != 0 below nkeys)
        a.lab_mx1[k] = is_list ? 0u : l.lab_max[k] + 1u;
    }
    a.lab_meta = s.d_lab_meta;
    a.zero64 = reinterpret_cast<const uint64_t *>(s.d_lab_meta + 64);
    a.p = p;
    a.units = (p + 7u) / 8u;
    a.debug = debug;
    a.has_tol = ptol != nullptr ? 1u : 0u;
    const bool do_fit = flags & KSCHED_FIT;
    const bool do_sel = (flags & KSCHED_SEL) && psel && l.nkeys;
    const bool do_taint = (flags & KSCHED_TAINT) && l.ngroups;
    const bool tile_pick = pick && pick_form == 2;
    const uint32_t lds = fused_lds_bytes(l, do_fit, do_sel, do_taint, &a, tile_pick);
    a.pick_acc = pick_acc;

    // chunks: as many pod ranges as keep every block resident at once (256 CUs x blocks per CU), but no
    // more than one round (64 pods) per wave needs.
    const uint32_t blocks_per_cu = std::max(1u, std::min(kLdsBudget / lds, 2048u / kFusedThreads));
    const uint32_t rounds = (a.units + 7u) / 8u;
    // chunks that give every wave one round; small batches whose pick rides along are cut finer (a block's time is its fill plus
    // ONE round either way, and the pick's pods spread over more CUs).  debug bits 18-19: A/B of that divisor (0: default).
    uint32_t per_block = kFusedWaves;
    if (pick) per_block = 4u;
    if (((debug >> 18) & 3u) == 1u) per_block = kFusedWaves;
    if (((debug >> 18) & 3u) == 2u) per_block = 4u;
    if (((debug >> 18) & 3u) == 3u) per_block = 1u;
    const uint32_t want = (rounds + per_block - 1u) / per_block;
    // grid_cus (KSCHED_OPT_GRID_CUS): the launch keeps to that many compute units, so that the launches of the FOLLOWING batches
    // (other streams) find free ones and fill while this one stores; 0 = the whole chip
    const uint32_t cus = grid_cus ? std::min(256u, grid_cus) : 256u;
    a.chunks = std::max(1u, std::min((cus * blocks_per_cu) / l.tiles, want));
    // Interleaved orders, launches of TWO rounds per wave (C3: 1 563 rounds over 51 x 16 waves): the launch lasts as long as its two-round waves, and
    // at the largest resident chunk count one wave in twelve has only one -- the smallest chunk count that still needs no third round (49: 784 waves x 2
    // rounds) fills fewer blocks for the same two rounds: step 18.05 -> 17.6 us (sweep of 44 .. 51 chunks, session r7i: 18.43 18.25 17.94 17.91 17.77
    // 17.6 17.8 18.05).  Longer launches are bound by their stores, not by the quantisation, and want every compute unit (session r7k, even / largest
    // chunk count: 150 k pods 23.5 / 23.3 us, 300 k 43.3 / 40.0, 400 k 55.7 / 52.4); one-round launches keep the finer cut (a riding pick's pods spread wider).
    if (round_order != 1 && !(debug & 0x80000000u)) {  // (debug bit 31: the largest resident chunk count, the A/B of this rule)
        const uint32_t streams = a.chunks * kFusedWaves;
        const uint32_t per_wave = (rounds + streams - 1u) / streams;
        if (per_wave == 2u) a.chunks = std::max(1u, std::min(a.chunks, (rounds + 2u * kFusedWaves - 1u) / (2u * kFusedWaves)));
    }
    a.unit_q = a.units / a.chunks;
    a.unit_rem = a.units % a.chunks;
    // KSCHED_OPT_ROUND_ORDER: 0 = interleaved, wave-major (default); 1 = blocked; 2 = interleaved, chunk-major
    a.u_stride = round_order == 1 ? 8u : a.chunks * kFusedWaves * 8u;
    a.wave_major = round_order == 2 ? 0u : 1u;
    a.tiles_rcp = (uint32_t)std::min<uint64_t>((1ull << 32) / l.tiles, 0xFFFFFFFFull);
    const uint32_t total = a.chunks * l.tiles;
    a.run = (total + 7u) / 8u;
    const dim3 grid((debug & 32u) ? total : a.run * 8u);
    a.trace = (trace && grid.x <= trace_blocks) ? trace : nullptr;
    const bool want_fit = (flags & KSCHED_WANT_FIT_MASK) && out_fit;
    const int sel = do_sel ? 1 : 0, tnt = do_taint ? 1 : 0, fit = do_fit ? 1 : 0;
    const bool list = do_sel && l.nlist > 0;
    if (pick && (want_fit || list)) return hipErrorInvalidValue;  // (the caller checks fused_pick_applicable first: a pick is never dropped silently)
    if (tile_pick && (!pick_acc || do_taint || (do_sel && l.nkeys > 8u) || pick->attempts != kPickAttempts || lds > kLdsBudget)) return hipErrorInvalidValue;
    if (pick) {
        a.pick_ppb = (p + total - 1u) / total;
        a.pick_waves = std::max(1u, std::min(8u, (a.pick_ppb + 63u) / 64u));
    }
    const FusedLaunch q{grid, lds, stream, &s, pcpu, pmem, psel, ptol, out_feas, out_fit, ev_start, ev_stop, pick, pick_form};
    a.nlist = list ? l.nlist : 0u;
    a.list_mask8 = 0;
    for (uint32_t j = 0; j < a.nlist; ++j) {
        a.list_col[j] = l.list_col[j];
        if (l.list_col[j] < 8u) a.list_mask8 |= 1u << l.list_col[j];
    }
#define KSCHED_FUSED_CASE(F, S, T) return launch_fused_t<F, S, T>(want_fit, list, q, a)
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
