// ubench_resident.hip -- what does it cost to hand batches to a grid that STAYS on the chip?
//
// the mask kernel pays its fill (tile index -> LDS, ~7 us of a 17-19 us launch at C3) on every launch although
// the snapshot does not change between steps.  A resident grid would stage once and then take batch descriptors in stream order.
// Before building that kernel this measures the PROTOCOL alone with a dummy resident grid that has the mask kernel's footprint
// (1024 threads, 128 VGPRs, 131 KB of LDS per block, one block per CU):
//   * can a one-wave "post" kernel be placed at all while the grid holds 245 / 255 / 256 CUs?
//   * per-batch time of  post -> grid sees it -> every block signals -> the stream continues,  with
//       A  our own post kernel + our own wait kernel on the caller's stream,
//       B  hipStreamWriteValue64 + hipStreamWaitValue64 (if the device offers them),
//       C  posts only, `depth` batches ahead of the waits (the pipe form: submit = post, wait = explicit)
//     at work = 0 (pure protocol) and work = 12 us per batch (each wave busy-waits: what a C3 batch takes at the store rate),
//   * against the baseline: back-to-back launches of an empty grid of the same footprint (what a step costs today on top of its work).
// Every kernel here ends by itself (idle / lifetime limits in device ticks): a broken protocol ends in a printed error, not a hung GPU.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_resident tools/ubench_resident.hip ; run under `timeout 120`.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

constexpr uint32_t kSlots = 8;     // batches in flight at most
constexpr uint32_t kLine = 16;     // uint64 words per 128-byte line
constexpr uint32_t kMaxBlocks = 256;
struct Ctl {                       // every field on its own 128-byte line
    uint64_t tail[kLine];          // [0] = last posted sequence number (1, 2, ...)
    uint64_t quit[kLine];          // [0] != 0: leave after the batches posted so far
    uint64_t err[kLine];           // [0]: a wait ran into its limit
    uint64_t count[kSlots][kLine]; // blocks that have finished the slot's batch
    uint64_t flag[kSlots][kLine];  // sequence number of the slot's last finished batch (own-kernel waits)
    uint64_t bell[kMaxBlocks][kLine];  // variant bit 3: one doorbell line per block (written by the post kernel's threads) ...
    uint64_t fin[kMaxBlocks][kLine];   // ... and one completion line per block (the block's last finished batch), read by the wait kernel's threads
    uint64_t ready[kLine];             // blocks that have started
    uint64_t stamps_on[kLine];
    uint64_t prog[16][kLine];          // block 0: [wave] = the batch that wave last finished ([0][1]: the tail wave 0 last saw, [0][2..9]: arrived[] as wave 0 last saw them)
    uint64_t post_stamp[1200];         // when the post kernel of [seq] ran
    uint64_t stamps[2][1200][2];       // blocks 0 and 100: [seq] -> {doorbell seen, last wave arrived} (100 MHz ticks)
};

__device__ inline uint64_t ld_acq(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_rel(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// The resident grid.  Each wave polls `tail`; on a new batch it "works" for work_ticks (100 MHz ticks) -- optionally storing
// store_bytes_per_wave bytes with the mask kernel's store policy -- then the block's last wave to finish adds one to the slot's
// counter; the block that completes the count publishes the sequence number (flag[slot], and *signal[slot] for hipStreamWaitValue64).
__device__ inline uint64_t ld_rlx(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint64_t ld_sys(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// variant: bit 0 = relaxed polls + one acquire fence per batch (else an acquire load per poll = a cache invalidate per poll);
//          bit 1 = long sleeps between polls (s_sleep 64 instead of 8); bit 2 = the doorbell lives in host memory (system-scope polls, host_tail / host_flag)
__global__ __launch_bounds__(1024) void k_resident(Ctl *c, uint32_t nblocks, uint32_t work_ticks, uint64_t idle_limit, uint64_t life_limit,
                                                   uint64_t *const *signal, uint8_t *out, uint32_t store_bytes_per_wave, uint32_t variant,
                                                   uint64_t *host_tail, uint64_t *host_flag) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    asm volatile("v_mov_b32 v127, 0" ::: "v127");  // the mask kernel's register footprint: 128 VGPRs -> 16 waves fill a CU's register file
    __shared__ uint32_t arrived[kSlots];
    if (threadIdx.x < kSlots) arrived[threadIdx.x] = 0;
    smem[threadIdx.x * 16u] = (uint8_t)threadIdx.x;  // (the dynamic LDS is really allocated)
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t t_start = wall_clock64();
    uint64_t t_idle = t_start;
    uint32_t polls = 0;
    (void)idle_limit;
    __shared__ uint64_t s_tail;
    __shared__ uint32_t s_quit;
    if (threadIdx.x == 0) {
        s_tail = 0;
        s_quit = 0;
        __hip_atomic_fetch_add(&c->ready[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (uint64_t seq = 1;; ++seq) {
        // wait for batch `seq`
        if (variant & 8u) {
            // one poller per block (its own doorbell line: nobody else reads it), the other waves watch an LDS word
            for (;;) {
                uint64_t tail = 0;
                uint32_t q = 0;
                if (wave == 0) {
                    if (lane == 0) {
                        tail = (variant & 64u) ? ld_sys(&c->bell[blockIdx.x][0]) : ld_rlx(&c->bell[blockIdx.x][0]);
                        if ((++polls & 1023u) == 0u) {  // the shared words and the chip-wide clock only once in a while
                            q = (uint32_t)ld_rlx(&c->quit[0]);
                            if (wall_clock64() - t_start > life_limit) q = 1;
                        }
                    }
                    tail = __shfl(tail, 0, 64);
                    q = __shfl(q, 0, 64);
                    if (tail >= seq && lane == 0 && c->stamps_on[0] && (blockIdx.x == 0u || blockIdx.x == 100u) && seq < 1200u)
                        c->stamps[blockIdx.x ? 1 : 0][seq][0] = wall_clock64();  // the doorbell was seen
                    if (lane == 0 && blockIdx.x == 0u) __hip_atomic_store(&c->prog[0][1], tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (tail >= seq || q) {
                        // ONE cache invalidate per block and batch (a buffer_inv per WAVE -- 496 per XCD and batch -- was what the first
                        // version of this loop spent its 32 us on), then the block's other waves are let go through LDS
                        if (tail >= seq && !(variant & 16u)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        if (lane == 0) {
                            __hip_atomic_store(&s_tail, tail, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (q && tail < seq) __hip_atomic_store(&s_quit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
                if (lane == 0) {
                    tail = __hip_atomic_load(&s_tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    q = __hip_atomic_load(&s_quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                tail = __shfl(tail, 0, 64);
                q = __shfl(q, 0, 64);
                if (tail >= seq) break;
                if (q) return;
                __builtin_amdgcn_s_sleep(4);
            }
            // (NO fence here: a buffer_inv sc1 per WAVE costs ~1.7 us and the waves of a CU take turns at it -- 16 waves, 28 us per
            // batch: what every earlier version of this loop measured.  Wave 0 above invalidates once for the block.)
        } else
        for (;;) {
            uint64_t tail = 0;
            if (lane == 0) tail = (variant & 4u) ? ld_sys(host_tail) : (variant & 1u) ? ld_rlx(&c->tail[0]) : ld_acq(&c->tail[0]);
            tail = __shfl(tail, 0, 64);
            if (tail >= seq) {
                if (variant & 5u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                break;
            }
            uint64_t q = 0;
            if (lane == 0) q = (variant & 1u) ? ld_rlx(&c->quit[0]) : ld_acq(&c->quit[0]);
            q = __shfl(q, 0, 64);
            if (q) return;
            if ((++polls & 255u) == 0u && wall_clock64() - t_start > life_limit) return;
            (void)t_idle;
            if (variant & 2u) __builtin_amdgcn_s_sleep(64);
            else __builtin_amdgcn_s_sleep(8);
        }
        const uint64_t t0 = 0;
        if (store_bytes_per_wave) {
            uint8_t *dst = out + ((size_t)(seq % 4u) * gridDim.x * 16u + (size_t)blockIdx.x * 16u + wave) * store_bytes_per_wave;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = {(uint32_t)seq, lane, wave, blockIdx.x};
            for (uint32_t off = lane * 16u; off < store_bytes_per_wave; off += 1024u)
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + off), "v"(v) : "memory");
        }
        // (no clock reads here: s_memrealtime is ONE counter for the whole chip -- 4 000 waves reading it in a loop serialise on it,
        // which is what the first versions of this benchmark measured: ~32 us per batch whatever the "work")
        for (uint32_t i = 0; i < work_ticks / 40u; ++i) __builtin_amdgcn_s_sleep(15);  // 15 x 64 cycles = 0.4 us at 2.4 GHz per trip
        (void)t0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // completion: last wave of the block -> one device-scope add; last block -> publish
        const uint32_t slot = (uint32_t)(seq % kSlots);
        if (lane == 0) {
            if (blockIdx.x == 0u) {
                __hip_atomic_store(&c->prog[wave][0], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (wave == 0)
                    for (uint32_t k = 0; k < kSlots; ++k) __hip_atomic_store(&c->prog[0][2 + k], (uint64_t)arrived[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const uint32_t a = atomicAdd(&arrived[slot], 1u);
            if (a == (blockDim.x >> 6) - 1u) {
                arrived[slot] = 0;
                if ((variant & 8u) && c->stamps_on[0] && (blockIdx.x == 0u || blockIdx.x == 100u) && seq < 1200u)
                    c->stamps[blockIdx.x ? 1 : 0][seq][1] = wall_clock64();  // the block's last wave has arrived
                if (variant & 8u) {
                    if (variant & 32u) __hip_atomic_store(&c->fin[blockIdx.x][0], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (every wave waited for its own write-through stores: vmcnt(0) above)
                    else st_rel(&c->fin[blockIdx.x][0], seq);
                }
                const uint64_t n = (variant & 8u) ? 0ull : __hip_atomic_fetch_add(&c->count[slot][0], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (!(variant & 8u) && n == nblocks - 1u) {
                    __hip_atomic_store(&c->count[slot][0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    st_rel(&c->flag[slot][0], seq);
                    if (signal) __hip_atomic_store(signal[slot], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (variant & 4u) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
}

__global__ void k_post(Ctl *c, uint64_t seq) { st_rel(&c->tail[0], seq); }
// per-block form: thread b rings block b's doorbell -- after batch seq - ring has left the grid (flow control on the device: the host
// enqueues posts without knowing how far the grid is)
__global__ void k_post_blocks(Ctl *c, uint64_t seq, uint32_t nblocks, uint64_t ring, uint64_t limit) {
    const uint32_t b = threadIdx.x;
    if (b == 0 && c->stamps_on[0] && seq < 1200u) c->post_stamp[seq] = wall_clock64();
    if (b >= nblocks) return;
    if (seq > ring) {
        uint64_t polls = 0;
        while (ld_rlx(&c->fin[b][0]) + ring < seq) {
            if ((polls & 1023u) == 1023u && ld_rlx(&c->err[0])) return;  // (somebody has already run into a limit: end at once)
            if (++polls > limit) {
                st_rel(&c->err[0], seq | (1ull << 62) | ((uint64_t)b << 40));  // bit 62: a POST's flow control ran into its limit (block b)
                return;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    st_rel(&c->bell[b][0], seq);
}
__global__ void k_wait_blocks(Ctl *c, uint64_t seq, uint32_t nblocks, uint64_t limit) {
    const uint32_t b = threadIdx.x;
    if (b >= nblocks) return;
    uint64_t polls = 0;
    while (ld_rlx(&c->fin[b][0]) < seq) {
        if ((polls & 1023u) == 1023u && ld_rlx(&c->err[0])) return;
        if (++polls > limit) {
            if (!ld_rlx(&c->err[0])) st_rel(&c->err[0], seq | ((uint64_t)b << 40) | (ld_rlx(&c->fin[b][0]) << 20));  // block b's fin in bits 20..39
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__global__ void k_quit(Ctl *c) { st_rel(&c->quit[0], 1ull); }
__global__ void k_wait(Ctl *c, uint64_t seq, uint64_t limit) {
    const uint64_t t0 = wall_clock64();
    while (ld_acq(&c->flag[seq % kSlots][0]) < seq) {
        if (ld_acq(&c->err[0]) || wall_clock64() - t0 > limit) {  // (one wait that ran into its limit ends all later ones at once)
            st_rel(&c->err[0], seq);
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
__global__ __launch_bounds__(1024) void k_empty_footprint(uint32_t *p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    smem[threadIdx.x * 16u] = 1;
    if (p && threadIdx.x == 5000u) *p = smem[0];
}
__global__ void k_tiny(uint32_t *p) {
    if (p && threadIdx.x == 5000u) *p = 1;
}

static uint32_t g_threads = 1024;
static uint64_t g_ring = 6;
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2000;
    const uint32_t lds = 131072;
    int dev = 0, can_wait = 0;
    CK(hipSetDevice(dev));
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, dev);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    printf("device %s, CUs %d, hipDeviceAttributeCanUseStreamWaitValue = %d\n", prop.gcnArchName, prop.multiProcessorCount, can_wait);
    CK(hipFuncSetAttribute((const void *)k_resident, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k_empty_footprint, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    hipStream_t s_res, s_call, s_post;
    CK(hipStreamCreateWithFlags(&s_post, hipStreamNonBlocking));
    int prio_lo = 0, prio_hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    CK(hipStreamCreateWithPriority(&s_res, hipStreamNonBlocking, prio_lo));  // its own priority class = its own hardware queue: nothing else may queue behind a kernel that never ends
    CK(hipStreamCreateWithFlags(&s_call, hipStreamNonBlocking));
    Ctl *ctl;
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    uint64_t *sig_host[kSlots] = {};
    uint64_t **sig_dev = nullptr;
    if (can_wait) {
        for (uint32_t i = 0; i < kSlots; ++i) {
            hipError_t e = hipExtMallocWithFlags((void **)&sig_host[i], 8, hipMallocSignalMemory);
            if (e != hipSuccess) {
                printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s -> no stream-value mode\n", hipGetErrorString(e));
                can_wait = 0;
                break;
            }
        }
        if (can_wait) {
            CK(hipMalloc(&sig_dev, sizeof(sig_host)));
            CK(hipMemcpy(sig_dev, sig_host, sizeof(sig_host), hipMemcpyHostToDevice));
        }
    }
    uint8_t *out;
    const size_t out_bytes = (size_t)4 * 256 * 16 * 16384;  // 4 rotating images of 256 blocks x 16 waves x 16 KiB
    CK(hipMalloc(&out, out_bytes));

    // ---- baseline: back-to-back launches -----------------------------------------------------------------
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 200; ++i) k_tiny<<<1, 64, 0, s_call>>>(nullptr);
        CK(hipStreamSynchronize(s_call));
        double t0 = now_us();
        for (int i = 0; i < K; ++i) k_tiny<<<1, 64, 0, s_call>>>(nullptr);
        CK(hipStreamSynchronize(s_call));
        double t1 = now_us();
        for (int i = 0; i < K; ++i) k_empty_footprint<<<255, 1024, lds, s_call>>>(nullptr);
        CK(hipStreamSynchronize(s_call));
        double t2 = now_us();
        if (rep) printf("baseline: tiny kernel back to back %.2f us each; empty 255 x 1024-thread x 128 KiB-LDS grid %.2f us each\n", (t1 - t0) / K, (t2 - t1) / K);
    }

    const uint64_t tick_ms = 100000;  // wall_clock64: 100 MHz
    uint64_t *host_tail = nullptr, *host_flag = nullptr;  // mode D: the doorbell and the completion word in pinned host memory
    CK(hipHostMalloc((void **)&host_tail, 128, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void **)&host_flag, 128, hipHostMallocMapped | hipHostMallocCoherent));
    static const char *names[] = {"A own post + own wait per batch", "B hipStreamWriteValue64 + hipStreamWaitValue64 per batch", "C posts 6 ahead of the waits (own kernels)",
                                  "C + every wave stores 16 KiB sc1, no timed work", "D host writes the doorbell, host polls the completion word (pinned host memory), per batch",
                                  "E like D, posts 6 ahead", "F per-block doorbells: post + wait per batch on the caller's stream",
                                  "G per-block doorbells: posts on an internal stream (ring of 6, flow control on the device), one wait per batch on the caller's stream",
                                  "H like G, one wait per 4 batches", "I per-block doorbells, ONE stream: posts 6 ahead of the waits",
                                  "J per-block doorbells: every post first (own stream, no flow control), one wait at the end"};
    // returns false when batches do not get through
    auto run = [&](uint32_t nblocks, uint32_t work, int mode, uint32_t variant) -> bool {
        if (mode == 1 && !can_wait) return true;
        if (mode == 4 || mode == 5) variant |= 4u | 1u;
        if (mode >= 6) variant |= 8u | 1u;
        CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), s_call));
        if (can_wait)
            for (uint32_t i = 0; i < kSlots; ++i) CK(hipMemsetAsync(sig_host[i], 0, 8, s_call));
        CK(hipStreamSynchronize(s_call));
        {
            const uint64_t one = 1;
            CK(hipMemcpy(&ctl->stamps_on[0], &one, 8, hipMemcpyHostToDevice));
        }
        *(volatile uint64_t *)host_tail = 0;
        *(volatile uint64_t *)host_flag = 0;
        const uint32_t store_per_wave = mode == 3 ? 16384u : 0u;
        k_resident<<<nblocks, g_threads, lds, s_res>>>(ctl, nblocks, mode == 3 ? 0u : work, 200 * tick_ms, 4000 * tick_ms, mode == 1 ? (uint64_t *const *)sig_dev : nullptr, out,
                                                  store_per_wave, variant, host_tail, host_flag);
        CK(hipGetLastError());
        {  // the grid is up before anything else of ours is enqueued (a spinning wait kernel on a CU a resident block still needs = deadlock)
            uint64_t ready = 0;
            const double tr = now_us();
            while (ready < nblocks && now_us() - tr < 2e6) CK(hipMemcpy(&ready, &ctl->ready[0], 8, hipMemcpyDeviceToHost));
            if (ready < nblocks) printf("  (only %llu of %u blocks came up)\n", (unsigned long long)ready, nblocks);
        }
        const int warm = 50;
        double t0 = 0;
        uint64_t seq = 0;
        bool host_timeout = false;
        auto post = [&]() {
            ++seq;
            if (mode == 6 || mode == 9) k_post_blocks<<<1, 256, 0, s_call>>>(ctl, seq, nblocks, 1000000, 200000);
            else if (mode >= 7) k_post_blocks<<<1, 256, 0, s_post>>>(ctl, seq, nblocks, g_ring, 200000);
            else if (mode >= 4) __atomic_store_n(host_tail, seq, __ATOMIC_RELEASE);
            else if (mode == 1) CK(hipStreamWriteValue64(s_call, &ctl->tail[0], seq, 0));
            else k_post<<<1, 1, 0, s_call>>>(ctl, seq);
        };
        auto wait = [&](uint64_t sq) {
            if (mode >= 6) {
                if (mode == 10) return;  // (J: no wait before the end)
                if (mode != 8 || sq % 4 == 0) k_wait_blocks<<<1, 256, 0, s_call>>>(ctl, sq, nblocks, 200000);
            } else if (mode >= 4) {  // (in-order completion is what the flag of this mode says: the last block of batch sq publishes sq; batches finish in order within a block)
                const double tw = now_us();
                while (__atomic_load_n(host_flag, __ATOMIC_ACQUIRE) < sq)
                    if (now_us() - tw > 1e6) {
                        host_timeout = true;
                        break;
                    }
            } else if (mode == 1) CK(hipStreamWaitValue64(s_call, sig_host[sq % kSlots], sq, hipStreamWaitValueGte, ~0ull));
            else k_wait<<<1, 1, 0, s_call>>>(ctl, sq, 1000 * tick_ms);
        };
        const uint32_t depth = (mode == 2 || mode == 3 || mode == 5 || mode == 9) ? kSlots - 2 : 1;
        // (modes G / H: the posts run ahead on their own stream; the device-side flow control bounds them)
        for (int i = 0; i < K + warm && !host_timeout; ++i) {
            if (i == warm) {
                CK(hipStreamSynchronize(s_call));
                t0 = now_us();
            }
            post();
            if (seq >= depth) wait(seq - depth + 1);
        }
        for (uint64_t sq = seq - depth + 2; sq <= seq && !host_timeout; ++sq) wait(sq);
        if (mode >= 7) k_wait_blocks<<<1, 256, 0, s_call>>>(ctl, seq, nblocks, 200000);
        CK(hipStreamSynchronize(s_call));
        CK(hipStreamSynchronize(s_post));
        const double t1 = now_us();
        k_quit<<<1, 1, 0, s_call>>>(ctl);
        CK(hipStreamSynchronize(s_call));
        CK(hipStreamSynchronize(s_res));
        uint64_t err = 0;
        CK(hipMemcpy(&err, &ctl->err[0], 8, hipMemcpyDeviceToHost));
        const bool bad = err || host_timeout;
        if (err) {
            uint64_t bell0 = 0, fin0 = 0;
            CK(hipMemcpy(&bell0, &ctl->bell[0][0], 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&fin0, &ctl->fin[0][0], 8, hipMemcpyDeviceToHost));
            static uint64_t prog[16][kLine];
            CK(hipMemcpy(prog, ctl->prog, sizeof prog, hipMemcpyDeviceToHost));
            printf("  block 0: waves finished batch");
            for (int w = 0; w < 16; ++w) printf(" %llu", (unsigned long long)prog[w][0]);
            printf("; wave 0 last saw tail %llu, arrived[] =", (unsigned long long)prog[0][1]);
            for (int k = 0; k < 8; ++k) printf(" %llu", (unsigned long long)prog[0][2 + k]);
            printf("\n");
            printf("  err word: seq %llu, block %llu, that block's fin then %llu, %s; now bell[0] = %llu, fin[0] = %llu, posted %llu\n", (unsigned long long)(err & 0xFFFFF),
                   (unsigned long long)((err >> 40) & 0xFFF), (unsigned long long)((err >> 20) & 0xFFFFF), (err >> 62) & 1 ? "a POST's flow control" : "a WAIT", (unsigned long long)bell0,
                   (unsigned long long)fin0, (unsigned long long)seq);
        }
        printf("grid %3u  work %5.1f us  polls: %-7s sleep %-2s  %-90s %7.2f us per batch%s\n", nblocks, work / 100.0, (variant & 4u) ? "system" : (variant & 1u) ? "relaxed" : "acquire",
               (variant & 2u) ? "64" : "8", names[mode], (t1 - t0) / K, bad ? "   ** a wait ran into its limit **" : "");
        if (mode == 6) {
            static uint64_t st[2][1200][2], ps[1200];
            CK(hipMemcpy(st, ctl->stamps, sizeof st, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ps, ctl->post_stamp, sizeof ps, hipMemcpyDeviceToHost));
            double post_to_seen[2] = {0, 0}, seen_to_arrived[2] = {0, 0}, arrived_to_next_post = 0;
            int cnt = 0;
            for (int q = 200; q < 1000; ++q) {
                for (int k = 0; k < 2; ++k) {
                    post_to_seen[k] += (double)(int64_t)(st[k][q][0] - ps[q]);
                    seen_to_arrived[k] += (double)(int64_t)(st[k][q][1] - st[k][q][0]);
                }
                arrived_to_next_post += (double)(int64_t)(ps[q + 1] - std::max(st[0][q][1], st[1][q][1]));
                ++cnt;
            }
            printf("    inside the grid (mean over %d batches, us): post kernel ran -> doorbell seen  block 0: %.2f  block 100: %.2f;  seen -> last wave arrived  %.2f / %.2f;  arrived -> NEXT post kernel ran %.2f\n",
                   cnt, post_to_seen[0] / cnt / 100, post_to_seen[1] / cnt / 100, seen_to_arrived[0] / cnt / 100, seen_to_arrived[1] / cnt / 100, arrived_to_next_post / cnt / 100);
        }
        fflush(stdout);
        return !bad;
    };
    g_ring = 1000000;
    printf("---- which part of the pipelined form stops the grid?\n");
    run(248, 1200, 9, 32);   // I: posts ahead, but one stream
    run(248, 1200, 10, 32);  // J: posts on their own stream, nothing else running
    for (uint32_t threads : {64u, 256u}) {  // G with fewer waves per resident block
        g_threads = threads;
        printf("resident blocks of %u threads: ", threads);
        run(248, 1200, 7, 32);
    }
    g_threads = 1024;
    for (uint32_t nblocks : {32u, 64u, 128u, 192u, 240u})  // G: how many resident blocks does it take?
        run(nblocks, 1200, 7, 32);
    return 0;
}
