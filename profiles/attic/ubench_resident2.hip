// ubench_resident2.hip -- follow-up to ubench_resident.hip: WHY does a dispatch on another queue cost ~16 us while a resident grid
// holds the chip (2.5 us on an idle chip)?  No protocol here: a grid of idle waves is parked on the chip and K empty kernels are
// launched back to back on a second stream.  Varied: blocks, waves per block, what the parked waves do (long sleeps / short sleeps
// / spinning on a global load / spinning on an LDS word), their LDS and register footprint.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_resident2 tools/ubench_resident2.hip ; run under `timeout 120`.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

// what: 0 = s_sleep 127 loop, 1 = s_sleep 8 loop, 2 = every wave polls its block's own global line (relaxed, agent scope) + s_sleep 8,
//       3 = wave 0 polls the global line, the others an LDS word, 4 = busy VALU loop (no sleep), 5 = one poll of a global line per ~2 us
template <bool BIG_REGS>
__global__ __launch_bounds__(1024) void k_park(uint64_t *lines, uint64_t *quit, uint32_t what, uint64_t life_ticks, uint64_t *ready) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (BIG_REGS) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    __shared__ uint32_t s_word;
    if (threadIdx.x == 0) {
        s_word = 0;
        smem[0] = 1;
        atomicAdd((unsigned long long *)ready, 1ull);
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t t0 = wall_clock64();
    uint32_t spin = 0;
    for (;;) {
        uint64_t q = 0;
        if ((spin++ & 255u) == 0u || what == 2u) {
            if (lane == 0) q = __hip_atomic_load(quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q = __shfl(q, 0, 64);
            if (q || wall_clock64() - t0 > life_ticks) return;
        }
        switch (what) {
            case 0: __builtin_amdgcn_s_sleep(127); break;
            case 1: __builtin_amdgcn_s_sleep(8); break;
            case 2:
                if (lane == 0) q = __hip_atomic_load(&lines[blockIdx.x * 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_sleep(8);
                break;
            case 3:
                if (wave == 0) {
                    if (lane == 0) q = __hip_atomic_load(&lines[blockIdx.x * 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (lane == 0) {
                    q = __hip_atomic_load(&s_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __builtin_amdgcn_s_sleep(8);
                break;
            case 4: asm volatile("v_add_u32 %0, %0, 1" : "+v"(spin)); break;
            default:
                if (lane == 0) q = __hip_atomic_load(&lines[blockIdx.x * 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(127);
        }
        if (q == 0xdeadbeefull) return;
    }
}
__global__ void k_tiny(uint32_t *p) {
    if (p && threadIdx.x == 5000u) *p = 1;
}
__global__ void k_quit(uint64_t *quit) { __hip_atomic_store(quit, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 1000;
    CK(hipSetDevice(0));
    CK(hipFuncSetAttribute((const void *)k_park<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_park<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipStream_t s_res, s_call;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d, greatest %d\n", lo, hi);
    CK(hipStreamCreateWithPriority(&s_res, hipStreamNonBlocking, lo));  // the parked grid on a stream of its own priority class (its own hardware queue)
    CK(hipStreamCreateWithFlags(&s_call, hipStreamNonBlocking));
    uint64_t *mem;
    CK(hipMalloc(&mem, 256 * 128 + 256));
    uint64_t *lines = mem, *quit = mem + 256 * 16, *ready = quit + 16;
    for (int i = 0; i < 300; ++i) k_tiny<<<1, 64, 0, s_call>>>(nullptr);
    CK(hipStreamSynchronize(s_call));
    {
        const double t0 = now_us();
        for (int i = 0; i < K; ++i) k_tiny<<<1, 64, 0, s_call>>>(nullptr);
        CK(hipStreamSynchronize(s_call));
        printf("idle chip: tiny kernels back to back %.2f us each\n", (now_us() - t0) / K);
    }
    struct Cfg {
        uint32_t blocks, threads, lds, what;
        bool big;
    };
    const Cfg cfgs[] = {
        {248, 1024, 131072, 0, true}, {248, 1024, 131072, 1, true}, {248, 1024, 131072, 2, true}, {248, 1024, 131072, 3, true}, {248, 1024, 131072, 4, true},
        {248, 1024, 131072, 5, true}, {248, 64, 131072, 1, true},   {248, 256, 131072, 1, true},  {248, 1024, 1024, 1, false},  {64, 1024, 131072, 1, true},
        {8, 1024, 131072, 1, true},   {248, 1024, 131072, 0, false},
    };
    static const char *whats[] = {"s_sleep 127", "s_sleep 8", "every wave polls a global line + s_sleep 8", "wave 0 polls global, others LDS, s_sleep 8", "busy VALU loop",
                                  "one global poll per ~2 us (40 x s_sleep 127)"};
    for (const Cfg &c : cfgs) {
        CK(hipMemsetAsync(mem, 0, 256 * 128 + 256, s_call));
        CK(hipStreamSynchronize(s_call));
        if (c.big) k_park<true><<<c.blocks, c.threads, c.lds, s_res>>>(lines, quit, c.what, 300ull * 100000ull, ready);
        else k_park<false><<<c.blocks, c.threads, c.lds, s_res>>>(lines, quit, c.what, 300ull * 100000ull, ready);
        CK(hipGetLastError());
        uint64_t up = 0;
        const double tr = now_us();
        while (up < c.blocks && now_us() - tr < 1e6) CK(hipMemcpy(&up, ready, 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < 50; ++i) k_tiny<<<1, 64, 0, s_call>>>(nullptr);
        CK(hipStreamSynchronize(s_call));
        const double t0 = now_us();
        for (int i = 0; i < K; ++i) k_tiny<<<1, 64, 0, s_call>>>(nullptr);
        CK(hipStreamSynchronize(s_call));
        const double t1 = now_us();
        k_quit<<<1, 1, 0, s_call>>>(quit);
        CK(hipStreamSynchronize(s_call));
        CK(hipStreamSynchronize(s_res));
        printf("parked: %3u blocks x %4u threads, %6u B LDS, %s VGPRs, waves: %-46s (%llu up) -> tiny kernels on the other stream %.2f us each\n", c.blocks, c.threads, c.lds,
               c.big ? "128" : "few", whats[c.what], (unsigned long long)up, (t1 - t0) / K);
        fflush(stdout);
    }
    return 0;
}
