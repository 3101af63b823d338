// ubench_kernarg.hip -- how long does a wave wait for its kernel arguments at the start of a launch?  The mask kernel's prologue
// makes four dependent groups of scalar loads from the kernarg segment before its first vector load.  Here wave 0 (and the last
// wave) of every block time, with s_memrealtime (100 MHz): the first s_load from the kernarg segment, a load from the next
// 64-byte line, one from a line 192 bytes further, the first line again (scalar-cache hit), and one global scalar load of a
// device buffer (for scale).  Launched back to back like the bench's steps.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_kernarg tools/ubench_kernarg.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

struct Big {
    uint32_t w[72];  // 288 bytes, like FusedArgs + the pointers
};

constexpr int kStamps = 6;

__global__ __launch_bounds__(1024) void k_args(uint64_t *out, const uint32_t *dev, Big a) {
#ifdef __HIP_DEVICE_COMPILE__
    const uint64_t kp = (uint64_t)(const __attribute__((address_space(4))) void *)__builtin_amdgcn_kernarg_segment_ptr();
#else
    const uint64_t kp = 0;
#endif
    uint64_t t0, t1, t2, t3, t4, t5;
    uint32_t v0, v1, v2, v3, v4;
    uint64_t p_out, p_dev;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    asm volatile("s_load_dword %0, %1, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=s"(v0) : "s"(kp));
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "s"(v0));
    asm volatile("s_load_dword %0, %1, 0x50\n\ts_waitcnt lgkmcnt(0)" : "=s"(v1) : "s"(kp));
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) : "s"(v1));
    asm volatile("s_load_dword %0, %1, 0x110\n\ts_waitcnt lgkmcnt(0)" : "=s"(v2) : "s"(kp));
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3) : "s"(v2));
    asm volatile("s_load_dword %0, %1, 0x14\n\ts_waitcnt lgkmcnt(0)" : "=s"(v3) : "s"(kp));
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t4) : "s"(v3));
    asm volatile("s_load_dwordx2 %0, %1, 0x8\n\ts_waitcnt lgkmcnt(0)" : "=s"(p_dev) : "s"(kp));
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v4) : "s"(p_dev));
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t5) : "s"(v4));
    if ((threadIdx.x & 63u) == 0) {
        const uint32_t wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
        uint64_t *o = out + ((size_t)blockIdx.x * waves + wave) * kStamps;
        o[0] = t0, o[1] = t1, o[2] = t2, o[3] = t3, o[4] = t4, o[5] = t5 + (v0 + v1 + v2 + v3 + v4 == 0xFFFFFFFFu ? 1 : 0) + (a.w[71] == 12345u ? 1 : 0);
    }
}

int main() {
    const int blocks = 256;
    uint64_t *d_out;
    uint32_t *d_dev;
    CK(hipMalloc(&d_out, (size_t)blocks * 16 * kStamps * 8));
    CK(hipMalloc(&d_dev, 4096));
    CK(hipMemset(d_dev, 0, 4096));
    std::vector<uint64_t> h((size_t)blocks * 16 * kStamps);
    Big a{};
    for (int threads : {64, 1024}) {
        const int waves = threads / 64;
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < (rep ? 50 : 1); ++i) hipLaunchKernelGGL(k_args, dim3(blocks), dim3(threads), 0, 0, d_out, d_dev, a);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), d_out, (size_t)blocks * waves * kStamps * 8, hipMemcpyDeviceToHost));
            uint64_t tmin = ~0ull;
            for (int b = 0; b < blocks; ++b) tmin = std::min(tmin, h[((size_t)b * waves) * kStamps]);
            const char *names[5] = {"first kernarg line", "next line", "line +192 B", "first line again", "kernarg ptr + device dword"};
            for (int wsel = 0; wsel < 2; ++wsel) {
                const int w = wsel ? waves - 1 : 0;
                if (wsel && waves == 1) break;
                printf("%4d threads/block, %s, wave %2d: entry (since the first block's) med ", threads, rep == 0 ? "first launch" : (rep == 1 ? "50th launch " : "100th launch"), w);
                std::vector<double> e;
                for (int b = 0; b < blocks; ++b) e.push_back((h[((size_t)b * waves + w) * kStamps] - tmin) * 0.01);
                std::sort(e.begin(), e.end());
                printf("%.2f max %.2f us |", e[blocks / 2], e[blocks - 1]);
                for (int s = 0; s < 5; ++s) {
                    std::vector<double> d;
                    for (int b = 0; b < blocks; ++b) {
                        const uint64_t *o = &h[((size_t)b * waves + w) * kStamps];
                        d.push_back((o[s + 1] - o[s]) * 0.01);
                    }
                    std::sort(d.begin(), d.end());
                    printf(" %s %.2f (p90 %.2f)", names[s], d[blocks / 2], d[blocks * 9 / 10]);
                }
                printf(" us\n");
            }
        }
    }
    return 0;
}
