// ubench_anyorder.hip -- does hipExtAnyOrderLaunch let consecutive kernels of ONE stream overlap on gfx950?  (hip_ext.h carries an old note
// that the flag "is not supported on AMD GFX9xx boards".)  Two kernels of 64 blocks that each spin ~50 us are launched back to back on one
// stream: in order they take ~100 us, overlapped ~50 us.  Then the launch-to-launch gap of a chip-filling kernel (256 blocks x 1024 threads,
// 150 KB of LDS each, ~20 us) with and without the flag.  Not part of the product; quoted in profiles/r05_*.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_anyorder tools/ubench_anyorder.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

__global__ __launch_bounds__(1024) void spin(uint64_t ticks, uint32_t *out) {  // wall_clock64: 100 MHz
    extern __shared__ uint8_t smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && out) out[blockIdx.x] = (uint32_t)smem[0];
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t *out;
    CK(hipMalloc(&out, 4096 * 4));
    CK(hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int flags = 0; flags < 2; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(s));
            const double t0 = now_us();
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, flags, (uint64_t)5000, out);
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, flags, (uint64_t)5000, out);
            CK(hipStreamSynchronize(s));
            printf("two 50 us kernels of 64 small blocks, one stream, flags=%d: %.1f us\n", flags, now_us() - t0);
        }
    }
    // chip-filling kernels (one block per CU): K launches, per-launch time
    for (int flags = 0; flags < 2; ++flags)
        for (int K : {20, 200}) {
            double best = 1e30;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipStreamSynchronize(s));
                const double t0 = now_us();
                for (int i = 0; i < K; ++i) hipExtLaunchKernelGGL(spin, dim3(256), dim3(1024), 150 * 1024, s, nullptr, nullptr, flags, (uint64_t)1800, out);
                CK(hipStreamSynchronize(s));
                best = std::min(best, (now_us() - t0) / K);
            }
            printf("K=%3d chip-filling 18 us kernels (256 blocks x 1024 threads, 150 KB LDS), flags=%d: %.2f us per launch\n", K, flags, best);
        }
    // mixed: every 8th launch in order
    for (int K : {20, 200}) {
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipStreamSynchronize(s));
            const double t0 = now_us();
            for (int i = 0; i < K; ++i) hipExtLaunchKernelGGL(spin, dim3(256), dim3(1024), 150 * 1024, s, nullptr, nullptr, (i % 8) ? 1 : 0, (uint64_t)1800, out);
            CK(hipStreamSynchronize(s));
            best = std::min(best, (now_us() - t0) / K);
        }
        printf("K=%3d the same, every 8th launch in order: %.2f us per launch\n", K, best);
    }
    return 0;
}
