// ubench_icache.hip -- what does a kernel pay for executing code for the FIRST time in a launch?  A block runs the same
// straight-line stretch of KB kilobytes (s_nop: 4 bytes, one issue cycle each) several times; pass 0 runs it with the
// instruction cache as the launch left it, the later passes with the code resident.  wave 0 and the last wave of every block report
// the wall clock (100 MHz) per pass.  Question behind it: how much of the mask kernel's 4-5 us from block entry to its first
// stores is instruction fetch (the fused kernel's hot path is ~10 KB of code, executed once or twice per wave at C3).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_icache tools/ubench_icache.hip
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

constexpr int kPasses = 4;

#define STRETCH(N) asm volatile(".rept " #N "\n\ts_nop 0\n\t.endr" ::: "memory")

template <int KB>
__global__ __launch_bounds__(1024) void k_code(uint64_t *out) {
    uint64_t t[kPasses + 1];
    t[0] = wall_clock64();
#pragma unroll 1
    for (int p = 0; p < kPasses; ++p) {
        if (KB == 2) STRETCH(512);
        if (KB == 4) STRETCH(1024);
        if (KB == 8) STRETCH(2048);
        if (KB == 16) STRETCH(4096);
        if (KB == 32) STRETCH(8192);
        t[p + 1] = wall_clock64();
        asm volatile("" ::"s"(t[p + 1]));
    }
    if ((threadIdx.x & 63u) == 0) {
        const uint32_t wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
        uint64_t *o = out + ((size_t)blockIdx.x * waves + wave) * kPasses;
        for (int p = 0; p < kPasses; ++p) o[p] = t[p + 1] - t[p];
    }
}

template <int KB>
void run(int threads, uint64_t *d_out, std::vector<uint64_t> &h) {
    const int blocks = 256, waves = threads / 64;
    for (int rep = 0; rep < 3; ++rep) {  // rep 0: first launch of this kernel in the process; later: back-to-back launches
        for (int i = 0; i < (rep ? 20 : 1); ++i) hipLaunchKernelGGL(k_code<KB>, dim3(blocks), dim3(threads), 0, 0, d_out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_out, (size_t)blocks * waves * kPasses * 8, hipMemcpyDeviceToHost));
        printf("%2d KB code, %4d threads/block, %s: ", KB, threads, rep == 0 ? "first launch " : (rep == 1 ? "20th launch  " : "40th launch  "));
        for (int p = 0; p < kPasses; ++p) {
            std::vector<uint64_t> w0, wl;
            for (int b = 0; b < blocks; ++b) {
                w0.push_back(h[((size_t)b * waves + 0) * kPasses + p]);
                wl.push_back(h[((size_t)b * waves + waves - 1) * kPasses + p]);
            }
            std::sort(w0.begin(), w0.end());
            std::sort(wl.begin(), wl.end());
            printf(" p%d w0 %.2f (max %.2f) wL %.2f |", p, w0[blocks / 2] * 0.01, w0[blocks - 1] * 0.01, wl[blocks / 2] * 0.01);
        }
        printf("  us\n");
    }
}

int main() {
    uint64_t *d_out;
    CK(hipMalloc(&d_out, 256 * 16 * kPasses * 8));
    std::vector<uint64_t> h(256 * 16 * kPasses);
    for (int threads : {64, 1024}) {
        run<2>(threads, d_out, h);
        run<4>(threads, d_out, h);
        run<8>(threads, d_out, h);
        run<16>(threads, d_out, h);
        run<32>(threads, d_out, h);
    }
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    printf("hipDeviceAttributeClockRate %d kHz (a warm pass of K KB is K * 256 one-cycle instructions)\n", khz);
    return 0;
}
