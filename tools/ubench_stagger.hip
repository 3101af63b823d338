// ubench_stagger.hip -- is the mask kernel's store rate a matter of WHICH ROWS the chip writes at the same time?
//
// Observation (round 5, tools/placement_probe.py): the same mask kernel on the same box writes the C5 shard's 748 MiB mask in 141 us into one
// allocation and in 172 us into the next, flat in time, bimodal by allocation.  In the kernel's block -> (chunk, tile) mapping the tile-blocks of a
// chunk walk the SAME pod rows at the same time (wave w of each of them its w-th sixteenth of the chunk): the whole chip writes into chunks x 16
// row-wide windows (C5 shard: 80 windows of 6 272 bytes).  If the memory channels are interleaved coarsely, that few windows cover them unevenly, and
// how unevenly depends on where the allocation's physical pieces lie.
//
// This program writes the mask's pattern (128-byte segments, the shipped store policy) in that order (STAG 0) and with every tile-block starting at a
// different place of its wave ranges and wrapping around (STAG 1: tiles x chunks x 16 windows of 128 bytes), into `allocs` successive allocations.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_stagger tools/ubench_stagger.hip      usage: tools/ubench_stagger [allocs=8]
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t err_ = (x);                                                              \
        if (err_ != hipSuccess) {                                                         \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(err_), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st16(uint64_t *p, u32x4 f) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(f) : "memory"); }

struct Args {
    uint32_t P, pitch, tiles, chunks, run, units, unit_q, unit_rem;
};

template <int STAG>
__global__ __launch_bounds__(1024) void st_tiles(uint64_t *__restrict__ out, const Args a) {
    extern __shared__ uint8_t smem[];
    const uint32_t b = blockIdx.x;
    const uint32_t l = (b & 7u) * a.run + (b >> 3);
    if ((b >> 3) >= a.run) return;
    const uint32_t chunk = l / a.tiles, tile = l % a.tiles;
    if (chunk >= a.chunks) return;
    if (threadIdx.x == 0) smem[0] = 1;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t wl = lane & 7u, sub = lane >> 3;  // 8 lanes per pod row, 8 pod rows per wave instruction
    const uint32_t w0 = tile * 16u + wl * 2u;
    const uint32_t c_lo = chunk * a.unit_q + min(chunk, a.unit_rem);
    const uint32_t c_n = a.unit_q + (chunk < a.unit_rem ? 1u : 0u);
    const uint32_t u_lo = c_lo + (wave * c_n) / 16u, u_hi = c_lo + ((wave + 1u) * c_n) / 16u;
    const uint32_t n = u_hi - u_lo;  // units of 8 pods = wave instructions
    uint32_t k = STAG ? (uint32_t)(((uint64_t)tile * n) / a.tiles) : 0u;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t pod = (u_lo + k) * 8u + sub;
        if (pod < a.P && w0 + 1u < a.pitch) st16(out + (size_t)pod * a.pitch + w0, u32x4{pod, lane, w0, 7u});
        k = (k + 1u == n) ? 0u : k + 1u;
    }
}

static float time_us(void (*kern)(uint64_t *, Args), const Args &a, std::vector<uint64_t *> &bufs, int reps = 20) {
    hipEvent_t s, e;
    CK(hipEventCreate(&s));
    CK(hipEventCreate(&e));
    const uint32_t lds = 100 * 1024;
    int k = 0;
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(kern, dim3(a.run * 8u), dim3(1024), lds, 0, bufs[(k++) % bufs.size()], a);
    CK(hipDeviceSynchronize());
    float tot = 0;
    for (int i = 0; i < reps; ++i) {
        hipExtLaunchKernelGGL(kern, dim3(a.run * 8u), dim3(1024), lds, 0, s, e, 0, bufs[(k++) % bufs.size()], a);
        CK(hipEventSynchronize(e));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, s, e));
        tot += ms;
    }
    CK(hipEventDestroy(s));
    CK(hipEventDestroy(e));
    return tot * 1000.f / reps;
}

int main(int argc, char **argv) {
    const int allocs = argc > 1 ? atoi(argv[1]) : 8;
    const uint32_t shapes[][2] = {{125000, 782}, {125000, 157}, {100000, 79}};
    const uint32_t lds = 100 * 1024;
    CK(hipFuncSetAttribute((const void *)st_tiles<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)st_tiles<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<void *> spacers;
    for (auto &sh : shapes) {
        const uint32_t P = sh[0], W = sh[1];
        Args a{};
        a.P = P;
        a.pitch = (W + 15u) & ~15u;
        a.tiles = (W + 15u) / 16u;
        a.units = (P + 7u) / 8u;
        const uint32_t rounds = (a.units + 7u) / 8u, want = (rounds + 15u) / 16u;
        a.chunks = std::max(1u, std::min(256u / a.tiles, want));
        a.unit_q = a.units / a.chunks;
        a.unit_rem = a.units % a.chunks;
        a.run = (a.chunks * a.tiles + 7u) / 8u;
        const size_t bytes = (size_t)P * a.pitch * 8;
        const int nrot = (int)std::max<size_t>(1, std::min<size_t>(12, (((size_t)320 << 20) + bytes - 1) / bytes));
        printf("--- mask %u x %u words, pitch %u B, %.1f MB algorithmic, %u tiles x %u chunks, %d buffer(s) in rotation\n", P, W, a.pitch * 8, (double)P * W * 8e-6, a.tiles,
               a.chunks, nrot);
        for (int t = 0; t < allocs; ++t) {
            std::vector<uint64_t *> bufs(nrot);
            for (auto &p : bufs) CK(hipMalloc(&p, bytes));
            const float u0 = time_us(st_tiles<0>, a, bufs), u1 = time_us(st_tiles<1>, a, bufs), u0b = time_us(st_tiles<0>, a, bufs);
            printf("  alloc %2d at %p: rows in step %7.2f us (%6.1f GB/s)  staggered %7.2f us (%6.1f GB/s)  rows in step again %7.2f us\n", t, (void *)bufs[0], u0,
                   (double)P * W * 8 / u0 * 1e-3, u1, (double)P * W * 8 / u1 * 1e-3, u0b);
            for (auto &p : bufs) CK(hipFree(p));
            void *sp;
            CK(hipMalloc(&sp, (size_t)(3 + 61 * t) << 20));  // kept: the next allocation lands somewhere else
            spacers.push_back(sp);
        }
    }
    return 0;
}
