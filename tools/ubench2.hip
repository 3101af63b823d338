// ubench2.hip -- store-pattern experiments for the fused mask kernel (tile = 16 words = 128 B per pod row).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench2 tools/ubench2.hip ; run on the GPU box.
// Not part of the product; the numbers it prints are quoted in profiles/HISTORY.md.
#include <hip/hip_runtime.h>

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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct Args {
    uint32_t P, W, tiles, chunks, pods_per_chunk, map;  // map: 0 plain (tile fastest), 1 xcd-grouped, 2 chunk fastest
};

// LPR lanes per row (8 -> 16 B per lane, 16 -> 8 B per lane); NT non-temporal; a wave owns 64-pod groups
template <int LPR, int NT, int THREADS>
__global__ __launch_bounds__(THREADS) void st_fused(uint64_t *__restrict__ out, const Args a, uint64_t v) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t b = blockIdx.x;
    uint32_t tile, chunk;
    if (a.map == 1) {
        const uint32_t xcd = b & 7u, bi = b >> 3;
        tile = bi % a.tiles;
        chunk = (bi / a.tiles) * 8u + xcd;
    } else if (a.map == 2) {
        chunk = b % a.chunks;
        tile = b / a.chunks;
    } else {
        tile = b % a.tiles;
        chunk = b / a.tiles;
    }
    if (chunk >= a.chunks || tile >= a.tiles) return;
    if (threadIdx.x == 0) smem[0] = 1;  // keep the LDS allocation
    constexpr uint32_t WAVES = THREADS / 64;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t pod_lo = chunk * a.pods_per_chunk, pod_hi = min(a.P, pod_lo + a.pods_per_chunk);
    constexpr uint32_t RPI = 64 / LPR;  // rows per wave instruction
    const uint32_t wl = lane % LPR, sub = lane / LPR;
    const uint32_t wpl = 16 / LPR;  // words per lane
    const uint32_t w0 = tile * 16u + wl * wpl;
    if (w0 >= a.W) return;
    const bool has_all = w0 + wpl <= a.W;
    for (uint32_t g0 = pod_lo + wave * 64u; g0 < pod_hi; g0 += WAVES * 64u) {
#pragma unroll
        for (uint32_t it = 0; it < 64 / RPI; ++it) {
            const uint32_t pod = g0 + it * RPI + sub;
            if (pod >= pod_hi) continue;
            const size_t o = (size_t)pod * a.W + w0;
            if (LPR == 8) {
                u32x4 f{(uint32_t)v, pod, it, lane};
                if (has_all) {
                    if (NT) __builtin_nontemporal_store(f, reinterpret_cast<u32x4_a8 *>(out + o));
                    else *reinterpret_cast<u32x4_a8 *>(out + o) = f;
                } else {
                    out[o] = v;
                }
            } else {
                if (NT) __builtin_nontemporal_store(v + pod, out + o);
                else out[o] = v + pod;
            }
        }
    }
}

template <class F>
float time_us(F f, int reps = 30) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    float tot = 0;
    for (int i = 0; i < reps; ++i) {
        f();  // f launches with hipExtLaunchKernelGGL(a, b) so the events bracket the dispatch exactly
    }
    hipDeviceSynchronize();
    (void)tot;
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}

template <int LPR, int NT, int THREADS>
void run(const char *name, uint64_t *out, uint32_t P, uint32_t W, uint32_t lds, uint32_t blocks_per_cu, int map) {
    Args a{};
    a.P = P;
    a.W = W;
    a.tiles = (W + 15) / 16;
    a.map = map;
    const uint32_t waves = THREADS / 64;
    const uint32_t slots = 256 * blocks_per_cu;
    const uint32_t groups = (P + 63) / 64;
    const uint32_t max_chunks = std::max(1u, slots / a.tiles);
    uint32_t chunks = std::min(max_chunks, (groups + waves - 1) / waves);
    const uint32_t k = (groups + chunks * waves - 1) / (chunks * waves);
    chunks = (groups + k * waves - 1) / (k * waves);
    a.chunks = chunks;
    a.pods_per_chunk = k * waves * 64;
    const uint32_t nblk = (map == 1 ? ((chunks + 7) & ~7u) : chunks) * a.tiles;
    auto kern = st_fused<LPR, NT, THREADS>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const float us = time_us([&] { hipLaunchKernelGGL(kern, dim3(nblk), dim3(THREADS), lds, 0, out, a, 0x1234ull); });
    const double bytes = (double)P * W * 8;
    printf("  %-44s %7.1f us  %7.1f GB/s   (blocks %u, k %u)\n", name, us, bytes / us * 1e-3, nblk, k);
}

int main() {
    const uint32_t shapes[][2] = {{100000, 79}, {100000, 80}, {125000, 157}, {125000, 782}};
    for (auto &sh : shapes) {
        const uint32_t P = sh[0], W = sh[1];
        uint64_t *out;
        CK(hipMalloc(&out, (size_t)P * W * 8 + 4096));
        printf("--- mask %u x %u words = %.1f MB (back-to-back launches, includes ~1.5 us boundary)\n", P, W, (double)P * W * 8e-6);
        run<8, 0, 1024>("16B/lane 1024thr 1blk/CU plain", out, P, W, 86 * 1024, 1, 0);
        run<8, 0, 1024>("16B/lane 1024thr 1blk/CU xcd-grouped", out, P, W, 86 * 1024, 1, 1);
        run<8, 0, 1024>("16B/lane 1024thr 1blk/CU chunk-fastest", out, P, W, 86 * 1024, 1, 2);
        run<8, 1, 1024>("16B/lane 1024thr 1blk/CU plain NT", out, P, W, 86 * 1024, 1, 0);
        run<16, 0, 1024>("8B/lane 1024thr 1blk/CU plain", out, P, W, 86 * 1024, 1, 0);
        run<16, 1, 1024>("8B/lane 1024thr 1blk/CU plain NT", out, P, W, 86 * 1024, 1, 0);
        run<8, 0, 1024>("16B/lane 1024thr 2blk/CU plain", out, P, W, 70 * 1024, 2, 0);
        run<8, 0, 1024>("16B/lane 1024thr 2blk/CU xcd-grouped", out, P, W, 70 * 1024, 2, 1);
        run<8, 1, 1024>("16B/lane 1024thr 2blk/CU plain NT", out, P, W, 70 * 1024, 2, 0);
        run<8, 0, 512>("16B/lane 512thr 2blk/CU plain", out, P, W, 70 * 1024, 2, 0);
        run<8, 0, 256>("16B/lane 256thr 8blk/CU plain", out, P, W, 16 * 1024, 8, 0);
        run<8, 0, 256>("16B/lane 256thr 8blk/CU xcd-grouped", out, P, W, 16 * 1024, 8, 1);
        run<16, 0, 256>("8B/lane 256thr 8blk/CU plain", out, P, W, 16 * 1024, 8, 0);
        CK(hipFree(out));
    }
    return 0;
}
