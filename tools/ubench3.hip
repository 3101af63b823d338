// ubench3.hip -- how fast can the mask's store pattern go?  Pure store kernels in the fused kernel's block -> (chunk, tile)
// mapping (XCD-contiguous runs), rows pitched to 128 B, for tile widths of 16 / 32 / 64 words per pod row and the store
// policies plain / sc1 (write-through) / nt, timed with events attached to the dispatch.  Also a flat stream (each wave
// instruction writes 1 KiB contiguous) as the ceiling.  Not part of the product; numbers are quoted in profiles/HISTORY.md.
// Round 5: an 8-word tile (64-byte segments: what a 512-node tile of the mask kernel would write) next to the 16-word one, and
// every pattern once more with the output ROTATED over enough buffers to exceed the 256 MiB Infinity Cache (ROT, the bench's form).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench3 tools/ubench3.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));

template <int POL>
__device__ __forceinline__ void st16(uint64_t *p, u32x4 f) {
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(f) : "memory");
    else if (POL == 2) __builtin_nontemporal_store(f, reinterpret_cast<u32x4_a8 *>(p));
    else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(f) : "memory");
    else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(f) : "memory");  // the shipped policy (round 5)
    else if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(f) : "memory");
    else *reinterpret_cast<u32x4_a8 *>(p) = f;
}

struct Args {
    uint32_t P, pitch, tiles, chunks, run, units, unit_q, unit_rem;
};

// TW words per (pod, tile) segment; LPR = TW / 2 lanes per pod row; a wave instruction writes 64 / LPR pod rows
// ORD: 0 = every wave owns a contiguous pod range (rounds 1 - 5), 1 = the launch's rounds of 64 pods dealt over its chunks * 16 waves, wave-major (round 6's default)
template <int TW, int POL, int ORD = 0>
__global__ __launch_bounds__(1024) void st_tiles(uint64_t *__restrict__ out, const Args a) {
    extern __shared__ uint8_t smem[];
    const uint32_t b = blockIdx.x;
    const uint32_t l = (b & 7u) * a.run + (b >> 3);
    if ((b >> 3) >= a.run) return;
    const uint32_t chunk = l / a.tiles, tile = l % a.tiles;
    if (chunk >= a.chunks) return;
    if (threadIdx.x == 0) smem[0] = 1;
    constexpr uint32_t LPR = TW / 2, RPI = 64 / LPR;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t wl = lane % LPR, sub = lane / LPR;
    const uint32_t w0 = tile * TW + wl * 2u;
    if (w0 + 1u >= a.pitch + 1u) return;
    // units of 8 pods, cut over chunks then over the 16 waves (as run_fused does)
    const uint32_t c_lo = chunk * a.unit_q + min(chunk, a.unit_rem);
    const uint32_t c_n = a.unit_q + (chunk < a.unit_rem ? 1u : 0u);
    const uint32_t u_lo = c_lo + (wave * c_n) / 16u, u_hi = c_lo + ((wave + 1u) * c_n) / 16u;
    if (ORD == 0) {
        for (uint32_t pod0 = u_lo * 8u; pod0 < u_hi * 8u; pod0 += RPI) {
            const uint32_t pod = pod0 + sub;
            if (pod < a.P && w0 < a.pitch) st16<POL>(out + (size_t)pod * a.pitch + w0, u32x4{pod, lane, w0, 7u});
        }
    } else {
        for (uint32_t r = wave * a.chunks + chunk; r * 64u < a.P; r += a.chunks * 16u)
            for (uint32_t pod0 = r * 64u; pod0 < r * 64u + 64u; pod0 += RPI) {
                const uint32_t pod = pod0 + sub;
                if (pod < a.P && w0 < a.pitch) st16<POL>(out + (size_t)pod * a.pitch + w0, u32x4{pod, lane, w0, 7u});
            }
    }
}

template <int POL>
__global__ __launch_bounds__(1024) void st_flat(uint64_t *__restrict__ out, size_t words) {
    const size_t stride = (size_t)gridDim.x * 1024u * 2u;
    for (size_t w = ((size_t)blockIdx.x * 1024u + threadIdx.x) * 2u; w + 1 < words; w += stride) st16<POL>(out + w, u32x4{(uint32_t)w, 1u, 2u, 3u});
}

static uint64_t *g_rot[16];
static int g_nrot = 1, g_k = 0;
static uint64_t *next_out() { return g_rot[(g_k++) % g_nrot]; }

template <class L>
float time_us(L launch, int reps = 24) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch(nullptr, nullptr);
    CK(hipDeviceSynchronize());
    float tot = 0;
    for (int i = 0; i < reps; ++i) {
        launch(a, b);
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        tot += ms;
    }
    return tot * 1000.f / reps;
}

template <int TW, int POL, int ORD = 0>
void run_tiles(const char *name, uint64_t *out, uint32_t P, uint32_t W) {
    Args a{};
    a.P = P;
    a.pitch = (W + 15u) & ~15u;
    a.tiles = (W * 64u + TW * 64u - 1u) / (TW * 64u);
    a.units = (P + 7u) / 8u;
    const uint32_t rounds = (a.units + 7u) / 8u, want = (rounds + 15u) / 16u;
    a.chunks = std::max(1u, std::min(256u / a.tiles, want));
    a.unit_q = a.units / a.chunks;
    a.unit_rem = a.units % a.chunks;
    const uint32_t total = a.chunks * a.tiles;
    a.run = (total + 7u) / 8u;
    auto kern = st_tiles<TW, POL, ORD>;
    const uint32_t lds = 100 * 1024;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const float us = time_us([&](hipEvent_t s, hipEvent_t e) {
        uint64_t *o = next_out();
        if (s) hipExtLaunchKernelGGL(kern, dim3(a.run * 8u), dim3(1024), lds, 0, s, e, 0, o, a);
        else hipLaunchKernelGGL(kern, dim3(a.run * 8u), dim3(1024), lds, 0, o, a);
    });
    const double bytes = (double)P * W * 8;
    printf("  %-40s %7.2f us  %7.1f GB/s  (blocks %u = %u chunks x %u tiles)\n", name, us, bytes / us * 1e-3, total, a.chunks, a.tiles);
}

template <int POL>
void run_flat(const char *name, uint64_t *out, uint32_t P, uint32_t W) {
    const size_t words = (size_t)P * ((W + 15u) & ~15u);
    auto kern = st_flat<POL>;
    const float us = time_us([&](hipEvent_t s, hipEvent_t e) {
        uint64_t *o = next_out();
        if (s) hipExtLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, s, e, 0, o, words);
        else hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, o, words);
    });
    printf("  %-40s %7.2f us  %7.1f GB/s  (pitched bytes %.1f MB)\n", name, us, (double)P * W * 8 / us * 1e-3, words * 8e-6);
}

int main() {
    const uint32_t shapes[][2] = {{100000, 79}, {125000, 157}, {125000, 782}};
    for (int rot = 0; rot < 2; ++rot)
    for (auto &sh : shapes) {
        const uint32_t P = sh[0], W = sh[1];
        const size_t bytes = (size_t)P * ((W + 15u) & ~15u) * 8 + 4096;
        g_nrot = rot ? (int)std::min<size_t>(16, ((size_t)320 << 20) / bytes + 1) : 1;
        if (rot && g_nrot == 1) continue;  // one buffer already exceeds the cache
        for (int i = 0; i < g_nrot; ++i) CK(hipMalloc(&g_rot[i], bytes));
        uint64_t *out = g_rot[0];
        printf("--- mask %u x %u words = %.1f MB algorithmic (rows pitched to 128 B; kernel time from dispatch events; output %s)\n", P, W, (double)P * W * 8e-6,
               rot ? "ROTATED over > 256 MiB of buffers" : "in place");
        run_tiles<8, 0>("tile 8 words (64 B) plain", out, P, W);
        run_tiles<8, 1>("tile 8 words (64 B) sc1", out, P, W);
        run_tiles<8, 2>("tile 8 words (64 B) nt", out, P, W);
        run_flat<0>("flat plain", out, P, W);
        run_flat<1>("flat sc1", out, P, W);
        run_flat<2>("flat nt", out, P, W);
        run_tiles<16, 0>("tile 16 words plain", out, P, W);
        run_tiles<16, 1>("tile 16 words sc1", out, P, W);
        run_tiles<16, 2>("tile 16 words nt", out, P, W);
        run_tiles<16, 3>("tile 16 words sc0 sc1", out, P, W);
        // round 6: the shipped policy `sc0 sc1 nt` and `sc1 nt` for both segment sizes, and the interleaved round order
        run_tiles<8, 4>("tile 8 words (64 B) sc0 sc1 nt", out, P, W);
        run_tiles<8, 5>("tile 8 words (64 B) sc1 nt", out, P, W);
        run_tiles<16, 4>("tile 16 words sc0 sc1 nt", out, P, W);
        run_tiles<16, 5>("tile 16 words sc1 nt", out, P, W);
        run_tiles<8, 4, 1>("tile 8 w (64 B) sc0 sc1 nt, interleaved", out, P, W);
        run_tiles<16, 4, 1>("tile 16 w sc0 sc1 nt, interleaved", out, P, W);
        run_tiles<16, 1, 1>("tile 16 w sc1, interleaved", out, P, W);
        run_tiles<16, 0, 1>("tile 16 w plain, interleaved", out, P, W);
        run_tiles<32, 4, 1>("tile 32 w sc0 sc1 nt, interleaved", out, P, W);
        run_tiles<32, 0>("tile 32 words plain", out, P, W);
        run_tiles<32, 1>("tile 32 words sc1", out, P, W);
        run_tiles<64, 0>("tile 64 words plain", out, P, W);
        run_tiles<64, 1>("tile 64 words sc1", out, P, W);
        for (int i = 0; i < g_nrot; ++i) CK(hipFree(g_rot[i]));
    }
    return 0;
}
