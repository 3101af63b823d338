// ubench.hip -- store-pattern and VALU micro-benchmarks that size the mask kernels' design.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
// Not part of the product; numbers are quoted in profiles/HISTORY.md.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

// V0: flat, fully coalesced 16 B per lane
__global__ void st_flat16(uint64_t *out, size_t nwords, uint64_t v) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 2;
    for (; i + 1 < nwords; i += stride) {
        ulonglong2 x{v + i, v ^ i};
        *reinterpret_cast<ulonglong2 *>(out + i) = x;
    }
}
// V0b: flat 8 B per lane
__global__ void st_flat8(uint64_t *out, size_t nwords, uint64_t v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < nwords; i += stride) out[i] = v + i;
}
// V1: column tile of TW words; a wave instruction covers (64/TW) rows x TW words, 8 B per lane
template <int TW>
__global__ void st_tile8(uint64_t *out, uint32_t P, uint32_t W, uint32_t rows_per_block, uint64_t v) {
    const uint32_t c0 = blockIdx.x * TW;
    const uint32_t lane_w = threadIdx.x % TW;
    const uint32_t lane_r = threadIdx.x / TW;
    const uint32_t rows_per_iter = blockDim.x / TW;
    const uint32_t r0 = blockIdx.y * rows_per_block;
    if (c0 + lane_w >= W) return;
    for (uint32_t r = r0 + lane_r; r < min(P, r0 + rows_per_block); r += rows_per_iter)
        out[(size_t)r * W + c0 + lane_w] = v + r;
}
// V3: row per lane, 4 consecutive words (direct-kernel pattern): block = 4 waves on adjacent 4-word columns
__global__ void st_rowlane(uint64_t *out, uint32_t P, uint32_t W, uint32_t tiles_per_block, uint64_t v) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w0 = (blockIdx.x * 4 + wave) * 4;
    if (w0 >= W) return;
    for (uint32_t t = 0; t < tiles_per_block; ++t) {
        const uint32_t p = (blockIdx.y * tiles_per_block + t) * 64 + lane;
        if (p >= P) break;
        for (int c = 0; c < 4; ++c)
            if (w0 + c < W) out[(size_t)p * W + w0 + c] = v + p + c;
    }
}

// VALU issue rates: N dependent-free compares producing ballots, folded on the scalar side
template <int MODE>
__global__ void valu_rate(const int64_t *nodes, const int64_t *pods, uint64_t *out, int iters) {
    const int lane = threadIdx.x & 63;
    int64_t a0 = nodes[threadIdx.x], a1 = nodes[threadIdx.x + 256];
    uint32_t b0 = (uint32_t)a0, b1 = (uint32_t)a1;
    uint64_t acc = 0;
    uint32_t keep_lo = 0, keep_hi = 0;
    for (int i = 0; i < iters; ++i) {
        const int64_t s = pods[i];  // uniform -> SGPR
        if (MODE == 0) {            // 2 x v_cmp_le_i64
            acc += __ballot(s <= a0) & __ballot((s ^ 5) <= a1);
        } else if (MODE == 1) {     // 2 x v_cmp_le_u32
            acc += __ballot((uint32_t)s <= b0) & __ballot((uint32_t)(s ^ 5) <= b1);
        } else if (MODE == 2) {     // v_pk_max_u16 + v_cmp_eq_u32
            acc += __ballot(((b0 | 0x80008000u) - (uint32_t)s & 0x80008000u) == 0x80008000u);
        }
        (void)lane;
    }
    (void)keep_lo; (void)keep_hi;
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

extern "C" __device__ uint32_t ub_writelane(uint32_t val, uint32_t lane, uint32_t old) __asm("llvm.amdgcn.writelane.i32");
__global__ void writelane_rate(const int64_t *nodes, const int64_t *pods, uint64_t *out, int iters) {
    uint32_t a = (uint32_t)nodes[threadIdx.x];
    uint32_t lo = 0, hi = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t s = (uint32_t)pods[i];
        const uint64_t m = __ballot(s <= a);
        lo = ub_writelane((uint32_t)m, i & 63, lo);
        hi = ub_writelane((uint32_t)(m >> 32), i & 63, hi);
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = ((uint64_t)hi << 32) | lo;
}

// LDS table reads: TW-lane groups read a contiguous row segment of a random row (ds_read_b64), R reads per output word
template <int TW, int R>
__global__ void lds_rows(const uint16_t *rowidx, uint64_t *out, uint32_t P, uint32_t W, uint32_t rows, uint32_t rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) uint64_t tab[];  // [rows][TW]
    for (uint32_t i = threadIdx.x; i < rows * TW; i += blockDim.x) tab[i] = 0x9E3779B97F4A7C15ull * (i + 1);
    __syncthreads();
    const uint32_t c0 = blockIdx.x * TW;
    const uint32_t lw = threadIdx.x % TW, lr = threadIdx.x / TW, rpi = blockDim.x / TW;
    const uint32_t r0 = blockIdx.y * rows_per_block;
    if (c0 + lw >= W) return;
    for (uint32_t r = r0 + lr; r < min(P, r0 + rows_per_block); r += rpi) {
        uint64_t acc = ~0ull;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const uint32_t row = rowidx[(size_t)r * R + k];
            acc &= tab[row * TW + lw] | (uint64_t)k;
        }
        out[(size_t)r * W + c0 + lw] = acc;
    }
}

template <class F>
float time_ms(F f, int reps = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs %d, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    const uint32_t shapes[][2] = {{100000, 79}, {125000, 157}, {125000, 782}, {10000, 16}};
    for (auto &sh : shapes) {
        const uint32_t P = sh[0], W = sh[1];
        const size_t nwords = (size_t)P * W;
        uint64_t *out;
        CK(hipMalloc(&out, nwords * 8));
        const double gb = nwords * 8 / 1e9;
        printf("--- mask %u x %u words = %.1f MB\n", P, W, gb * 1e3);
        float ms = time_ms([&] { hipLaunchKernelGGL(st_flat16, dim3(2048), dim3(256), 0, 0, out, nwords, 1ull); });
        printf("  flat 16B/lane           %8.1f us  %7.1f GB/s\n", ms * 1e3, gb / ms * 1e3);
        ms = time_ms([&] { hipLaunchKernelGGL(st_flat8, dim3(2048), dim3(256), 0, 0, out, nwords, 1ull); });
        printf("  flat  8B/lane           %8.1f us  %7.1f GB/s\n", ms * 1e3, gb / ms * 1e3);
        {
            const uint32_t gx = (W + 15) / 16, gy = (2048 + gx - 1) / gx, rpb = (P + gy - 1) / gy;
            ms = time_ms([&] { hipLaunchKernelGGL((st_tile8<16>), dim3(gx, (P + rpb - 1) / rpb), dim3(256), 0, 0, out, P, W, rpb, 1ull); });
            printf("  tile 16 words (128B/row) %7.1f us  %7.1f GB/s   grid %ux%u\n", ms * 1e3, gb / ms * 1e3, gx, (P + rpb - 1) / rpb);
        }
        {
            const uint32_t gx = (W + 31) / 32, gy = (2048 + gx - 1) / gx, rpb = (P + gy - 1) / gy;
            ms = time_ms([&] { hipLaunchKernelGGL((st_tile8<32>), dim3(gx, (P + rpb - 1) / rpb), dim3(256), 0, 0, out, P, W, rpb, 1ull); });
            printf("  tile 32 words (256B/row) %7.1f us  %7.1f GB/s\n", ms * 1e3, gb / ms * 1e3);
        }
        {
            const uint32_t gx = (W + 63) / 64, gy = (2048 + gx - 1) / gx, rpb = (P + gy - 1) / gy;
            ms = time_ms([&] { hipLaunchKernelGGL((st_tile8<64>), dim3(gx, (P + rpb - 1) / rpb), dim3(256), 0, 0, out, P, W, rpb, 1ull); });
            printf("  tile 64 words (512B/row) %7.1f us  %7.1f GB/s\n", ms * 1e3, gb / ms * 1e3);
        }
        {
            const uint32_t gx = (W + 15) / 16, tiles = (P + 63) / 64;
            uint32_t gy = (2048 + gx - 1) / gx;
            const uint32_t tpb = (tiles + gy - 1) / gy;
            gy = (tiles + tpb - 1) / tpb;
            ms = time_ms([&] { hipLaunchKernelGGL(st_rowlane, dim3(gx, gy), dim3(256), 0, 0, out, P, W, tpb, 1ull); });
            printf("  row-per-lane 4 words     %7.1f us  %7.1f GB/s\n", ms * 1e3, gb / ms * 1e3);
        }
        // LDS-table kernel prototype: R table reads per output word
        {
            const uint32_t rows = 512;
            std::vector<uint16_t> h((size_t)P * 12);
            uint32_t x = 12345;
            for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)((x >> 8) % rows); }
            uint16_t *ri;
            CK(hipMalloc(&ri, h.size() * 2));
            CK(hipMemcpy(ri, h.data(), h.size() * 2, hipMemcpyHostToDevice));
            const uint32_t gx = (W + 15) / 16, gy = std::max(1u, (512 + gx - 1) / gx), rpb = (P + gy - 1) / gy;
            const dim3 grid(gx, (P + rpb - 1) / rpb);
            ms = time_ms([&] { hipLaunchKernelGGL((lds_rows<16, 6>), grid, dim3(1024), rows * 16 * 8, 0, ri, out, P, W, rows, rpb); });
            printf("  LDS tile16 R=6  (1024thr) %7.1f us  %7.1f GB/s   grid %ux%u\n", ms * 1e3, gb / ms * 1e3, grid.x, grid.y);
            ms = time_ms([&] { hipLaunchKernelGGL((lds_rows<16, 12>), grid, dim3(1024), rows * 16 * 8, 0, ri, out, P, W, rows, rpb); });
            printf("  LDS tile16 R=12 (1024thr) %7.1f us  %7.1f GB/s\n", ms * 1e3, gb / ms * 1e3);
            ms = time_ms([&] { hipLaunchKernelGGL((lds_rows<16, 12>), grid, dim3(512), rows * 16 * 8, 0, ri, out, P, W, rows, rpb); });
            printf("  LDS tile16 R=12 (512thr)  %7.1f us  %7.1f GB/s\n", ms * 1e3, gb / ms * 1e3);
            CK(hipFree(ri));
        }
        CK(hipFree(out));
    }
    // VALU rates
    {
        const int iters = 4096, blocks = 256 * 8;
        int64_t *nodes, *pods;
        uint64_t *out;
        CK(hipMalloc(&nodes, 512 * 8));
        CK(hipMalloc(&pods, iters * 8));
        CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
        CK(hipMemset(nodes, 1, 512 * 8));
        CK(hipMemset(pods, 2, iters * 8));
        auto rep = [&](const char *name, float ms, double ops_per_iter) {
            const double waves = (double)blocks * 4;
            printf("  %-28s %8.1f us  %.2f cycles/wave-iter at 2.4GHz/SIMD share (%.3g wave-ops/s)\n", name, ms * 1e3,
                   ms * 1e-3 * 2.4e9 / (iters * waves / 1024.0), ops_per_iter * iters * waves / (ms * 1e-3));
        };
        printf("--- VALU issue (2048 blocks x 4 waves, %d iters)\n", iters);
        rep("2x v_cmp_le_i64", time_ms([&] { hipLaunchKernelGGL((valu_rate<0>), dim3(blocks), dim3(256), 0, 0, nodes, pods, out, iters); }), 2);
        rep("2x v_cmp_le_u32", time_ms([&] { hipLaunchKernelGGL((valu_rate<1>), dim3(blocks), dim3(256), 0, 0, nodes, pods, out, iters); }), 2);
        rep("packed16 sub+and+cmp", time_ms([&] { hipLaunchKernelGGL((valu_rate<2>), dim3(blocks), dim3(256), 0, 0, nodes, pods, out, iters); }), 3);
        rep("cmp + 2x writelane", time_ms([&] { hipLaunchKernelGGL(writelane_rate, dim3(blocks), dim3(256), 0, 0, nodes, pods, out, iters); }), 3);
    }
    return 0;
}
