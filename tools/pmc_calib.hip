// pmc_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md "HBM": FETCH_SIZE under-reports wide coalesced reads by 2x, WRITE_SIZE is uncalibrated:
// "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip
// Run under  rocprofv3 --kernel-trace --pmc WRITE_SIZE  (and, separately, --pmc FETCH_SIZE); tools/pmc_traffic.py
// divides the known bytes below by the counter value to get the per-pattern correction factor.
// Not part of the product.
#include <hip/hip_runtime.h>

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

// flat streaming write, 16 B per lane: exactly n16 * 16 bytes
__global__ __launch_bounds__(256) void calib_write_flat16(u32x4 *__restrict__ out, size_t n16, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        out[i] = u32x4{v, (uint32_t)i, v, v};
}

// the mask kernel's store shape: 8 lanes write one 128-byte segment of a pod row; a wave instruction covers 8
// consecutive rows (row pitch `pitch16` 16-byte units), a block column owns one 128-byte column (tile) of all rows.
// Exactly rows * tiles * 128 bytes.
__global__ __launch_bounds__(1024) void calib_write_tile128(u32x4 *__restrict__ out, uint32_t rows, uint32_t tiles, uint32_t pitch16,
                                                           uint32_t v) {
    const uint32_t tile = blockIdx.x % tiles, chunk = blockIdx.x / tiles, chunks = gridDim.x / tiles;
    const uint32_t wp = threadIdx.x & 7u;
    for (uint32_t r = chunk * 128u + (threadIdx.x >> 3); r < rows; r += chunks * 128u)
        out[(size_t)r * pitch16 + tile * 8u + wp] = u32x4{v, r, tile, wp};
}

// flat streaming read, 16 B per lane: exactly n16 * 16 bytes (the store never happens)
__global__ __launch_bounds__(256) void calib_read_flat16(const u32x4 *__restrict__ in, size_t n16, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 x = in[i];
        acc ^= x.x ^ x.y ^ x.z ^ x.w;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t)512 << 20;  // past L2 (32 MiB) and the Infinity Cache (256 MiB)
    u32x4 *buf;
    uint32_t *sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    const size_t n16 = bytes / 16;
    // mask-shaped case: 125000 rows x 157 words, pitch 160 words (C4 shard), 10 tiles -> 160 MB exactly
    const uint32_t rows = 125000, tiles = 10, pitch16 = 80;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(calib_write_flat16, dim3(256 * 8), dim3(256), 0, 0, buf, n16, 7u + rep);
        hipLaunchKernelGGL(calib_read_flat16, dim3(256 * 8), dim3(256), 0, 0, buf, n16, sink);
        hipLaunchKernelGGL(calib_write_tile128, dim3(tiles * 51), dim3(1024), 0, 0, buf, rows, tiles, pitch16, 3u + rep);
    }
    CK(hipDeviceSynchronize());
    printf("calib_write_flat16 bytes %zu\ncalib_read_flat16 bytes %zu\ncalib_write_tile128 bytes %zu\n", bytes, bytes,
           (size_t)rows * tiles * 128);
    return 0;
}
