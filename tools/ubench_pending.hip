// Does the L1 (TCP) of gfx950 stall on a request that hits a line whose fill is still pending?  One lane = one dependent chain of R rounds over
// a 4 MiB table (L2-resident); per round the lane reads 64 bytes of ONE random line as
//   A: four 16-byte loads in flight together          B: one 16-byte load, then -- once it is back -- the other three
//   C: four 16-byte loads from FOUR random lines      D: one 16-byte load            E: two 8-byte loads of one 16-byte unit, together
// Grid = 125 000 lanes in blocks of 256 (the shape of k_pick_bestfit_lanes).      build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_pending tools/ubench_pending.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef long long i64x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kLines = 65536, kRounds = 14;

template <int V>
__global__ __launch_bounds__(256) void k(const i64x2 *__restrict__ t, uint32_t n, uint64_t *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    uint64_t h = tid * 0x9E3779B97F4A7C15ull + 12345u;
    uint64_t acc = 0;
    for (uint32_t r = 0; r < kRounds; ++r) {
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t line = (uint32_t)(h >> 40) & (kLines - 1u);
        const i64x2 *v = t + (size_t)line * 4u;
        if (V == 0) {
            const i64x2 a = v[0], b = v[1], c = v[2], d = v[3];
            acc += a.x + b.y + c.x + d.y;
        } else if (V == 1) {
            const i64x2 a = v[0];
            asm volatile("" : "+v"(v) : "v"(a));
            const i64x2 b = v[1], c = v[2], d = v[3];
            acc += a.x + b.y + c.x + d.y;
        } else if (V == 2) {
            const i64x2 a = v[0], b = t[(size_t)((line * 7u + 1u) & (kLines - 1u)) * 4u + 1u], c = t[(size_t)((line * 13u + 5u) & (kLines - 1u)) * 4u + 2u],
                        d = t[(size_t)((line * 29u + 3u) & (kLines - 1u)) * 4u + 3u];
            acc += a.x + b.y + c.x + d.y;
        } else if (V == 3) {
            const i64x2 a = v[0];
            acc += a.x + a.y;
        } else {
            const volatile long long *w = reinterpret_cast<const volatile long long *>(v);
            const long long a = w[0], b = w[1];
            acc += a + b;
        }
        h ^= acc;  // the next round's line depends on this round's data
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

int main() {
    const uint32_t n = 125000;
    std::vector<uint64_t> host((size_t)kLines * 8);
    for (size_t i = 0; i < host.size(); ++i) host[i] = i * 0x9E3779B97F4A7C15ull;
    i64x2 *t;
    uint64_t *out;
    hipMalloc(&t, host.size() * 8);
    hipMalloc(&out, 8);
    hipMemcpy(t, host.data(), host.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[5] = {"A four units of one line together", "B first unit, then the other three", "C four units of four lines", "D one unit", "E two 8-byte halves of one unit"};
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 5; ++v) {
            auto launch = [&]() {
                const dim3 g((n + 255) / 256), b(256);
                switch (v) {
                    case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, t, n, out); break;
                    case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, t, n, out); break;
                    case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, t, n, out); break;
                    case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, t, n, out); break;
                    default: hipLaunchKernelGGL(k<4>, g, b, 0, 0, t, n, out); break;
                }
            };
            for (int i = 0; i < 5; ++i) launch();
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 50; ++i) launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-40s %7.2f us per launch (%u lanes x %u dependent rounds) = %.2f us per round\n", names[v], ms * 1e3 / 50, n, kRounds, ms * 1e3 / 50 / kRounds);
        }
    return 0;
}
