// kernarg.hpp -- kernarg_warm<BYTES>(): touch every 64-byte line of the kernel-argument segment with ONE group of scalar loads.
//
// The compiler fetches kernel arguments where a basic block first needs them: the mask kernel's prologue made four dependent
// groups of s_load (block mapping -> early exit -> pointers -> operand addresses), the sampled pick's three to four.  Every
// group that touches a new line of the segment is a scalar-cache miss: 0.20-0.24 us each on MI355X, back-to-back launches,
// kernarg segment in device memory; a line that has been touched costs 0.04 us (tools/ubench_kernarg.hip,
// profiles/r02_m_kernarg_latency.txt).  One early group of loads (one per line, results discarded) turns the later groups into
// hits: one miss latency per launch instead of three or four.  No registers stay reserved.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef KSCHED_KERNARG_WARM
#define KSCHED_KERNARG_WARM 1
#endif

namespace ksched {

template <uint32_t BYTES>
__device__ __forceinline__ void kernarg_warm() {
#if KSCHED_KERNARG_WARM && defined(__HIP_DEVICE_COMPILE__)
    static_assert(BYTES <= 512, "eight lines at most");
    const uint64_t kp = (uint64_t)(const __attribute__((address_space(4))) void *)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t d0, d1, d2, d3, d4, d5, d6, d7;
    // (the segment is at least 64-byte aligned in practice; were it not, a line's tail could be missed -- speed only)
    if constexpr (BYTES <= 64)
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0) : "s"(kp));
    else if constexpr (BYTES <= 128)
        asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1) : "s"(kp));
    else if constexpr (BYTES <= 192)
        asm volatile("s_load_dword %0, %3, 0x0\n\ts_load_dword %1, %3, 0x40\n\ts_load_dword %2, %3, 0x80\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2)
                     : "s"(kp));
    else if constexpr (BYTES <= 256)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3)
                     : "s"(kp));
    else if constexpr (BYTES <= 384)
        asm volatile("s_load_dword %0, %6, 0x0\n\ts_load_dword %1, %6, 0x40\n\ts_load_dword %2, %6, 0x80\n\ts_load_dword %3, %6, 0xc0\n\t"
                     "s_load_dword %4, %6, 0x100\n\ts_load_dword %5, %6, 0x140\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5)
                     : "s"(kp));
    else
        asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\ts_load_dword %3, %8, 0xc0\n\t"
                     "s_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\ts_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7)
                     : "s"(kp));
#endif
}

}  // namespace ksched
