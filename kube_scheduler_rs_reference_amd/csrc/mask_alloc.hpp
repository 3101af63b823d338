// mask_alloc.hpp -- where the feasibility masks live (ksched_mask_alloc / ksched_mask_free, include/ksched.h).
//
// The mask is 92 % of the path's HBM traffic and the rate at which the mask kernel runs depends on the PHYSICAL pages behind
// the buffer it writes (profiles/r06_mask_alloc.md: HBM-side write stalls, a ladder of sticky per-allocation rates, no allocation
// path that selects one).  The shipped library allocates masks with hipMalloc -- plainly, or probe-and-keep -- and nothing else.
// The MEASUREMENT paths round 6 compared (HIP's virtual-memory API at either granularity, one contiguous physical range, scattered
// pieces) live in tests/cpp/test_hooks.cpp, i.e. in the TEST build of the library only (tests/cpp/hooks/libksched_hip.so): they
// are how tools/alloc_probe.py re-measures the choice, and one of them showed stale shader reads after another in the same
// process (same file, section 3) -- not something a production library should be able to do on request.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

// defined by tests/cpp/test_hooks.cpp only; null in the shipped library (the calls below then answer KSCHED_E_UNSUPPORTED)
extern "C" __attribute__((weak)) int ksched_test_mask_alloc(int device, size_t bytes, uint32_t how, void **out_ptr, void **out_token);
extern "C" __attribute__((weak)) int ksched_test_mask_release(void *ptr, void *token);

namespace ksched {

struct MaskAllocation {
    void *ptr = nullptr;
    size_t bytes = 0;            // what the caller asked for
    uint32_t how = 0;            // KSCHED_MASK_ALLOC_* that produced it
    int device = 0;
    void *test_token = nullptr;  // a measurement path of the test build: what ksched_test_mask_release needs
    const void *owner = nullptr;  // the ctx that allocated it (ksched_destroy frees what its caller left)
};

// process-wide registry (a mask may outlive the ctx that allocated it only until ksched_mask_free; the ctx's destructor frees what is left)
struct MaskRegistry {
    std::mutex mu;
    std::vector<MaskAllocation> live;
};
inline MaskRegistry &mask_registry() {
    static MaskRegistry r;
    return r;
}

inline hipError_t mask_release(MaskAllocation &a) {
    hipError_t e = hipSuccess;
    if (!a.ptr) return e;
    if (a.test_token) e = (ksched_test_mask_release && ksched_test_mask_release(a.ptr, a.test_token) == 0) ? hipSuccess : hipErrorUnknown;
    else e = hipFree(a.ptr);
    a.ptr = nullptr;
    a.test_token = nullptr;
    return e;
}

}  // namespace ksched
