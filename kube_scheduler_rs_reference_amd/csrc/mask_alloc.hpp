// mask_alloc.hpp -- where the feasibility masks live (ksched_mask_alloc / ksched_mask_free, include/ksched.h).
//
// The mask is 92 % of the path's HBM traffic and the rate at which the mask kernel runs depends on the PHYSICAL pages behind
// the buffer it writes (profiles/r05_bimodal_by_allocation.md: C5 shard 141 us or 172 us by allocation, nothing in between;
// profiles/r06_mask_alloc.md: what separates the two and which allocation path selects the fast one).  A caller that lets the
// library allocate the masks gets the placement the measurements chose; the `how` values below exist so that the choice can be
// re-measured on other hardware (tools/alloc_probe.py).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

namespace ksched {

struct MaskAllocation {
    void *ptr = nullptr;
    size_t bytes = 0;      // what the caller asked for
    size_t mapped = 0;     // what was mapped / allocated
    uint32_t how = 0;      // KSCHED_MASK_ALLOC_* that produced it
    int device = 0;
    hipMemGenericAllocationHandle_t handle{};  // VMM forms
    bool vmm = false;
    bool pooled = false;                       // pool form (the device's pool below)
    const void *owner = nullptr;               // the ctx that allocated it (ksched_destroy frees what its caller left)
};

// process-wide registry (a mask may outlive the ctx that allocated it only until ksched_mask_free; the ctx's destructor frees what is left)
struct MaskRegistry {
    std::mutex mu;
    std::vector<MaskAllocation> live;
    std::vector<std::pair<int, hipMemPool_t>> pools;  // one per device, release threshold = never
};
inline MaskRegistry &mask_registry() {
    static MaskRegistry r;
    return r;
}

inline size_t round_up(size_t v, size_t a) { return a ? (v + a - 1) / a * a : v; }

// VMM: physical memory created in ONE piece of `gran`-multiples and mapped at a VA aligned to `va_align`, so that the page tables can describe
// it with the largest fragments the pieces allow.
inline hipError_t mask_alloc_vmm(int device, size_t bytes, size_t va_align, bool recommended, MaskAllocation *out) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, recommended ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran == 0) gran = 2u << 20;
    const size_t sz = round_up(bytes ? bytes : 1, std::max(gran, va_align ? std::min<size_t>(va_align, 2u << 20) : gran));
    hipMemGenericAllocationHandle_t h{};
    e = hipMemCreate(&h, sz, &prop, 0);
    if (e != hipSuccess) return e;
    void *va = nullptr;
    e = hipMemAddressReserve(&va, sz, va_align, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipMemRelease(h);
        return e;
    }
    e = hipMemMap(va, sz, 0, h, 0);
    if (e != hipSuccess) {
        (void)hipMemAddressFree(va, sz);
        (void)hipMemRelease(h);
        return e;
    }
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(va, sz, &acc, 1);
    if (e != hipSuccess) {
        (void)hipMemUnmap(va, sz);
        (void)hipMemAddressFree(va, sz);
        (void)hipMemRelease(h);
        return e;
    }
    out->ptr = va;
    out->mapped = sz;
    out->handle = h;
    out->vmm = true;
    return hipSuccess;
}

inline hipError_t mask_release(MaskAllocation &a) {
    hipError_t e = hipSuccess;
    if (!a.ptr) return e;
    if (a.vmm) {
        e = hipMemUnmap(a.ptr, a.mapped);
        hipError_t e2 = hipMemAddressFree(a.ptr, a.mapped);
        hipError_t e3 = hipMemRelease(a.handle);
        if (e == hipSuccess) e = e2;
        if (e == hipSuccess) e = e3;
    } else if (a.pooled) {
        e = hipFreeAsync(a.ptr, nullptr);
        hipError_t e2 = hipStreamSynchronize(nullptr);
        if (e == hipSuccess) e = e2;
    } else {
        e = hipFree(a.ptr);
    }
    a.ptr = nullptr;
    return e;
}

}  // namespace ksched
