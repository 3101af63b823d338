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
    std::vector<hipMemGenericAllocationHandle_t> pieces;  // the scattered form: one handle per piece, mapped at consecutive addresses
    size_t piece_bytes = 0;
    const void *owner = nullptr;               // the ctx that allocated it (ksched_destroy frees what its caller left)
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

// Scattered: the buffer is built from `piece`-sized physical allocations made one by one -- more of them than needed are created, in an
// order shuffled by a fixed generator, and the surplus is released -- and mapped at consecutive virtual addresses: what the memory
// channels see behind a linear sweep of the mask is then a pseudo-random walk over the pieces instead of one physical run
// (profiles/r06_mask_alloc.md: one contiguous physical range is reliably in the SLOW half of the rates).
inline hipError_t mask_alloc_scattered(int device, size_t bytes, size_t piece, uint32_t surplus_pct, MaskAllocation *out) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran == 0) gran = 4096;
    piece = round_up(std::max(piece, gran), gran);
    const size_t n = (std::max<size_t>(bytes, 1) + piece - 1) / piece, total = n * piece;
    const size_t make = n + n * surplus_pct / 100;
    std::vector<hipMemGenericAllocationHandle_t> all;
    all.reserve(make);
    auto drop = [&](size_t from) {
        for (size_t i = from; i < all.size(); ++i) (void)hipMemRelease(all[i]);
        all.resize(std::min(all.size(), from));
    };
    for (size_t i = 0; i < make; ++i) {
        hipMemGenericAllocationHandle_t h{};
        e = hipMemCreate(&h, piece, &prop, 0);
        if (e != hipSuccess) {
            if (all.size() >= n) break;  // the surplus is optional
            drop(0);
            return e;
        }
        all.push_back(h);
    }
    // Fisher-Yates with a fixed xorshift: which pieces are kept and in which order they are mapped
    uint64_t x = 0x9E3779B97F4A7C15ull ^ (uint64_t)all.size();
    for (size_t i = all.size(); i > 1; --i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        std::swap(all[i - 1], all[(size_t)(x % i)]);
    }
    drop(n);
    void *va = nullptr;
    e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
    if (e != hipSuccess) {
        drop(0);
        return e;
    }
    size_t mapped = 0;
    for (; mapped < n && e == hipSuccess; ++mapped) e = hipMemMap((char *)va + mapped * piece, piece, 0, all[mapped], 0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc{};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, total, &acc, 1);
    } else {
        --mapped;  // the piece whose map failed
    }
    if (e != hipSuccess) {
        for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap((char *)va + i * piece, piece);
        (void)hipMemAddressFree(va, total);
        drop(0);
        return e;
    }
    out->ptr = va;
    out->mapped = total;
    out->vmm = true;
    out->pieces = std::move(all);
    out->piece_bytes = piece;
    return hipSuccess;
}

inline hipError_t mask_release(MaskAllocation &a) {
    hipError_t e = hipSuccess;
    if (!a.ptr) return e;
    if (a.vmm && !a.pieces.empty()) {
        for (size_t i = 0; i < a.pieces.size(); ++i) {
            hipError_t e1 = hipMemUnmap((char *)a.ptr + i * a.piece_bytes, a.piece_bytes);
            if (e == hipSuccess) e = e1;
        }
        hipError_t e2 = hipMemAddressFree(a.ptr, a.mapped);
        if (e == hipSuccess) e = e2;
        for (auto h : a.pieces) {
            hipError_t e3 = hipMemRelease(h);
            if (e == hipSuccess) e = e3;
        }
        a.pieces.clear();
    } else if (a.vmm) {
        e = hipMemUnmap(a.ptr, a.mapped);
        hipError_t e2 = hipMemAddressFree(a.ptr, a.mapped);
        hipError_t e3 = hipMemRelease(a.handle);
        if (e == hipSuccess) e = e2;
        if (e == hipSuccess) e = e3;
    } else {
        e = hipFree(a.ptr);
    }
    a.ptr = nullptr;
    return e;
}

}  // namespace ksched
