// test_hooks.cpp -- linked ONLY into tests/cpp/hooks/libksched_hip.so, the TEST build of the evaluator library (make test-lib).
//
// The shipped kube_scheduler_rs_reference_amd/libksched_hip.so does not contain this file: in it the weak reference to
// ksched_test_hooks_enabled stays null, `$KSCHED_RCCL_LIB` is never read, KSCHED_OPT_FAULT answers KSCHED_E_UNSUPPORTED and the host mirror
// never multiplies one device into k replicas -- a production library cannot be redirected or made to throw by two environment variables
// (ADVICE r5).  In the test build the hooks are still off unless $KSCHED_TEST_HOOKS=1.
#include <cstdlib>
#include <cstring>
#include <new>

#include <algorithm>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/ksched.h"

extern "C" int ksched_test_hooks_enabled(void) {
    const char *e = std::getenv("KSCHED_TEST_HOOKS");
    return e && std::strcmp(e, "1") == 0;
}
// the test build answers this whatever the environment says: "is this the library with the hooks linked in?"
extern "C" int ksched_test_hooks_linked(void) { return 1; }

// The RCCL stand-in's path ($KSCHED_RCCL_LIB), or null.  *refused is set when the variable names a library but the switch is off: the
// test build answers that with an error instead of a silent substitute (tests/test_host_mirror.py).
extern "C" const char *ksched_test_rccl_lib(int *refused) {
    const char *over = std::getenv("KSCHED_RCCL_LIB");
    if (refused) *refused = 0;
    if (!over) return nullptr;
    if (!ksched_test_hooks_enabled()) {
        if (refused) *refused = 1;
        return nullptr;
    }
    return over;
}

// ---- the MEASUREMENT paths of ksched_mask_alloc (profiles/r06_mask_alloc.md section 3; tools/alloc_probe.py) ---------------------------
// None of them selects the fast placement; kept so that the comparison can be repeated.  KSCHED_MASK_ALLOC_VMM / _VMM_MIN: hipMemCreate in one piece
// at the recommended / minimum granularity + hipMemAddressReserve + hipMemMap.  _CONTIGUOUS: one physical range.  _SCATTER_2M / _16M: the buffer
// is built from pieces created one by one -- a quarter more than needed, in an order shuffled by a fixed generator, the surplus released -- and
// mapped at consecutive virtual addresses.
namespace {
struct TestMask {
    bool vmm = false;
    size_t mapped = 0, piece = 0;
    std::vector<hipMemGenericAllocationHandle_t> pieces;
};
size_t round_up(size_t v, size_t a) { return a ? (v + a - 1) / a * a : v; }

hipError_t map_pieces(int device, size_t bytes, size_t piece, bool recommended, uint32_t surplus_pct, void **out, TestMask *m) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, recommended ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran == 0) gran = 4096;
    if (piece == 0) piece = round_up(std::max<size_t>(bytes, 1), gran);  // ONE piece
    piece = round_up(std::max(piece, gran), gran);
    const size_t n = (std::max<size_t>(bytes, 1) + piece - 1) / piece, total = n * piece, make = n + n * surplus_pct / 100;
    std::vector<hipMemGenericAllocationHandle_t> all;
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
    unsigned long long x = 0x9E3779B97F4A7C15ull ^ (unsigned long long)all.size();  // Fisher-Yates with a fixed xorshift
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
    for (; mapped < n; ++mapped)
        if ((e = hipMemMap((char *)va + mapped * piece, piece, 0, all[mapped], 0)) != hipSuccess) break;
    if (e == hipSuccess) {
        hipMemAccessDesc acc{};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, total, &acc, 1);
    }
    if (e != hipSuccess) {
        for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap((char *)va + i * piece, piece);
        (void)hipMemAddressFree(va, total);
        drop(0);
        return e;
    }
    *out = va;
    m->vmm = true;
    m->mapped = total;
    m->piece = piece;
    m->pieces = std::move(all);
    return hipSuccess;
}
}  // namespace

// 0 = done (*out_ptr, *out_token set); a positive hipError_t otherwise; -1 = not a measurement path
extern "C" int ksched_test_mask_alloc(int device, size_t bytes, uint32_t how, void **out_ptr, void **out_token) {
    if (!out_ptr || !out_token) return -1;
    TestMask *m = new (std::nothrow) TestMask();
    if (!m) return (int)hipErrorOutOfMemory;
    hipError_t e = hipErrorInvalidValue;
    switch (how) {
        case KSCHED_MASK_ALLOC_VMM: e = map_pieces(device, bytes, 0, true, 0, out_ptr, m); break;
        case KSCHED_MASK_ALLOC_VMM_MIN: e = map_pieces(device, bytes, 0, false, 0, out_ptr, m); break;
        case KSCHED_MASK_ALLOC_SCATTER_2M: e = map_pieces(device, bytes, 2u << 20, false, 25, out_ptr, m); break;
        case KSCHED_MASK_ALLOC_SCATTER_16M: e = map_pieces(device, bytes, 16u << 20, false, 25, out_ptr, m); break;
        case KSCHED_MASK_ALLOC_CONTIGUOUS: e = hipExtMallocWithFlags(out_ptr, bytes, hipDeviceMallocContiguous); break;
        default:
            delete m;
            return -1;
    }
    if (e != hipSuccess) {
        delete m;
        (void)hipGetLastError();
        return (int)e;
    }
    *out_token = m;
    return 0;
}

extern "C" int ksched_test_mask_release(void *ptr, void *token) {
    TestMask *m = static_cast<TestMask *>(token);
    if (!m || !ptr) return -1;
    hipError_t e = hipSuccess;
    if (m->vmm) {
        for (size_t i = 0; i < m->pieces.size(); ++i) {
            const hipError_t e1 = hipMemUnmap((char *)ptr + i * m->piece, m->piece);
            if (e == hipSuccess) e = e1;
        }
        const hipError_t e2 = hipMemAddressFree(ptr, m->mapped);
        if (e == hipSuccess) e = e2;
        for (auto h : m->pieces) {
            const hipError_t e3 = hipMemRelease(h);
            if (e == hipSuccess) e = e3;
        }
    } else {
        e = hipFree(ptr);
    }
    delete m;
    return e == hipSuccess ? 0 : (int)e;
}
