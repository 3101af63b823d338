// kernels_indexed.hpp -- "indexed" mask kernel: LDS-resident per-tile bitmap index (placeholder).
//
// The interface the API layer programs against; the kernel itself lands in a later commit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ksched {

struct IndexedSnapshot {
    bool built = false;
};

inline hipError_t indexed_build(IndexedSnapshot &s, uint32_t, const int64_t *, const int64_t *, const uint32_t *, uint32_t,
                                const uint64_t *) {
    s.built = false;
    return hipSuccess;
}
inline void indexed_release(IndexedSnapshot &) {}
inline bool indexed_applicable(const IndexedSnapshot &s, uint32_t, bool) { return s.built; }
inline size_t indexed_scratch_bytes(const IndexedSnapshot &, uint32_t) { return 0; }
inline hipError_t run_indexed(const IndexedSnapshot &, uint32_t, const int64_t *, const int64_t *, const uint32_t *,
                              const uint64_t *, uint32_t, uint64_t *, uint64_t *, uint8_t *, hipStream_t) {
    return hipErrorNotSupported;
}

}  // namespace ksched
