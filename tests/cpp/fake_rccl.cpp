// fake_rccl.cpp -- TEST-ONLY stand-in for librccl (never shipped, never linked: the library dlopen()s it only when BOTH
// $KSCHED_RCCL_LIB names it and $KSCHED_TEST_HOOKS=1 is set, csrc/comm_rccl.hpp).  A test box has ONE MI355X and RCCL refuses a
// communicator that names a device twice, so the product sequence of the multi-device host
//     ksched_comm_create_local -> ksched_eval_begin x n -> ksched_gather_buffer x n -> ksched_allgather_bindings_local -> ksched_eval_end
// could only ever run with n = 1.  This library implements the eight RCCL entry points that sequence uses with the semantics the
// product relies on, for n "ranks" that may all sit on one physical device:
//   * ncclCommInitAll accepts any device list (duplicates included) and returns n communicators of one clique;
//   * ncclCommInitRank accepts nranks == 1 only (a second process cannot be reached without the real transport);
//   * ncclAllGather inside ncclGroupStart / ncclGroupEnd is STREAM-ORDERED like the real one: rank i's stream waits until every
//     rank's stream has reached the collective (the send buffers are then complete), copies the n contributions into its receive
//     buffer (hipMemcpyAsync device -> device, peer copies across devices), and no rank's stream runs past the collective before every
//     rank has finished reading (a send buffer may be overwritten right behind it).  Nothing blocks the host.
//   * $FAKE_RCCL_FAIL_ALLGATHER=k makes the k-th ncclAllGather call of the process fail (ncclInternalError) -- inside a group that
//     leaves the collective half-issued, which is what the library's abort path is for.
// What it does NOT cover: the xGMI transport, RCCL's own kernels and their interaction with the evaluator's kernels on the CUs.
// Build: make host  (tests/cpp/libfake_rccl.so)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct Clique {
    int n = 0;
};
struct Pending {
    ncclComm_t comm;
    const void *send;
    void *recv;
    size_t bytes;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Pending> g_pending;
std::atomic<long> g_allgathers{0};
std::atomic<long> g_collectives{0};

}  // namespace

struct ncclComm {  // (opaque in rccl.h)
    std::shared_ptr<Clique> clique;
    int rank = 0, device = 0;
    bool aborted = false;
};

namespace {

ncclResult_t run_collective(std::vector<Pending> &ps) {
    // every rank of one clique exactly once, same byte count
    if (ps.empty()) return ncclSuccess;
    const int n = ps[0].comm->clique->n;
    if ((int)ps.size() != n) return ncclInvalidUsage;
    std::vector<const Pending *> by_rank((size_t)n, nullptr);
    for (const Pending &p : ps) {
        if (p.comm->clique != ps[0].comm->clique || p.bytes != ps[0].bytes || p.comm->aborted) return ncclInvalidUsage;
        if (by_rank[(size_t)p.comm->rank]) return ncclInvalidUsage;
        by_rank[(size_t)p.comm->rank] = &p;
    }
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<hipEvent_t> ready((size_t)n), done((size_t)n);
    bool ok = true;
    for (int j = 0; j < n && ok; ++j) {
        ok = hipSetDevice(by_rank[(size_t)j]->comm->device) == hipSuccess && hipEventCreateWithFlags(&ready[(size_t)j], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&done[(size_t)j], hipEventDisableTiming) == hipSuccess && hipEventRecord(ready[(size_t)j], by_rank[(size_t)j]->stream) == hipSuccess;
    }
    for (int i = 0; i < n && ok; ++i) {
        const Pending &me = *by_rank[(size_t)i];
        ok = hipSetDevice(me.comm->device) == hipSuccess;
        for (int j = 0; j < n && ok; ++j)
            if (j != i) ok = hipStreamWaitEvent(me.stream, ready[(size_t)j], 0) == hipSuccess;
        for (int j = 0; j < n && ok; ++j) {
            const Pending &src = *by_rank[(size_t)j];
            char *dst = static_cast<char *>(me.recv) + (size_t)j * me.bytes;
            if (dst == src.send) continue;  // in place
            if (src.comm->device == me.comm->device) ok = hipMemcpyAsync(dst, src.send, me.bytes, hipMemcpyDeviceToDevice, me.stream) == hipSuccess;
            else ok = hipMemcpyPeerAsync(dst, me.comm->device, src.send, src.comm->device, me.bytes, me.stream) == hipSuccess;
        }
        if (ok) ok = hipEventRecord(done[(size_t)i], me.stream) == hipSuccess;
    }
    for (int j = 0; j < n && ok; ++j) {  // nobody runs past the collective before everybody has read
        ok = hipSetDevice(by_rank[(size_t)j]->comm->device) == hipSuccess;
        for (int i = 0; i < n && ok; ++i)
            if (i != j) ok = hipStreamWaitEvent(by_rank[(size_t)j]->stream, done[(size_t)i], 0) == hipSuccess;
    }
    for (int j = 0; j < n; ++j) {  // (an event may be destroyed while still pending: its resources go when it has fired)
        (void)hipEventDestroy(ready[(size_t)j]);
        (void)hipEventDestroy(done[(size_t)j]);
    }
    (void)hipSetDevice(prev);
    ++g_collectives;
    return ok ? ncclSuccess : ncclUnhandledCudaError;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    static std::atomic<unsigned> counter{1};
    std::memset(id, 0, sizeof *id);
    const unsigned v = counter++;
    std::memcpy(id->internal, "fake-rccl", 9);
    std::memcpy(id->internal + 16, &v, sizeof v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank) {
    if (!comm || nranks != 1 || rank != 0) return ncclInvalidArgument;  // one process per GPU needs the real transport
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    ncclComm *c = new ncclComm();
    c->clique = std::make_shared<Clique>();
    c->clique->n = 1;
    c->device = dev;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    if (!comms || ndev <= 0) return ncclInvalidArgument;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess) return ncclUnhandledCudaError;
    auto clique = std::make_shared<Clique>();
    clique->n = ndev;
    for (int i = 0; i < ndev; ++i) {
        const int dev = devlist ? devlist[i] : i;
        if (dev < 0 || dev >= visible) {
            for (int j = 0; j < i; ++j) delete comms[j];
            return ncclInvalidArgument;
        }
        ncclComm *c = new ncclComm();
        c->clique = clique;
        c->rank = i;
        c->device = dev;
        comms[i] = c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Pending> ps;
    ps.swap(g_pending);
    return run_collective(ps);
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    const long call = ++g_allgathers;
    if (const char *f = std::getenv("FAKE_RCCL_FAIL_ALLGATHER"))
        if (std::atol(f) == call) return ncclInternalError;
    if (!comm || comm->aborted || !sendbuff || !recvbuff) return ncclInvalidArgument;
    if (datatype != ncclInt32 && datatype != ncclUint32 && datatype != ncclFloat32) return ncclInvalidArgument;  // 4-byte elements only
    Pending p{comm, sendbuff, recvbuff, sendcount * 4u, stream};
    if (g_depth > 0) {
        g_pending.push_back(p);
        return ncclSuccess;
    }
    std::vector<Pending> one{p};
    return run_collective(one);  // ungrouped: a clique of one
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (fake RCCL)";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake RCCL)";
        case ncclInternalError: return "internal error (fake RCCL: injected)";
        case ncclInvalidArgument: return "invalid argument (fake RCCL)";
        case ncclInvalidUsage: return "invalid usage (fake RCCL: the group does not hold every rank of the clique exactly once)";
        default: return "error (fake RCCL)";
    }
}

// test observability: how many collectives have actually been enqueued by this library instance
long fake_rccl_collectives(void) { return g_collectives.load(); }
// ... and how many ncclAllGather calls it has seen ($FAKE_RCCL_FAIL_ALLGATHER counts the same calls)
long fake_rccl_allgather_calls(void) { return g_allgathers.load(); }

}  // extern "C"
