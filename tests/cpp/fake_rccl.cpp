// fake_rccl.cpp -- TEST-ONLY stand-in for librccl (never shipped, never linked: the library dlopen()s it only when BOTH
// $KSCHED_RCCL_LIB names it and $KSCHED_TEST_HOOKS=1 is set, csrc/comm_rccl.hpp).  A test box has ONE MI355X and RCCL refuses a
// communicator that names a device twice, so the product sequence of the multi-device host
//     ksched_comm_create_local -> ksched_eval_begin x n -> ksched_gather_buffer x n -> ksched_allgather_bindings_local -> ksched_eval_end
// could only ever run with n = 1.  This library implements the eight RCCL entry points that sequence uses with the semantics the
// product relies on, for n "ranks" that may all sit on one physical device:
//   * ncclCommInitAll accepts any device list (duplicates included) and returns n communicators of one clique;
//   * ncclCommInitRank with nranks == 1 is a clique of one; with nranks > 1 it joins a clique of PROCESSES (one per rank, possibly all on one
//     physical device: `bench.py --gpus 2` on a one-GPU box, KSCHED_BENCH_ONE_GPU=1) that meets in a POSIX shared-memory segment named by the
//     unique id.  Their ncclAllGather is BLOCKING: the caller's stream is drained, the contribution goes through the segment, every rank waits
//     (on the host, with a time-out: a rank that died is an error, never a hung GPU) for the others and copies the table back.  Ordered with the
//     stream like the real one, just not asynchronous;
//   * ncclAllGather inside ncclGroupStart / ncclGroupEnd is STREAM-ORDERED like the real one: rank i's stream waits until every
//     rank's stream has reached the collective (the send buffers are then complete), copies the n contributions into its receive
//     buffer (hipMemcpyAsync device -> device, peer copies across devices), and no rank's stream runs past the collective before every
//     rank has finished reading (a send buffer may be overwritten right behind it).  Nothing blocks the host.
//   * $FAKE_RCCL_CORRUPT_RANK=r (clique of processes): rank r's copy of the gathered table gets one wrong word in a peer's part -- what only a
//     comparison ACROSS ranks can notice (bench.py parity_check.gathered_table);
//   * $FAKE_RCCL_FAIL_ALLGATHER=k makes the k-th ncclAllGather call of the process fail (ncclInternalError) -- inside a group that
//     leaves the collective half-issued, which is what the library's abort path is for.
// What it does NOT cover: the xGMI transport, RCCL's own kernels and their interaction with the evaluator's kernels on the CUs.
// Build: make host  (tests/cpp/libfake_rccl.so)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Clique {
    int n = 0;
};
// the meeting place of a clique of processes: [header][slot 0] .. [slot n - 1]; a fresh segment is all zero, which is every field's start value
constexpr int kMaxProcs = 64;
constexpr size_t kSlotBytes = (size_t)16 << 20;  // per rank and collective (bench.py's largest: 500 k bindings x 4 bytes)
constexpr double kWaitSeconds = 60.0;
struct Meeting {
    std::atomic<uint32_t> joined;
    std::atomic<uint64_t> arrived[kMaxProcs];  // collectives this rank has contributed to
    std::atomic<uint64_t> left[kMaxProcs];     // ... and has finished reading
};
static_assert(sizeof(Meeting) <= 4096, "the header page");
constexpr size_t kHeaderBytes = 4096;
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
    // a clique of processes (ncclCommInitRank, nranks > 1)
    Meeting *meeting = nullptr;
    size_t mapped = 0;
    uint64_t seq = 0;
    ~ncclComm() {
        if (meeting) munmap((void *)meeting, mapped);
    }
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

template <class Pred>
bool wait_for(Pred done) {  // host-side, bounded: a peer that died must become an error here, not a wait without end
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; !done(); ++spins) {
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kWaitSeconds) return false;
    }
    return true;
}

// the all-gather of a clique of processes: blocking (header comment)
ncclResult_t gather_across_processes(ncclComm *c, const void *send, void *recv, size_t bytes, hipStream_t stream) {
    if (bytes > kSlotBytes) return ncclInvalidArgument;
    const int n = c->clique->n;
    Meeting *m = c->meeting;
    char *slots = reinterpret_cast<char *>(m) + kHeaderBytes;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(c->device) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = ncclSuccess;
    const uint64_t seq = ++c->seq;
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(slots + (size_t)c->rank * kSlotBytes, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) r = ncclUnhandledCudaError;
    m->arrived[c->rank].store(seq, std::memory_order_release);  // (also after a failed copy: the peers must not wait for the time-out)
    if (!wait_for([&] {
            for (int j = 0; j < n; ++j)
                if (m->arrived[j].load(std::memory_order_acquire) < seq) return false;
            return true;
        }))
        r = ncclSystemError;
    for (int j = 0; j < n && r == ncclSuccess; ++j)
        if (hipMemcpy(static_cast<char *>(recv) + (size_t)j * bytes, slots + (size_t)j * kSlotBytes, bytes, hipMemcpyHostToDevice) != hipSuccess) r = ncclUnhandledCudaError;
    if (const char *bad = std::getenv("FAKE_RCCL_CORRUPT_RANK"))  // fault injection: THIS rank receives a wrong word in a peer's part of the table
        if (r == ncclSuccess && n > 1 && std::atoi(bad) == c->rank) {
            const int32_t wrong = -7;
            (void)hipMemcpy(static_cast<char *>(recv) + (size_t)((c->rank + 1) % n) * bytes, &wrong, sizeof wrong, hipMemcpyHostToDevice);
        }
    m->left[c->rank].store(seq, std::memory_order_release);
    if (!wait_for([&] {  // nobody writes its next contribution before everybody has read this one
            for (int j = 0; j < n; ++j)
                if (m->left[j].load(std::memory_order_acquire) < seq) return false;
            return true;
        }) && r == ncclSuccess)
        r = ncclSystemError;
    (void)hipSetDevice(prev);
    ++g_collectives;
    return r;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    static std::atomic<unsigned> counter{1};
    std::memset(id, 0, sizeof *id);
    // the name of the segment a clique of processes meets in (unused by a clique of one)
    std::snprintf(id->internal, sizeof id->internal, "/fake-rccl-%ld-%u-%llx", (long)getpid(), counter++,
                  (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxProcs || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    std::unique_ptr<ncclComm> c(new ncclComm());
    c->clique = std::make_shared<Clique>();
    c->clique->n = nranks;
    c->rank = rank;
    c->device = dev;
    if (nranks > 1) {  // one process per rank: meet in the segment the id names
        id.internal[sizeof id.internal - 1] = 0;
        if (std::strncmp(id.internal, "/fake-rccl-", 11) != 0) return ncclInvalidArgument;
        const size_t total = kHeaderBytes + (size_t)nranks * kSlotBytes;
        const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
        if (fd < 0) return ncclSystemError;
        if (ftruncate(fd, (off_t)total) != 0) {  // (every rank sets the same size; a fresh segment reads as zeros)
            close(fd);
            return ncclSystemError;
        }
        void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return ncclSystemError;
        c->meeting = static_cast<Meeting *>(p);
        c->mapped = total;
        c->meeting->joined.fetch_add(1);
        const bool all = wait_for([&] { return c->meeting->joined.load() >= (uint32_t)nranks; });
        if (rank == 0) shm_unlink(id.internal);  // (the mappings keep it alive; nothing is left behind in /dev/shm)
        if (!all) return ncclSystemError;
    }
    *comm = c.release();
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
    if (comm->meeting) return gather_across_processes(comm, sendbuff, recvbuff, sendcount * 4u, stream);  // (a clique of processes: blocking, grouped or not)
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
        case ncclSystemError: return "system error (fake RCCL: the shared segment, or a rank of the clique did not show up within the time-out)";
        default: return "error (fake RCCL)";
    }
}

// test observability: how many collectives have actually been enqueued by this library instance
long fake_rccl_collectives(void) { return g_collectives.load(); }
// ... and how many ncclAllGather calls it has seen ($FAKE_RCCL_FAIL_ALLGATHER counts the same calls)
long fake_rccl_allgather_calls(void) { return g_allgathers.load(); }

}  // extern "C"
