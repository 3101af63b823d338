// comm_rccl.hpp -- the one exchange step of the path, behind the C ABI: an RCCL all-gather of the int32 (pod -> node)
// bindings over xGMI (pod rows shard contiguously over the GPUs of one node, the node snapshot is
// replicated, masks stay where they were produced; 4 B per pod are exchanged).
//
// RCCL is resolved at run time (dlopen of librccl.so.1), not at link time: the evaluator itself has no use for it, a
// single-GPU caller never loads it, and inside a process that already carries an RCCL (e.g. PyTorch's bundled
// librccl.so.1) the same library instance is reused instead of a second copy being mapped next to it.  The header is
// only used for its types and prototypes.
//
// Test hook, in the TEST build of the library only (tests/cpp/hooks/libksched_hip.so = the same object code + tests/cpp/test_hooks.cpp, which
// defines ksched_test_hooks_enabled): with $KSCHED_TEST_HOOKS=1 and $KSCHED_RCCL_LIB=<path> that library is loaded instead of RCCL --
// tests/cpp/fake_rccl.cpp, which lets one GPU stand for n ranks so that the multi-device host's exchange runs with n > 1 on a one-GPU box.
// The shipped library does not define the symbol: the weak reference below is null there and $KSCHED_RCCL_LIB is never read.
#pragma once
#include <dlfcn.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

extern "C" __attribute__((weak)) int ksched_test_hooks_enabled(void);  // tests/cpp/test_hooks.cpp (the test build) or null (the shipped library)
extern "C" __attribute__((weak)) const char *ksched_test_rccl_lib(int *refused);  // the same: the stand-in's path, read from the environment THERE

namespace ksched {

inline bool test_hooks_on() { return ksched_test_hooks_enabled != nullptr && ksched_test_hooks_enabled() != 0; }

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;  // optional: used to give up a clique whose collective was only half issued
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    std::string error;
    std::string substitute;  // path of the test stand-in, empty = the real RCCL
};

inline RcclApi &rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        int refused = 0;
        const char *over = ksched_test_rccl_lib != nullptr ? ksched_test_rccl_lib(&refused) : nullptr;  // (the shipped library has no such function)
        if (refused) {
            api.error = "a substitute for RCCL is named in the environment but the test hooks are not switched on: refusing a substitute for RCCL";
            return;
        }
        if (over) {
            h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!h) {
                api.error = std::string("cannot load the RCCL stand-in: ") + dlerror();
                return;
            }
            api.substitute = over;
        } else {
            const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char *n : names)
                if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
            if (!h) {
                api.error = std::string("cannot load librccl.so.1: ") + dlerror();
                return;
            }
        }
        bool all = true;
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(h, name);
            if (!p) {
                all = false;
                api.error = std::string("librccl lacks ") + name;
            }
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.ok = all;
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(h, "ncclCommAbort"));
    });
    return api;
}

}  // namespace ksched
