// test_hooks.cpp -- linked ONLY into tests/cpp/hooks/libksched_hip.so, the TEST build of the evaluator library (make test-lib).
//
// The shipped kube_scheduler_rs_reference_amd/libksched_hip.so does not contain this file: in it the weak reference to
// ksched_test_hooks_enabled stays null, `$KSCHED_RCCL_LIB` is never read, KSCHED_OPT_FAULT answers KSCHED_E_UNSUPPORTED and the host mirror
// never multiplies one device into k replicas -- a production library cannot be redirected or made to throw by two environment variables
// (ADVICE r5).  In the test build the hooks are still off unless $KSCHED_TEST_HOOKS=1.
#include <cstdlib>
#include <cstring>

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
