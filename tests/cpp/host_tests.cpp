// host_tests.cpp -- tests of the host mirror (kube_scheduler_rs_reference_amd/host), written to read like
// the reference's own src/predicates/test.rs: the same two fixtures (test_pod, test_node), the same three
// known answers (KAT-S1..S3), then the derived vectors of SURVEY.md section 8c.
//
//   host_tests cpu   wire-format and host logic only (no device is touched)
//   host_tests gpu   everything that evaluates a predicate: runs on the MI355X through the C ABI
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>  // device buffers / streams for the *_device entry points of the C ABI (tests only)

#include "../../kube_scheduler_rs_reference_amd/host/batcher.hpp"
#include "../../kube_scheduler_rs_reference_amd/host/encoder.hpp"
#include "../../kube_scheduler_rs_reference_amd/host/predicates.hpp"
#include "../../kube_scheduler_rs_reference_amd/host/sharded.hpp"
#include <dlfcn.h>
#include "../../kube_scheduler_rs_reference_amd/host/scheduler.hpp"
#include "../../kube_scheduler_rs_reference_amd/host/pool.hpp"
#include "../../kube_scheduler_rs_reference_amd/host/util.hpp"

using namespace ksched_host;
using predicates::InvalidNodeReason;

static int g_fail = 0, g_run = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            ++g_fail;                                                            \
            std::printf("    FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
        }                                                                        \
    } while (0)
#define CHECK_THROWS(expr)                                                      \
    do {                                                                        \
        bool threw_ = false;                                                    \
        try {                                                                   \
            (void)(expr);                                                       \
        } catch (const std::exception &) {                                      \
            threw_ = true;                                                      \
        }                                                                       \
        if (!threw_) {                                                          \
            ++g_fail;                                                           \
            std::printf("    FAILED %s:%d: no exception: %s\n", __FILE__, __LINE__, #expr); \
        }                                                                       \
    } while (0)

static void run(const char *name, const std::function<void()> &f) {
    ++g_run;
    const int before = g_fail;
    try {
        f();
    } catch (const std::exception &e) {
        ++g_fail;
        std::printf("    EXCEPTION in %s: %s\n", name, e.what());
    }
    std::printf("%s %s\n", g_fail == before ? "ok  " : "FAIL", name);
}

// ---- fixtures, as in src/predicates/test.rs:8-40 -------------------------------------------------------
static const char *POD_NAMESPACE = "test";
static const char *POD_NAME = "pod1";
static const char *NODE_NAME = "node1";

static corev1::Pod test_pod(const char *k = nullptr, const char *v = nullptr) {  // #[default(None)] selector_key
    corev1::Pod pod;
    pod.metadata.namespace_ = POD_NAMESPACE;
    pod.metadata.name = POD_NAME;
    if (k) {
        corev1::PodSpec spec;
        spec.node_selector = corev1::StringMap{{k, v}};
        pod.spec = spec;
    }
    return pod;
}

static corev1::Node test_node() {
    corev1::Node node;
    node.metadata.name = NODE_NAME;
    node.metadata.labels = corev1::StringMap{{"name", NODE_NAME}};
    return node;
}

// ---- helpers for the derived vectors ---------------------------------------------------------------------
static corev1::Container container(const char *cpu, const char *mem) {
    corev1::Container c;
    c.name = "c";
    corev1::ResourceRequirements rr;
    std::map<std::string, corev1::Quantity> req;
    if (cpu) req["cpu"] = cpu;
    if (mem) req["memory"] = mem;
    rr.requests = req;
    c.resources = rr;
    return c;
}

static corev1::Pod pod_with(const std::string &name, std::vector<corev1::Container> cs, const char *node_name = nullptr) {
    corev1::Pod p;
    p.metadata.namespace_ = POD_NAMESPACE;
    p.metadata.name = name;
    corev1::PodSpec spec;
    spec.containers = std::move(cs);
    if (node_name) spec.node_name = std::string(node_name);
    p.spec = spec;
    return p;
}

static corev1::Node node_with(const std::string &name, const char *cpu, const char *mem) {
    corev1::Node n;
    n.metadata.name = name;
    if (cpu || mem) {
        corev1::NodeStatus st;
        std::map<std::string, corev1::Quantity> al;
        if (cpu) al["cpu"] = cpu;
        if (mem) al["memory"] = mem;
        st.allocatable = al;
        n.status = st;
    }
    return n;
}

static Context make_ctx(std::vector<corev1::Node> nodes, std::vector<corev1::Pod> cluster_pods = {}) {
    Context ctx;
    ctx.warn = nullptr;  // (the WARN level is off unless a test listens: hundreds of rejected candidates would drown the test output)
    auto lister = std::make_shared<StaticPodLister>();
    lister->pods = std::move(cluster_pods);
    ctx.client = lister;
    ctx.node_store = std::move(nodes);
    return ctx;
}

struct RecordingSink : BindingSink {
    std::vector<std::pair<std::string, std::string>> posts;  // pod full name -> node
    bool fail = false;
    bool create_pod_binding(const std::string &pod_name, const std::string &ns, const Binding &b) override {
        if (fail) return false;
        posts.push_back({ns + "/" + pod_name, b.target_name});
        return true;
    }
};

// =========================================== CPU-side tests ================================================
static void cpu_tests() {
    run("quantity: canonical domain D (SURVEY.md 8c)", [] {
        CHECK(ParsedQuantity::try_from("500m").to_milli() == 500);
        CHECK(ParsedQuantity::try_from("2").to_milli() == 2000);
        CHECK(ParsedQuantity::try_from("0").to_units() == 0);
        CHECK(ParsedQuantity::try_from("134217728").to_units() == 134217728);
        CHECK(ParsedQuantity::try_from("1k").to_units() == 1000);
        CHECK(ParsedQuantity::try_from("3M").to_units() == 3000000);
        CHECK(ParsedQuantity::try_from("128Mi").to_units() == 128ll << 20);
        CHECK(ParsedQuantity::try_from("1Gi").to_units() == 1ll << 30);  // true Kubernetes semantics (documented divergence)
        CHECK(ParsedQuantity::try_from("1.5").to_milli() == 1500);
        CHECK(ParsedQuantity::try_from("1e3").to_units() == 1000);
        CHECK(ParsedQuantity::try_from("-250m").to_milli() == -250);
        CHECK_THROWS(ParsedQuantity::try_from(""));
        CHECK_THROWS(ParsedQuantity::try_from("abc"));
        CHECK_THROWS(ParsedQuantity::try_from("1Xi"));
        CHECK_THROWS(ParsedQuantity::try_from("100n").to_milli());  // finer than a milli-core: outside the exact domain
    });
    run("quantity: += -= <= (src/util.rs:33-34,65; src/predicates.rs:42)", [] {
        ParsedQuantity a = ParsedQuantity::try_from("250m"), b = ParsedQuantity::try_from("250m");
        a += b;
        CHECK(a.to_milli() == 500);
        a -= ParsedQuantity::try_from("1");
        CHECK(a.to_milli() == -500);
        CHECK(a <= ParsedQuantity::try_from("0"));
        CHECK(!(ParsedQuantity::try_from("1") <= ParsedQuantity::try_from("999m")));
        CHECK(ParsedQuantity::try_from("1") <= ParsedQuantity::try_from("1000m"));
    });
    run("total_pod_resources: containers only, requests only (D-R6, D-R7; src/util.rs:54-75)", [] {
        corev1::Pod p = pod_with("p", {container("250m", "64Mi"), container("250m", nullptr)});
        corev1::Container bare;
        bare.name = "no-resources";
        p.spec->containers.push_back(bare);
        p.spec->init_containers.push_back(container("64", "1Ti"));  // ignored
        PodResources r = total_pod_resources(p);
        CHECK(r.cpu.to_milli() == 500);
        CHECK(r.memory.to_units() == 64ll << 20);
        corev1::Pod nospec;
        PodResources z = total_pod_resources(nospec);
        CHECK(z.cpu.to_milli() == 0 && z.memory.to_units() == 0);
        corev1::Pod bad = pod_with("bad", {container("lots", nullptr)});
        CHECK_THROWS(total_pod_resources(bad));  // the reference panics: "invalid pod spec: cpu request"
    });
    run("is_pod_bound / full_name (src/util.rs:38-52)", [] {
        CHECK(!is_pod_bound(test_pod()));
        CHECK(!is_pod_bound(test_pod("a", "b")));
        CHECK(is_pod_bound(pod_with("x", {}, "node1")));
        CHECK(full_name(test_pod().metadata) == "test/pod1");
        CHECK(full_name(test_node().metadata) == "node1");
    });
    run("InvalidNodeReason / ReconcileError text (src/predicates.rs:14-18, src/error.rs:5-15)", [] {
        CHECK(std::string(predicates::debug_name(InvalidNodeReason::NotEnoughResources)) == "NotEnoughResources");
        CHECK(std::string(predicates::debug_name(InvalidNodeReason::NodeSelectorMismatch)) == "NodeSelectorMismatch");
        CHECK((int)InvalidNodeReason::NotEnoughResources == 0 && (int)InvalidNodeReason::NodeSelectorMismatch == 1);
        CHECK(std::string(error_text(ReconcileError::NoNodeFound)) == "no-node-found");
        CHECK(std::string(error_text(ReconcileError::CreateBindingFailed)) == "create-binding-failed");
        CHECK(std::string(error_text(ReconcileError::CreateBindingObjectFailed)) == "create-binding-object-failed");
        CHECK(ATTEMPTS == 5);
    });
    run("encoder: canonical order, available = allocatable - LIST, label interning, taints", [] {
        std::vector<corev1::Node> nodes = {node_with("node-b", "4", "8589934592"), node_with("node-a", "2", "4294967296"),
                                           node_with("node-c", nullptr, nullptr)};
        nodes[0].metadata.labels = corev1::StringMap{{"zone", "z1"}, {"disk", ""}};
        nodes[1].metadata.labels = corev1::StringMap{{"zone", "z2"}};
        corev1::NodeSpec ns;
        ns.taints = std::vector<corev1::Taint>{{"dedicated", std::string("gpu"), "NoSchedule"}, {"soft", std::nullopt, "PreferNoSchedule"}};
        nodes[0].spec = ns;
        StaticPodLister lister;
        lister.pods = {pod_with("r1", {container("500m", "1073741824")}, "node-a"), pod_with("r2", {container("3", "0")}, "node-a"),
                       pod_with("done", {container("1", "1")}, "node-b")};
        lister.pods[2].status = corev1::PodStatus{std::string("Succeeded")};  // D-R8: still subtracted
        Snapshot snap(Snapshot::kEncodeOnly);
        snap.rebuild(nodes, &lister);
        const NodeColumns &c = snap.columns();
        CHECK(c.n == 3);
        CHECK(c.names[0] == "node-a" && c.names[1] == "node-b" && c.names[2] == "node-c");
        CHECK(c.avail_cpu_milli[0] == 2000 - 500 - 3000);  // over-committed: negative (D-R4)
        CHECK(c.avail_mem_bytes[0] == 4294967296ll - 1073741824ll);
        CHECK(c.avail_cpu_milli[1] == 3000 && c.avail_mem_bytes[1] == 8589934592ll - 1);
        CHECK(c.avail_cpu_milli[2] == 0 && c.avail_mem_bytes[2] == 0);  // no status: 0/0 (D-R5)
        CHECK(lister.list_calls == 3);                                   // one LIST per node
        CHECK(snap.store_index(0) == 1 && snap.canonical_index(0) == 1 && snap.index_of("node-c") == 2 && snap.index_of("nope") == -1);
        CHECK(c.taints[1] == 0ull && snap.has_taints() && !snap.taints_enabled());  // extension E2 is opt-in: nothing interned yet
        snap.enable_taints();
        CHECK(snap.columns().taints[1] == 1ull && snap.columns().taints[0] == 0ull);  // PreferNoSchedule never filters
        // label columns appear when a pod asks for the key
        corev1::Pod p1 = test_pod("zone", "z1"), p2 = test_pod("disk", ""), p3 = test_pod("zone", "nowhere"), p4 = test_pod();
        corev1::Toleration tol;
        tol.key = std::string("dedicated");
        tol.operator_ = std::string("Exists");
        p4.spec = corev1::PodSpec{};
        p4.spec->tolerations = std::vector<corev1::Toleration>{tol};
        PodColumns pc = snap.encode_pods({&p1, &p2, &p3, &p4});
        CHECK(pc.p == 4 && pc.n_keys == 2);
        const NodeColumns &c2 = snap.columns();
        const uint32_t kz = (uint32_t)(std::find(c2.keys.begin(), c2.keys.end(), "zone") - c2.keys.begin());
        const uint32_t kd = (uint32_t)(std::find(c2.keys.begin(), c2.keys.end(), "disk") - c2.keys.begin());
        CHECK(kz < 2 && kd < 2);
        CHECK(c2.label_val_ids[kz * 3 + 1] != 0 && c2.label_val_ids[kz * 3 + 0] != 0 && c2.label_val_ids[kz * 3 + 2] == 0);
        CHECK(c2.label_val_ids[kd * 3 + 1] != 0);  // "" is a value, not "absent" (D-S9)
        CHECK(pc.sel_val_ids[kz * 4 + 0] == c2.label_val_ids[kz * 3 + 1]);
        CHECK(pc.sel_val_ids[kd * 4 + 1] == c2.label_val_ids[kd * 3 + 1]);
        CHECK(pc.sel_val_ids[kz * 4 + 2] == KSCHED_SEL_NEVER);
        CHECK(pc.sel_val_ids[kz * 4 + 3] == 0 && pc.sel_val_ids[kd * 4 + 3] == 0);
        CHECK(pc.tolerations[3] == 1ull && pc.tolerations[0] == 0ull);
        // more than 64 distinct taints (e.g. per-node dedicated=<name>): the parity path (no taint predicate in the reference) is
        // unaffected; only asking for the extension fails, and it leaves the snapshot usable
        {
            std::vector<corev1::Node> many;
            for (int i = 0; i < 70; ++i) {
                many.push_back(node_with("t" + std::to_string(100 + i), "1", "1"));
                corev1::NodeSpec sp;
                sp.taints = std::vector<corev1::Taint>{{"dedicated", std::string("n") + std::to_string(i), "NoSchedule"}};
                many.back().spec = sp;
            }
            Snapshot s3(Snapshot::kEncodeOnly);
            s3.rebuild(many, nullptr);  // does not throw
            CHECK(s3.has_taints());
            corev1::Pod q = test_pod();
            CHECK(s3.encode_pods({&q}).p == 1);
            CHECK_THROWS(s3.enable_taints());
            CHECK(!s3.taints_enabled());
            CHECK(s3.encode_pods({&q}).tolerations[0] == 0ull);
        }
        // label columns are a per-batch working set: a scheduler that has seen more than KSCHED_MAX_KEYS keys over its lifetime
        // keeps encoding (unused columns are evicted); only ONE batch using more than the limit is refused here (the batched
        // entry point splits such a batch)
        {
            Snapshot s4(Snapshot::kEncodeOnly);
            s4.rebuild(nodes, nullptr);
            for (int k = 0; k < 40; ++k) {
                corev1::Pod q = test_pod(("key" + std::to_string(k)).c_str(), "v");
                PodColumns one = s4.encode_pods({&q});
                CHECK(one.n_keys <= KSCHED_MAX_KEYS && one.n_keys >= 1);
            }
            CHECK(s4.columns().keys.size() <= KSCHED_MAX_KEYS);
            corev1::Pod z = test_pod("zone", "z1");
            PodColumns pz = s4.encode_pods({&z});
            const auto &ks = s4.columns().keys;
            const uint32_t col = (uint32_t)(std::find(ks.begin(), ks.end(), "zone") - ks.begin());
            CHECK(col < pz.n_keys && pz.sel_val_ids[col] != 0 && pz.sel_val_ids[col] != KSCHED_SEL_NEVER);
            std::vector<corev1::Pod> wide(40);
            std::vector<const corev1::Pod *> wp;
            for (int k = 0; k < 40; ++k) {
                wide[k] = test_pod(("wide" + std::to_string(k)).c_str(), "v");
                wp.push_back(&wide[k]);
            }
            CHECK_THROWS(s4.encode_pods(wp));
        }
        // allocatable present but lacking memory: the reference panics (src/predicates.rs:29-31)
        std::vector<corev1::Node> badn = {node_with("x", "1", nullptr)};
        Snapshot s2(Snapshot::kEncodeOnly);
        CHECK_THROWS(s2.rebuild(badn, nullptr));
        CHECK_THROWS(snap.device());  // encode-only snapshots cannot evaluate
    });
    run("snapshot builder: apply_bound_pod / apply_deleted_pod == re-LIST (SURVEY.md 8f n1; src/predicates.rs:34-38)", [] {
        std::vector<corev1::Node> nodes = {node_with("node-b", "4", "8589934592"), node_with("node-a", "2", "4294967296")};
        StaticPodLister lister;
        lister.pods = {pod_with("r1", {container("500m", "1073741824")}, "node-a")};
        Snapshot inc(Snapshot::kEncodeOnly);
        inc.rebuild(nodes, &lister);
        const uint64_t g0 = inc.generation();
        // a pod lands on node-b, another on node-a, then r1 goes away: patch the columns event by event
        corev1::Pod n1 = pod_with("n1", {container("1500m", "1000"), container("250m", nullptr)}, "node-b");
        corev1::Pod n2 = pod_with("n2", {container("3", "4294967296")}, "node-a");  // over-commits node-a: goes negative (D-R4)
        CHECK(inc.apply_bound_pod(n1));
        CHECK(inc.apply_pod_events({{&n2, true}, {&lister.pods[0], false}}) == 2);
        CHECK(inc.generation() > g0);
        // the same cluster state, re-LISTed from scratch as the reference does per evaluation
        StaticPodLister after;
        after.pods = {n1, n2};
        Snapshot full(Snapshot::kEncodeOnly);
        full.rebuild(nodes, &after);
        CHECK(inc.columns().avail_cpu_milli == full.columns().avail_cpu_milli);
        CHECK(inc.columns().avail_mem_bytes == full.columns().avail_mem_bytes);
        CHECK(inc.columns().avail_cpu_milli[0] == 2000 - 3000 && inc.columns().avail_mem_bytes[0] == 0);
        CHECK(inc.columns().avail_cpu_milli[1] == 4000 - 1750 && inc.columns().avail_mem_bytes[1] == 8589934592ll - 1000);
        // not applicable: unbound pod, unknown node -> false, nothing changes
        corev1::Pod unbound = pod_with("u", {container("1", "1")});
        corev1::Pod elsewhere = pod_with("e", {container("1", "1")}, "node-zz");
        CHECK(!inc.apply_bound_pod(unbound) && !inc.apply_deleted_pod(elsewhere));
        CHECK(inc.columns().avail_cpu_milli == full.columns().avail_cpu_milli);
        // delete then bind again is the identity
        CHECK(inc.apply_deleted_pod(n1) && inc.apply_bound_pod(n1));
        CHECK(inc.columns().avail_cpu_milli == full.columns().avail_cpu_milli && inc.columns().avail_mem_bytes == full.columns().avail_mem_bytes);
        // unparsable requests are an error, as in rebuild (the reference panics, src/util.rs:65,68)
        corev1::Pod bad = pod_with("bad", {container("lots", nullptr)}, "node-a");
        CHECK_THROWS(inc.apply_bound_pod(bad));
    });
    run("snapshot builder: observe_pods is idempotent, follows moves, and changes nothing when it throws (SURVEY.md 8f n1)", [] {
        std::vector<corev1::Node> nodes = {node_with("a", "8", "16Gi"), node_with("b", "8", "16Gi")};
        auto lister = std::make_shared<StaticPodLister>();
        lister->pods = {pod_with("old", {container("1", "1Gi")}, "a")};
        Snapshot snap(Snapshot::kEncodeOnly);
        snap.rebuild(nodes, lister.get());
        using E = Snapshot::PodEvent;
        CHECK(snap.counted_pods() == 1 && snap.columns().avail_cpu_milli[0] == 7000);
        const corev1::Pod p1 = pod_with("p1", {container("2", "2Gi")}, "a");
        CHECK(snap.observe_pod(E::Applied, p1));
        CHECK(!snap.observe_pod(E::Applied, p1));  // the watch echoes the binding: nothing changes
        CHECK(!snap.observe_pod(E::Applied, p1));
        CHECK(snap.columns().avail_cpu_milli[0] == 5000 && snap.columns().avail_mem_bytes[0] == (16ll - 1 - 2) << 30);
        CHECK(!snap.observe_pod(E::Applied, pod_with("old", {container("1", "1Gi")}, "a")));  // a MODIFIED event of a LISTed pod: already counted
        const corev1::Pod moved = pod_with("p1", {container("2", "2Gi")}, "b");  // (a pod cannot move in Kubernetes; a delete + re-create under one name can look like it)
        CHECK(snap.observe_pod(E::Applied, moved));
        CHECK(snap.columns().avail_cpu_milli[0] == 7000 && snap.columns().avail_cpu_milli[1] == 6000);
        CHECK(snap.observe_pod(E::Deleted, pod_with("p1", {}, nullptr)));  // the Deleted event's object need not carry the spec that was counted
        CHECK(!snap.observe_pod(E::Deleted, moved));
        CHECK(snap.columns().avail_cpu_milli[1] == 8000 && snap.counted_pods() == 1);
        CHECK(!snap.observe_pod(E::Applied, pod_with("pending", {container("1", "1Gi")})));          // no nodeName: nothing to count
        CHECK(!snap.observe_pod(E::Applied, pod_with("away", {container("1", "1Gi")}, "not-here")));  // a node outside the snapshot
        // strong guarantee: the second event cannot be encoded -> the first one is not applied either
        const corev1::Pod ok = pod_with("ok", {container("1", "1Gi")}, "b"), bad = pod_with("bad", {container("lots", nullptr)}, "b");
        const auto cpu_before = snap.columns().avail_cpu_milli;
        CHECK_THROWS(snap.observe_pods({{E::Applied, &ok}, {E::Applied, &bad}}));
        CHECK(snap.columns().avail_cpu_milli == cpu_before && snap.counted_pods() == 1);
        // finer than a milli-core: the reference compares decimals and schedules it like any other pod (src/util.rs:64-69) -- the column's unit follows
        const corev1::Pod fine = pod_with("fine", {container("100n", nullptr)}, "b");
        CHECK(snap.cpu_unit_nanos() == 1000000 && snap.mem_unit_nanos() == 1000000000);
        CHECK(snap.observe_pod(E::Applied, fine));
        CHECK(snap.cpu_unit_nanos() == 1 && snap.mem_unit_nanos() == 1000000000);  // nano-cores now; memory still in bytes
        CHECK(snap.columns().avail_cpu_milli[0] == 7000ll * 1000000 && snap.columns().avail_cpu_milli[1] == 8000ll * 1000000 - 100 && snap.counted_pods() == 2);
        {   // a request is encoded as ceil(request / unit): exact against a column of whole units
            const corev1::Pod one = pod_with("one", {container("1", "1")}), tiny = pod_with("tiny", {container("1500n", "100m")});
            const PodColumns pc = snap.encode_pods({&one, &tiny});
            CHECK(pc.req_cpu_milli[0] == 1000000000ll && pc.req_cpu_milli[1] == 1500);
            CHECK(pc.req_mem_bytes[0] == 1 && pc.req_mem_bytes[1] == 1);  // 0.1 byte -> 1 byte: 0.1 <= available <=> 1 <= available for whole-byte values
        }
        CHECK(snap.observe_pod(E::Deleted, fine));  // whole milli-cores again; the unit stays until the next rebuild (the comparison is exact in any unit)
        CHECK(snap.cpu_unit_nanos() == 1 && snap.columns().avail_cpu_milli[1] == 8000ll * 1000000 && snap.counted_pods() == 1);
        // two halves of a milli-core in one call
        const corev1::Pod h1 = pod_with("h1", {container("500u", nullptr)}, "b"), h2 = pod_with("h2", {container("500u", nullptr)}, "b");
        CHECK(snap.observe_pods({{E::Applied, &h1}, {E::Applied, &h2}}) == 2);
        CHECK(snap.columns().avail_cpu_milli[1] == 7999ll * 1000000 && snap.counted_pods() == 3);
        // what no unit can hold is still refused, with nothing changed: 9.3e9 cores (more than int64 nano-cores) on a node that counts a "100n" pod
        const auto before = snap.columns().avail_cpu_milli;
        const corev1::Pod huge = pod_with("huge", {container("9300000000", nullptr)}, "b");
        CHECK(snap.observe_pod(E::Applied, fine));
        CHECK_THROWS(snap.observe_pod(E::Applied, huge));
        CHECK(snap.counted_pods() == 4 && snap.columns().avail_cpu_milli[0] == before[0]);
    });
    run("quantity domain: a cluster with sub-milli CPU and sub-byte memory is scheduled like the reference schedules it (VERDICT r3 item 6)", [] {
        // node a: 2 cores - a bound pod of 100u = 1.9999 cores; memory 1000 bytes - 100m (0.1 byte) = 999.9 bytes
        std::vector<corev1::Node> nodes = {node_with("a", "2", "1000"), node_with("b", "1500n", "1")};
        std::vector<corev1::Pod> bound = {pod_with("load", {container("100u", "100m")}, "a")};
        Snapshot snap(Snapshot::kEncodeOnly);
        StaticPodLister lister;
        lister.pods = bound;
        snap.rebuild(nodes, &lister);
        CHECK(snap.cpu_unit_nanos() == 1 && snap.mem_unit_nanos() == 1000000);  // nano-cores (1500n), milli-bytes (100m)
        CHECK(snap.columns().avail_cpu_milli[0] == 2000000000ll - 100000 && snap.columns().avail_mem_bytes[0] == 1000000ll - 100);
        CHECK(snap.columns().avail_cpu_milli[1] == 1500 && snap.columns().avail_mem_bytes[1] == 1000);
        // exact-fit boundaries: request == available fits, one nano-core / milli-byte more does not
        const corev1::Pod exact = pod_with("exact", {container("1999900u", "999900m")}), over_cpu = pod_with("oc", {container("1999900001n", "1")}),
                          over_mem = pod_with("om", {container("1", "999901m")}), small = pod_with("s", {container("1500n", "1")});
        const PodColumns pc = snap.encode_pods({&exact, &over_cpu, &over_mem, &small});
        auto fits = [&](uint32_t pod, uint32_t node) { return pc.req_cpu_milli[pod] <= snap.columns().avail_cpu_milli[node] && pc.req_mem_bytes[pod] <= snap.columns().avail_mem_bytes[node]; };
        CHECK(fits(0, 0) && !fits(1, 0) && !fits(2, 0) && fits(3, 0));
        CHECK(!fits(0, 1) && fits(3, 1));  // node b holds exactly 1500n / 1 byte
        // a cluster without such values keeps milli-cores / bytes (what every earlier test and the fixtures assume)
        Snapshot plain(Snapshot::kEncodeOnly);
        plain.rebuild({node_with("a", "2", "1Gi")}, nullptr);
        CHECK(plain.cpu_unit_nanos() == 1000000 && plain.mem_unit_nanos() == 1000000000);
    });
    run("snapshot builder: observe_bound == observe_pods on copies that carry the node name, also for a batch large enough for the thread fan-out", [] {
        std::vector<corev1::Node> nodes;
        for (int i = 0; i < 40; ++i) nodes.push_back(node_with("n" + std::to_string(100 + i), "64", "256Gi"));
        Snapshot a(Snapshot::kEncodeOnly), b(Snapshot::kEncodeOnly);
        a.rebuild(nodes, nullptr);
        b.rebuild(nodes, nullptr);
        const size_t P = 6000;  // (>= 4096: the per-event string work runs on several threads)
        std::vector<corev1::Pod> pods, copies;
        std::vector<std::string> names;
        for (size_t i = 0; i < P; ++i) {
            const std::string cpu = std::to_string(1 + i % 7) + "m", mem = std::to_string(1 + i % 5) + "Mi";
            pods.push_back(pod_with("p" + std::to_string(i), {container(cpu.c_str(), mem.c_str())}, nullptr));
            names.push_back(i % 97 == 0 ? std::string("not-in-the-snapshot") : "n" + std::to_string(100 + i % 40));
            copies.push_back(pods.back());
            copies.back().spec->node_name = names.back();
        }
        std::vector<std::pair<const corev1::Pod *, const std::string *>> bound;
        std::vector<std::pair<Snapshot::PodEvent, const corev1::Pod *>> events;
        for (size_t i = 0; i < P; ++i) {
            bound.emplace_back(&pods[i], &names[i]);
            events.emplace_back(Snapshot::PodEvent::Applied, &copies[i]);
        }
        const size_t ca = a.observe_bound(bound), cb = b.observe_pods(events);
        CHECK(ca == cb && ca == P - (P + 96) / 97);
        CHECK(a.columns().avail_cpu_milli == b.columns().avail_cpu_milli && a.columns().avail_mem_bytes == b.columns().avail_mem_bytes);
        CHECK(a.counted_pods() == b.counted_pods());
        CHECK(a.observe_bound(bound) == 0);  // the same bindings again (the watch's echo): nothing changes
        // strong guarantee with the fan-out: one unparsable pod in the middle -> nothing is applied
        pods[P / 2] = pod_with("late", {container("lots", nullptr)}, nullptr);
        std::vector<std::pair<const corev1::Pod *, const std::string *>> again;
        std::vector<std::string> other(P, "n101");
        for (size_t i = 0; i < P; ++i) again.emplace_back(&pods[i], &other[i]);
        const auto before = a.columns().avail_cpu_milli;
        CHECK_THROWS(a.observe_bound(again));
        CHECK(a.columns().avail_cpu_milli == before && a.counted_pods() == b.counted_pods());
    });
    run("snapshot builder: stage_bound + commit_staged == observe_bound (the two halves reconcile_batch runs beside and behind its POSTs)", [] {
        std::vector<corev1::Node> nodes;
        for (int i = 0; i < 40; ++i) nodes.push_back(node_with("n" + std::to_string(100 + i), "64", "256Gi"));
        Snapshot a(Snapshot::kEncodeOnly), b(Snapshot::kEncodeOnly);
        a.rebuild(nodes, nullptr);
        b.rebuild(nodes, nullptr);
        const size_t P = 5000;
        std::vector<corev1::Pod> pods;
        std::vector<std::string> names;
        for (size_t i = 0; i < P; ++i) {
            const std::string cpu = std::to_string(1 + i % 7) + "m", mem = std::to_string(1 + i % 5) + "Mi";
            pods.push_back(pod_with("p" + std::to_string(i % 4900), {container(cpu.c_str(), mem.c_str())}, nullptr));  // (the last hundred repeat earlier keys: moves)
            names.push_back("n" + std::to_string(100 + i % 40));
        }
        std::vector<const corev1::Pod *> pp;
        for (const auto &p : pods) pp.push_back(&p);
        const PodColumns pc = a.encode_pods(pp);  // the exact request sums, as the evaluation leaves them
        std::vector<Snapshot::Bound> bound;
        std::vector<std::pair<const corev1::Pod *, const std::string *>> by_name;
        for (size_t i = 0; i < P; ++i) {
            bound.push_back({&pods[i], (uint32_t)a.index_of(names[i]), pc.req_cpu_nanos[i], pc.req_mem_nanos[i]});
            by_name.emplace_back(&pods[i], &names[i]);
        }
        const auto before_cpu = a.columns().avail_cpu_milli;
        const uint64_t gen = a.generation();
        auto staged = a.stage_bound(bound);
        CHECK(a.columns().avail_cpu_milli == before_cpu && a.generation() == gen && a.counted_pods() == 0);  // staging changes nothing
        const size_t ca = a.commit_staged(*staged), cb = b.observe_bound(by_name);
        CHECK(ca == cb && ca == P);
        CHECK(a.columns().avail_cpu_milli == b.columns().avail_cpu_milli && a.columns().avail_mem_bytes == b.columns().avail_mem_bytes);
        CHECK(a.counted_pods() == b.counted_pods() && a.counted_pods() == 4900);
        CHECK_THROWS(a.commit_staged(*staged));  // an update is committed once
        // ... and the same events again leave `available` where it is: the 4 800 single bindings change nothing, the hundred keys bound twice
        // move to their first node and back (two changes each)
        CHECK(a.observe_bound(bound) == 200 && a.columns().avail_cpu_milli == b.columns().avail_cpu_milli && a.columns().avail_mem_bytes == b.columns().avail_mem_bytes);
        // an update staged against a snapshot that has changed since is refused, and changes nothing
        auto stale = a.stage_bound({{&pods[0], 3u, pc.req_cpu_nanos[0], pc.req_mem_nanos[0]}});
        corev1::Pod late = pod_with("late", {container("5m", "1Mi")}, nullptr);
        late.spec->node_name = "n101";
        CHECK(a.observe_pod(Snapshot::PodEvent::Applied, late));
        const auto cpu_now = a.columns().avail_cpu_milli;
        CHECK_THROWS(a.commit_staged(*stale));
        CHECK(a.columns().avail_cpu_milli == cpu_now);
        // a node index outside the snapshot = "no node": the pod is not counted (and an earlier count of it is dropped)
        CHECK(a.observe_bound(std::vector<Snapshot::Bound>{{&pods[1], 4000000u, pc.req_cpu_nanos[1], pc.req_mem_nanos[1]}}) == 1);
        CHECK(a.counted_pods() == 4900);  // p1 out, "late" in
    });
    run("worker pool: parts cover the range once, the lowest part's exception comes back, a nested region runs on the calling thread", [] {
        WorkerPool &pool = WorkerPool::instance();
        std::vector<uint8_t> hit(100000, 0);
        std::atomic<int> parts_seen{0};
        pool.run(hit.size(), 7, [&](size_t lo, size_t hi, uint32_t) {
            ++parts_seen;
            for (size_t i = lo; i < hi; ++i) ++hit[i];
        });
        CHECK(parts_seen == 7 && std::count(hit.begin(), hit.end(), (uint8_t)1) == (long)hit.size());
        bool threw = false;
        try {
            pool.run(1000, 5, [&](size_t, size_t, uint32_t part) {
                if (part == 3 || part == 1) throw EncodeError("part " + std::to_string(part));
            });
        } catch (const EncodeError &e) {
            threw = std::string(e.what()) == "part 1";
        }
        CHECK(threw);
        std::atomic<int> inner{0};
        pool.run(8, 4, [&](size_t lo, size_t hi, uint32_t) {
            for (size_t i = lo; i < hi; ++i) pool.run(10, 3, [&](size_t a_, size_t b_, uint32_t) { inner += (int)(b_ - a_); });  // (no deadlock: runs inline)
        });
        CHECK(inner == 80);
        pool.run(0, 4, [&](size_t lo, size_t hi, uint32_t) { CHECK(lo == 0 && hi == 0); });
        CHECK(WorkerPool::parts_for(100, 1024) == 1 && WorkerPool::parts_for(1 << 20, 1024, 4) <= 4);
    });
    run("snapshot builder: a rebuild that throws leaves the snapshot as it was (commit at the end)", [] {
        // 5 nodes with taints enabled; then a rebuild of 2 nodes carrying 65 distinct taints: "more than 64" -- thrown BEFORE anything is
        // committed (it used to come after the columns had been replaced: host n = 2, device still 5 nodes)
        std::vector<corev1::Node> five;
        for (int i = 0; i < 5; ++i) five.push_back(node_with("n" + std::to_string(i), "4", "1000"));
        corev1::NodeSpec one;
        one.taints = std::vector<corev1::Taint>{{"dedicated", std::string("gpu"), "NoSchedule"}};
        five[2].spec = one;
        five[1].metadata.labels = corev1::StringMap{{"zone", "z1"}};
        Snapshot snap(Snapshot::kEncodeOnly);
        snap.rebuild(five, nullptr);
        snap.enable_taints();
        corev1::Pod asks = test_pod("zone", "z1");
        snap.encode_pods({&asks});
        const NodeColumns before = snap.columns();
        const uint64_t gen = snap.generation();
        std::vector<corev1::Node> two = {node_with("a", "1", "1"), node_with("b", "1", "1")};
        corev1::NodeSpec many;
        many.taints = std::vector<corev1::Taint>{};
        for (int t = 0; t < 65; ++t) many.taints->push_back({"k" + std::to_string(t), std::nullopt, "NoSchedule"});
        two[1].spec = many;
        CHECK_THROWS(snap.rebuild(two, nullptr));
        const NodeColumns &after = snap.columns();
        CHECK(after.n == 5 && after.names == before.names && after.avail_cpu_milli == before.avail_cpu_milli && after.taints == before.taints);
        CHECK(after.label_val_ids == before.label_val_ids && after.keys == before.keys && snap.generation() == gen && snap.taints_enabled());
        CHECK(snap.index_of("n3") == 3 && snap.index_of("a") == -1 && snap.store_index(4) == 4);
        const PodColumns pc = snap.encode_pods({&asks});  // still encodes against the five nodes' dictionaries
        CHECK(pc.n_keys == 1 && pc.sel_val_ids[0] == 1u);
        // a node that cannot be encoded: the same
        std::vector<corev1::Node> bad = {node_with("x", "lots", "1")};
        CHECK_THROWS(snap.rebuild(bad, nullptr));
        CHECK(snap.columns().n == 5 && snap.index_of("n0") == 0);
        // enable_taints that fails leaves the extension off (and can be retried after the cluster changed)
        Snapshot s2(Snapshot::kEncodeOnly);
        s2.rebuild(two, nullptr);
        CHECK_THROWS(s2.enable_taints());
        CHECK(!s2.taints_enabled() && s2.taint_ids().empty());
    });
    run("toleration_matches (extension E2)", [] {
        TaintId t{"k", "v", "NoSchedule"};
        corev1::Toleration a;
        a.key = std::string("k");
        a.value = std::string("v");
        CHECK(toleration_matches(a, t));  // default operator Equal
        a.value = std::string("w");
        CHECK(!toleration_matches(a, t));
        a.operator_ = std::string("Exists");
        CHECK(toleration_matches(a, t));
        a.effect = std::string("NoExecute");
        CHECK(!toleration_matches(a, t));
        corev1::Toleration all;
        all.operator_ = std::string("Exists");
        CHECK(toleration_matches(all, t));
    });
    run("post_bindings: the POSTs of a batch overlap (SURVEY.md 8f n4; src/main.rs:94-108, 141-144), outcomes per pod", [] {
        struct SlowSink : BindingSink {  // a POST = 2 ms of "network"; counts how many are in flight at once
            std::mutex mu;
            std::vector<std::pair<std::string, std::string>> posts;
            std::atomic<int> in_flight{0}, max_in_flight{0}, calls{0};
            bool create_pod_binding(const std::string &pod_name, const std::string &ns, const Binding &b) override {
                const int now = ++in_flight;
                int seen = max_in_flight.load();
                while (now > seen && !max_in_flight.compare_exchange_weak(seen, now)) {
                }
                ++calls;
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
                --in_flight;
                if (pod_name == "p7" || pod_name == "p21") return false;         // the API server refused these (src/main.rs:105-108)
                if (pod_name == "p33") throw std::runtime_error("connection reset");  // a sink that throws = a failed POST
                std::lock_guard<std::mutex> lk(mu);
                posts.push_back({ns + "/" + pod_name, b.target_name});
                return true;
            }
        };
        std::vector<corev1::Node> nodes;
        for (int i = 0; i < 4; ++i) nodes.push_back(node_with("node" + std::to_string(i), "8", "1Gi"));
        std::vector<corev1::Pod> pods;
        for (int i = 0; i < 64; ++i) pods.push_back(pod_with("p" + std::to_string(i), {container("100m", "1Mi")}));
        pods[40].metadata.namespace_.reset();  // pod.namespace().unwrap() panics in the reference (src/main.rs:80): reported, no POST
        std::vector<const corev1::Pod *> pp;
        std::vector<const corev1::Node *> chosen;
        for (int i = 0; i < 64; ++i) {
            pp.push_back(&pods[i]);
            chosen.push_back(i % 9 == 8 ? nullptr : &nodes[i % 4]);  // every ninth pod found no node (src/main.rs:116-118)
        }
        auto check = [&](const std::vector<ReconcileOutcome> &out, SlowSink &sink) {
            CHECK(out.size() == 64);
            int ok = 0;
            for (int i = 0; i < 64; ++i) {
                const ReconcileOutcome &o = out[i];
                if (i % 9 == 8) CHECK(!o.ok && o.error == ReconcileError::NoNodeFound && o.action == Action::RequeueAfter5Min && !o.bound_to);
                else if (i == 40) CHECK(!o.ok && o.error == ReconcileError::CreateBindingObjectFailed);
                else if (i == 7 || i == 21 || i == 33) CHECK(!o.ok && o.error == ReconcileError::CreateBindingFailed && o.action == Action::RequeueAfter5Min);
                else {
                    CHECK(o.ok && o.action == Action::AwaitChange && o.bound_to && *o.bound_to == "node" + std::to_string(i % 4));
                    ++ok;
                }
            }
            CHECK((int)sink.posts.size() == ok);
            CHECK(sink.calls == 64 - 7 - 1);  // no POST for the seven pods without a node nor for the one without a namespace
            std::sort(sink.posts.begin(), sink.posts.end());
            CHECK(std::adjacent_find(sink.posts.begin(), sink.posts.end()) == sink.posts.end());  // every binding POSTed once
        };
        SlowSink serial, pooled;
        const auto t0 = std::chrono::steady_clock::now();
        const auto a = post_bindings(pp, chosen, serial, 1);
        const auto t1 = std::chrono::steady_clock::now();
        const auto b = post_bindings(pp, chosen, pooled, 16);
        const auto t2 = std::chrono::steady_clock::now();
        check(a, serial);
        check(b, pooled);
        CHECK(serial.max_in_flight == 1);
        CHECK(pooled.max_in_flight > 4 && pooled.max_in_flight <= 16);
        const double ms_serial = std::chrono::duration<double, std::milli>(t1 - t0).count(), ms_pooled = std::chrono::duration<double, std::milli>(t2 - t1).count();
        CHECK(ms_serial >= 56 * 2.0);          // 56 POSTs one after the other
        CHECK(ms_pooled < ms_serial / 4.0);    // 16 in flight: a quarter of that at the very most
        for (int i = 0; i < 64; ++i) CHECK(a[i].ok == b[i].ok && a[i].error == b[i].error && a[i].bound_to == b[i].bound_to);
        // fewer pods than workers, and an empty batch
        SlowSink few;
        const auto c = post_bindings({pp[0], pp[1]}, {chosen[0], chosen[1]}, few, 16);
        CHECK(c.size() == 2 && c[0].ok && c[1].ok && few.calls == 2);
        CHECK(post_bindings({}, {}, few, 8).empty());
    });
    run("PodBatcher: ready_chunks semantics, one queued entry per pod, close drains (SURVEY.md 8f n2; src/main.rs:141-144)", [] {
        auto mk = [](const std::string &name, const char *cpu) { return std::make_shared<const corev1::Pod>(pod_with(name, {container(cpu, "1Mi")})); };
        PodBatcher b(4);
        CHECK(b.try_next_batch().empty());
        for (int i = 0; i < 10; ++i) CHECK(b.push(mk("p" + std::to_string(i), "100m")));
        CHECK(b.pending() == 10);
        CHECK(b.push(mk("p2", "900m")));  // p2 again while queued: replaces the object, keeps its place
        CHECK(b.push(mk("p9", "700m")));
        CHECK(b.pending() == 10 && b.coalesced() == 2);
        auto one = b.next_batch();
        CHECK(one.size() == 4 && *one[0]->metadata.name == "p0" && *one[2]->metadata.name == "p2");
        CHECK(total_pod_resources(*one[2]).cpu.to_milli() == 900);  // the latest object
        CHECK(b.push(mk("p2", "50m")));  // p2 is out of the queue now: this is a new request for it
        CHECK(b.pending() == 7 && b.coalesced() == 2);
        auto two = b.next_batch(), three = b.next_batch();
        CHECK(two.size() == 4 && three.size() == 3);
        CHECK(*three[1]->metadata.name == "p9" && total_pod_resources(*three[1]).cpu.to_milli() == 700);
        CHECK(*three[2]->metadata.name == "p2" && total_pod_resources(*three[2]).cpu.to_milli() == 50);
        b.close();
        CHECK(!b.push(mk("late", "1")));
        CHECK(b.next_batch().empty() && b.closed());
    });
    run("run_batches: producers and the batch loop concurrently, every pod reconciled exactly once", [] {
        PodBatcher b(64);
        constexpr int kProducers = 4, kPerProducer = 500;
        std::vector<std::thread> producers;
        for (int t = 0; t < kProducers; ++t)
            producers.emplace_back([&b, t] {
                for (int i = 0; i < kPerProducer; ++i) {
                    b.push(std::make_shared<const corev1::Pod>(pod_with("t" + std::to_string(t) + "-" + std::to_string(i), {container("100m", "1Mi")})));
                    if (i % 97 == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));  // bursts and gaps
                }
            });
        std::map<std::string, int> seen;
        size_t reconcile_calls = 0;
        std::thread closer([&] {
            for (auto &t : producers) t.join();
            b.close();
        });
        const BatchLoopStats st = run_batches(
            b,
            [&](const std::vector<const corev1::Pod *> &pods) {
                ++reconcile_calls;
                std::vector<ReconcileOutcome> out(pods.size());
                for (size_t i = 0; i < pods.size(); ++i)
                    if (pods[i]->metadata.name->back() == '7') {  // "no node found" for some
                        out[i].ok = false;
                        out[i].error = ReconcileError::NoNodeFound;
                        out[i].action = Action::RequeueAfter5Min;
                    }
                return out;
            },
            [&](const PodBatcher::PodPtr &pod, const ReconcileOutcome &o) {
                ++seen[*pod->metadata.name];
                CHECK(o.ok == (pod->metadata.name->back() != '7'));
            });
        closer.join();
        CHECK(st.pods == (uint64_t)kProducers * kPerProducer && seen.size() == (size_t)kProducers * kPerProducer);
        bool once = true;
        for (const auto &kv : seen) once = once && kv.second == 1;
        CHECK(once);
        CHECK(st.batches == reconcile_calls && st.largest <= 64 && st.batches >= (uint64_t)kProducers * kPerProducer / 64);
        CHECK(b.pending() == 0);
        // one pod the host cannot encode: its batch is reconciled pod by pod, the offender is reported, nobody else is lost
        {
            PodBatcher d(8);
            for (int i = 0; i < 20; ++i) d.push(std::make_shared<const corev1::Pod>(pod_with(i == 11 ? "poison" : "q" + std::to_string(i), {container("1", "1Mi")})));
            d.close();
            int done_n = 0, failed_n = 0, calls = 0;
            std::string what;
            auto rec = [&](const std::vector<const corev1::Pod *> &pods) {
                ++calls;
                for (const auto *p : pods)
                    if (*p->metadata.name == "poison") throw PodEncodeError("pod test/poison: invalid pod spec: 'lots'");  // what Snapshot::encode_pods throws
                return std::vector<ReconcileOutcome>(pods.size());
            };
            const BatchLoopStats s2 = run_batches(
                d, rec, [&](const PodBatcher::PodPtr &, const ReconcileOutcome &o) { done_n += o.ok ? 1 : 0; },
                [&](const PodBatcher::PodPtr &p, const std::string &w) {
                    ++failed_n;
                    what = *p->metadata.name + ": " + w;
                });
            CHECK(done_n == 19 && failed_n == 1 && what.find("poison: pod test/poison: invalid pod spec") == 0);
            CHECK(s2.batches == 3 && s2.pods == 20 && s2.isolated_batches == 1 && s2.failed_pods == 1);
            CHECK(calls == 3 + 8);  // three batch calls (the second throws), then that batch's eight pods one by one
            // without the callback the exception is the caller's
            PodBatcher e(8);
            e.push(std::make_shared<const corev1::Pod>(pod_with("poison", {container("1", "1Mi")})));
            e.close();
            CHECK_THROWS(run_batches(e, rec, [](const PodBatcher::PodPtr &, const ReconcileOutcome &) {}));
            // Any OTHER exception is not "a pod could not be encoded, nothing was POSTed": it may come from after the batch's POSTs
            // (a device failure while the batch's own bindings are applied to the snapshot).  The loop must NOT reconcile -- and
            // POST -- the batch's pods a second time: the exception is the caller's, and the reconcile function ran once.
            PodBatcher f(8);
            for (int i = 0; i < 5; ++i) f.push(std::make_shared<const corev1::Pod>(pod_with("r" + std::to_string(i), {container("1", "1Mi")})));
            f.close();
            int calls2 = 0, failed2 = 0;
            CHECK_THROWS(run_batches(
                f,
                [&](const std::vector<const corev1::Pod *> &pods) -> std::vector<ReconcileOutcome> {
                    ++calls2;
                    (void)pods;
                    throw EncodeError("ksched_update_nodes: HIP runtime error (after the POSTs)");
                },
                [](const PodBatcher::PodPtr &, const ReconcileOutcome &) {}, [&](const PodBatcher::PodPtr &, const std::string &) { ++failed2; }));
            CHECK(calls2 == 1 && failed2 == 0);
        }
        // a reconcile function that loses a pod is a programming error, reported loudly
        PodBatcher c(8);
        c.push(std::make_shared<const corev1::Pod>(pod_with("x", {container("1", "1Mi")})));
        c.close();
        CHECK_THROWS(run_batches(c, [](const std::vector<const corev1::Pod *> &) { return std::vector<ReconcileOutcome>{}; },
                                 [](const PodBatcher::PodPtr &, const ReconcileOutcome &) {}));
    });
    run("row sharding: ksched_shard_bounds / merge_gathered (SURVEY.md 8e: contiguous rows, ceil(p / n) per device, padded all-gather)", [] {
        for (uint32_t n = 1; n <= 9; ++n)
            for (uint32_t p = 0; p <= 200; ++p) {
                uint32_t next = 0, cpr0 = shard_bounds(p, n, 0).count_per_rank;
                CHECK(cpr0 == (p + n - 1) / n);
                std::vector<int32_t> table((size_t)n * cpr0, -7), want(p), got(p, -9);
                for (uint32_t r = 0; r < n; ++r) {
                    const ShardBounds b = shard_bounds(p, n, r);
                    CHECK(b.lo == next && b.hi >= b.lo && b.hi - b.lo <= cpr0 && b.count_per_rank == cpr0);  // contiguous, in rank order, within the slot
                    CHECK(b.lo == std::min(p, r * cpr0));
                    next = b.hi;
                    for (uint32_t i = b.lo; i < b.hi; ++i) {
                        want[i] = (int32_t)(i * 31u % 1000u) - 1;                      // what rank r's device wrote for its row i - lo
                        table[(size_t)r * cpr0 + (i - b.lo)] = want[i];
                    }
                }
                CHECK(next == p);  // every row owned exactly once
                merge_gathered(table.data(), p, n, got.data());
                CHECK(got == want);
            }
        // the C ABI's own entry point with null outputs, and ranks past the end
        ksched_shard_bounds(10, 4, 3, nullptr, nullptr, nullptr);
        const ShardBounds last = shard_bounds(10, 4, 3), beyond = shard_bounds(2, 4, 3);
        CHECK(last.lo == 9 && last.hi == 10 && last.count_per_rank == 3);
        CHECK(beyond.lo == 2 && beyond.hi == 2);  // an empty shard
        // $KSCHED_DEVICES
        CHECK(devices_from_env(nullptr, 5) == std::vector<int>{5});
        CHECK(devices_from_env("", 2) == std::vector<int>{2});
        CHECK_THROWS(devices_from_env("zero", 0));
        CHECK_THROWS(devices_from_env("0;1", 0));
        CHECK_THROWS(devices_from_env("-1", 0));
        // the vectors the Rust twin's parse_device_ids reads the same way: blanks around an entry are ignored, an empty entry is an error
        CHECK_THROWS(devices_from_env("0,", 0));
        CHECK_THROWS(devices_from_env("0,,1", 0));
        CHECK_THROWS(devices_from_env(",0", 0));
        CHECK_THROWS(devices_from_env("+0", 0));
        CHECK_THROWS(devices_from_env("0 1", 0));
        if (ksched_device_count() >= 1) {
            CHECK(devices_from_env("0", 3) == std::vector<int>{0});
            CHECK(devices_from_env(" 0 ", 3) == std::vector<int>{0});
            CHECK(devices_from_env("all", 3).size() == (size_t)ksched_device_count());
            CHECK_THROWS(devices_from_env("0,0", 0));  // a device listed twice
            CHECK_THROWS(devices_from_env(std::to_string(ksched_device_count()).c_str(), 0));  // a device the process does not see
        } else {
            CHECK_THROWS(devices_from_env("0", 0));
            CHECK_THROWS(devices_from_env("all", 0));
        }
    });
    run("choosers: scripted and SplitMix draws", [] {
        ScriptedChooser s;
        s.script = {3, 3, 7, 1, 0};
        CHECK(*s.choose(10) == 3 && *s.choose(10) == 3 && *s.choose(10) == 7 && *s.choose(10) == 1 && *s.choose(10) == 0);
        CHECK(!s.choose(10).has_value());
        CHECK(!s.choose(0).has_value());
        SplitMixChooser a(42), b(42);
        for (int i = 0; i < 100; ++i) {
            auto x = a.choose(17), y = b.choose(17);
            CHECK(x && y && *x == *y && *x < 17);
        }
        CHECK(!a.choose(0).has_value());
    });
}

// =========================================== GPU-side tests ================================================
static void gpu_tests() {
    using predicates::can_pod_fit;
    using predicates::check_node_validity;
    using predicates::does_node_selector_match;

    // --- the reference's three tests, src/predicates/test.rs:42-58 ---
    run("test_does_node_selector_match_no_selector (KAT-S1)", [] { CHECK(does_node_selector_match(test_pod(), test_node()) == true); });
    run("test_does_node_selector_match_false (KAT-S2)", [] { CHECK(does_node_selector_match(test_pod("foo", "bar"), test_node()) == false); });
    run("test_does_node_selector_match_true (KAT-S3)", [] { CHECK(does_node_selector_match(test_pod("name", NODE_NAME), test_node()) == true); });

    run("does_node_selector_match: D-S4..D-S9", [] {
        corev1::Pod empty_sel = test_pod();
        empty_sel.spec = corev1::PodSpec{};
        empty_sel.spec->node_selector = corev1::StringMap{};
        CHECK(does_node_selector_match(empty_sel, test_node()));  // D-S4
        corev1::Node nolabels;
        nolabels.metadata.name = "bare";
        CHECK(!does_node_selector_match(test_pod("a", "b"), nolabels));  // D-S5
        CHECK(does_node_selector_match(test_pod(), nolabels));           // D-S6
        CHECK(!does_node_selector_match(test_pod("name", "other"), test_node()));  // D-S7
        corev1::Pod two = test_pod("a", "1");
        (*two.spec->node_selector)["b"] = "2";
        corev1::Node n3 = test_node();
        n3.metadata.labels = corev1::StringMap{{"a", "1"}, {"b", "2"}, {"c", "3"}};
        CHECK(does_node_selector_match(two, n3));  // D-S8
        n3.metadata.labels = corev1::StringMap{{"a", "1"}, {"b", "9"}};
        CHECK(!does_node_selector_match(two, n3));
        corev1::Node ne = test_node();
        ne.metadata.labels = corev1::StringMap{{"a", ""}};
        CHECK(does_node_selector_match(test_pod("a", ""), ne));  // D-S9
        CHECK(!does_node_selector_match(test_pod("a", ""), test_node()));
    });

    run("can_pod_fit: D-R1..D-R8", [] {
        const char *GiB = "1073741824";
        {  // D-R1
            Context ctx = make_ctx({node_with("n", "1", GiB)});
            CHECK(can_pod_fit(pod_with("p", {container("500m", "134217728")}), ctx.node_store[0], ctx));
            CHECK(std::static_pointer_cast<StaticPodLister>(ctx.client)->list_calls == 1);  // one LIST per evaluation
        }
        {  // D-R2 exact fit, D-R3 one byte over
            Context ctx = make_ctx({node_with("n", "2", GiB)});
            CHECK(can_pod_fit(pod_with("p", {container("2", GiB)}), ctx.node_store[0], ctx));
            CHECK(!can_pod_fit(pod_with("p", {container("1", "1073741825")}), ctx.node_store[0], ctx));
            CHECK(!can_pod_fit(pod_with("p", {container("2001m", "1")}), ctx.node_store[0], ctx));
        }
        {  // D-R4 over-committed node, zero-request pod: 0 <= -x is false
            Context ctx = make_ctx({node_with("n", "1", GiB)}, {pod_with("hog", {container("1500m", "1")}, "n")});
            CHECK(!can_pod_fit(pod_with("p", {}), ctx.node_store[0], ctx));
        }
        {  // D-R5 no status: only a zero-request pod fits
            Context ctx = make_ctx({node_with("n", nullptr, nullptr)});
            CHECK(can_pod_fit(pod_with("p", {}), ctx.node_store[0], ctx));
            CHECK(!can_pod_fit(pod_with("p", {container("1m", nullptr)}), ctx.node_store[0], ctx));
        }
        {  // D-R6 two containers + one without resources; D-R7 init container ignored
            Context ctx = make_ctx({node_with("n", "500m", GiB)});
            corev1::Pod p = pod_with("p", {container("250m", nullptr), container("250m", nullptr)});
            corev1::Container bare;
            bare.name = "bare";
            p.spec->containers.push_back(bare);
            p.spec->init_containers.push_back(container("64", "1Ti"));
            CHECK(can_pod_fit(p, ctx.node_store[0], ctx));
            p.spec->containers.push_back(container("1m", nullptr));
            CHECK(!can_pod_fit(p, ctx.node_store[0], ctx));
        }
        {  // D-R8 a Succeeded pod bound to the node still counts
            corev1::Pod done = pod_with("done", {container("600m", "1")}, "n");
            done.status = corev1::PodStatus{std::string("Succeeded")};
            Context ctx = make_ctx({node_with("n", "1", GiB)}, {done});
            CHECK(!can_pod_fit(pod_with("p", {container("500m", "1")}), ctx.node_store[0], ctx));
            CHECK(can_pod_fit(pod_with("p", {container("400m", "1")}), ctx.node_store[0], ctx));
        }
        {  // allocatable lacks memory: the reference panics; here an exception, never a bit
            Context ctx = make_ctx({node_with("n", "1", nullptr)});
            CHECK_THROWS(can_pod_fit(pod_with("p", {}), ctx.node_store[0], ctx));
        }
    });

    run("check_node_validity: order of reasons (D-V1, D-V2; src/predicates.rs:68-74)", [] {
        corev1::Node n = node_with("n", "1", "1073741824");
        n.metadata.labels = corev1::StringMap{{"zone", "a"}};
        Context ctx = make_ctx({n});
        corev1::Pod big_wrong = pod_with("p", {container("2", "1")});
        big_wrong.spec->node_selector = corev1::StringMap{{"zone", "b"}};
        auto v1 = check_node_validity(big_wrong, n, ctx);
        CHECK(v1 && *v1 == InvalidNodeReason::NotEnoughResources);  // both fail: resources win
        corev1::Pod small_wrong = pod_with("p", {container("100m", "1")});
        small_wrong.spec->node_selector = corev1::StringMap{{"zone", "b"}};
        auto v2 = check_node_validity(small_wrong, n, ctx);
        CHECK(v2 && *v2 == InvalidNodeReason::NodeSelectorMismatch);
        corev1::Pod ok = pod_with("p", {container("100m", "1")});
        ok.spec->node_selector = corev1::StringMap{{"zone", "a"}};
        CHECK(!check_node_validity(ok, n, ctx).has_value());
    });

    run("explain_pairs: selector and taint failures told apart on the device (ksched_explain)", [] {
        corev1::Node plain = node_with("n-plain", "4", "8589934592"), tainted = node_with("n-tainted", "4", "8589934592"),
                     small = node_with("n-small", "1", "8589934592");
        plain.metadata.labels = corev1::StringMap{{"zone", "a"}};
        tainted.metadata.labels = corev1::StringMap{{"zone", "a"}};
        small.metadata.labels = corev1::StringMap{{"zone", "a"}};
        corev1::NodeSpec ns;
        ns.taints = std::vector<corev1::Taint>{{"dedicated", std::string("gpu"), "NoSchedule"}};
        tainted.spec = ns;
        Context ctx = make_ctx({tainted, small, plain});  // canonical order: n-plain 0, n-small 1, n-tainted 2
        corev1::Pod ok = pod_with("ok", {container("2", "1")}), wrong_zone = pod_with("wz", {container("2", "1")});
        ok.spec->node_selector = corev1::StringMap{{"zone", "a"}};
        wrong_zone.spec->node_selector = corev1::StringMap{{"zone", "b"}};
        const std::vector<const corev1::Pod *> pods = {&ok, &wrong_zone};
        const auto v = predicates::explain_pairs(pods, ctx, {{0, 0}, {0, 1}, {0, 2}, {1, 0}, {1, 2}, {1, 1}}, /*taints=*/true);
        CHECK(!v[0]);                                                        // fits, zone matches, no taint
        CHECK(v[1] && *v[1] == InvalidNodeReason::NotEnoughResources);       // 2 cpu on a 1-cpu node
        CHECK(v[2] && *v[2] == InvalidNodeReason::TaintNotTolerated);        // two masks would say NodeSelectorMismatch here
        CHECK(v[3] && *v[3] == InvalidNodeReason::NodeSelectorMismatch);
        CHECK(v[4] && *v[4] == InvalidNodeReason::NodeSelectorMismatch);     // selector before taint
        CHECK(v[5] && *v[5] == InvalidNodeReason::NotEnoughResources);       // resources first (src/predicates.rs:68-70)
        const auto w = predicates::explain_pairs(pods, ctx, {{0, 2}}, /*taints=*/false);
        CHECK(!w[0]);                                                        // the reference's own path has no taint predicate
    });
    run("select_node_for_pod: D-P1..D-P3 (src/main.rs:51-71)", [] {
        std::vector<corev1::Node> nodes;
        for (int i = 0; i < 10; ++i) nodes.push_back(node_with("node-" + std::to_string(i), i == 7 ? "4" : "100m", "1073741824"));
        Context ctx = make_ctx(nodes);
        corev1::Pod pod = pod_with("p", {container("1", "1")});
        ScriptedChooser s;
        s.script = {3, 3, 7, 1, 0};
        std::vector<RejectedCandidate> rej;
        auto got = select_node_for_pod(pod, ctx, s, &rej);
        CHECK(got && corev1::name_any(got->metadata) == "node-7");  // D-P1: third attempt
        CHECK(rej.size() == 2 && rej[0].node_name == "node-3" && rej[0].reason == InvalidNodeReason::NotEnoughResources);
        CHECK(s.next == 3);  // the reference stops drawing after the first success
        ScriptedChooser miss;
        miss.script = {0, 1, 2, 3, 4};
        CHECK(!select_node_for_pod(pod, ctx, miss).has_value());  // D-P3: a feasible node exists, the draws miss it
        Context empty = make_ctx({});
        ScriptedChooser any;
        any.script = {1, 2, 3};
        CHECK(!select_node_for_pod(pod, empty, any).has_value());  // D-P2: empty store
    });

    run("select_nodes_for_pods (batched) == select_node_for_pod draw for draw", [] {
        // 100 pods x 20 nodes (BASELINE.json configs[0] shape), labels + bound load
        std::vector<corev1::Node> nodes;
        std::vector<corev1::Pod> bound;
        for (int i = 0; i < 20; ++i) {
            corev1::Node n = node_with("node-" + std::to_string(100 + (i * 7) % 20), (i % 3) ? "4" : "2", "8589934592");
            n.metadata.labels = corev1::StringMap{{"zone", (i % 2) ? "a" : "b"}, {"tier", std::to_string(i % 4)}};
            if (i % 5 == 0) n.metadata.labels.reset();
            nodes.push_back(n);
            bound.push_back(pod_with("load-" + std::to_string(i), {container((i % 4) ? "1500m" : "3900m", "1073741824")},
                                     corev1::name_any(n.metadata).c_str()));
        }
        std::vector<corev1::Pod> pods;
        for (int i = 0; i < 100; ++i) {
            corev1::Pod p = pod_with("pod-" + std::to_string(i), {container((i % 3) ? "500m" : "2500m", "2147483648")});
            if (i % 4 == 1) p.spec->node_selector = corev1::StringMap{{"zone", "a"}};
            if (i % 4 == 2) p.spec->node_selector = corev1::StringMap{{"zone", "b"}, {"tier", "2"}};
            if (i % 10 == 3) p.spec->node_selector = corev1::StringMap{{"gpu", "yes"}};
            pods.push_back(p);
        }
        Context ctx = make_ctx(nodes, bound);
        std::vector<const corev1::Pod *> ptrs;
        for (auto &p : pods) ptrs.push_back(&p);
        SplitMixChooser c1(7);
        BatchSelection sel = select_nodes_for_pods(ptrs, ctx, c1, /*want_rejected=*/true);
        CHECK(sel.node_store_index.size() == 100);
        // per-pod reference loop with the same draw stream: consume ATTEMPTS draws per pod
        SplitMixChooser c2(7);
        int bound_n = 0;
        for (int i = 0; i < 100; ++i) {
            ScriptedChooser five;
            for (uint32_t t = 0; t < ATTEMPTS; ++t) five.script.push_back(*c2.choose(nodes.size()));
            std::vector<RejectedCandidate> rej;
            auto one = select_node_for_pod(pods[i], ctx, five, &rej);
            const int32_t idx = sel.node_store_index[i];
            CHECK(one.has_value() == (idx >= 0));
            if (one && idx >= 0) CHECK(corev1::name_any(one->metadata) == corev1::name_any(nodes[idx].metadata));
            CHECK(rej.size() == sel.rejected[i].size());
            for (size_t k = 0; k < rej.size() && k < sel.rejected[i].size(); ++k)
                CHECK(rej[k].node_name == sel.rejected[i][k].node_name && rej[k].reason == sel.rejected[i][k].reason);
            bound_n += idx >= 0;
        }
        CHECK(bound_n > 10 && bound_n < 100);
        // every (pod, node) bit of the batch equals the per-pair check_node_validity
        for (int i = 0; i < 100; i += 9)
            for (int j = 0; j < 20; ++j) {
                auto v = check_node_validity(pods[i], nodes[j], ctx);
                auto b = sel.validity.validity(i, ctx.snapshot->canonical_index(j));
                CHECK(v.has_value() == b.has_value());
                if (v && b) CHECK(*v == *b);
            }
    });

    run("a pod with more than KSCHED_MAX_KEYS selector keys is scheduled like any other (the reference has no limit, src/predicates.rs:48-53): nothing is isolated", [] {
        std::vector<corev1::Node> nodes = {node_with("node-a", "64", "68719476736"), node_with("node-b", "64", "68719476736")};
        corev1::StringMap many;
        for (uint32_t k = 0; k <= KSCHED_MAX_KEYS; ++k) many["key-" + std::to_string(k)] = "v";  // 33 keys on ONE pod
        nodes[0].metadata.labels = many;  // node-a carries them all, node-b none
        Context ctx = make_ctx(nodes);
        PodBatcher b(16);
        std::vector<corev1::Pod> pods;
        for (int i = 0; i < 7; ++i) pods.push_back(pod_with("p" + std::to_string(i), {container("100m", "1048576")}));
        pods[3].spec->node_selector = many;
        for (auto &p : pods) CHECK(b.push(std::make_shared<corev1::Pod>(p)));
        b.close();
        RecordingSink sink;
        SplitMixChooser chooser(3);
        int done_n = 0, failed_n = 0;
        const BatchLoopStats st = run_batches(
            b, [&](const std::vector<const corev1::Pod *> &batch) { return reconcile_batch(batch, ctx, chooser, sink); },
            [&](const PodBatcher::PodPtr &p, const ReconcileOutcome &o) {
                ++done_n;
                if (*p->metadata.name == "p3") CHECK(!o.bound_to || *o.bound_to == "node-a");  // only node-a matches its 33 keys (bound when a draw hit it)
                else CHECK(o.ok && o.bound_to);
            },
            [&](const PodBatcher::PodPtr &, const std::string &) { ++failed_n; });
        CHECK(done_n == 7 && failed_n == 0);
        CHECK(st.isolated_batches == 0 && st.failed_pods == 0 && st.pods == 7);
    });
    run("selectors wider than one device call (40 keys): batch masks == per-pair predicates, the pick comes from the ANDed mask, rejected reasons by ksched_explain == by the masks", [] {
        auto labels40 = [](const char *v17) {
            corev1::StringMap m;
            for (int k = 0; k < 40; ++k) m["wide-" + std::to_string(100 + k)] = (k == 17) ? v17 : "v";
            return m;
        };
        std::vector<corev1::Node> nodes;
        for (int i = 0; i < 9; ++i) {
            corev1::Node n = node_with("node-" + std::to_string(i), (i % 4 == 3) ? "100m" : "8", "17179869184");
            if (i % 3 == 0) n.metadata.labels = labels40("v");       // carries all forty
            else if (i % 3 == 1) n.metadata.labels = labels40("x");  // one value differs (key 17: in the SECOND group of 32 sorted keys or the first -- either way one group fails)
            else {
                corev1::StringMap few = labels40("v");
                few.erase("wide-139");  // the last key missing: the second group fails, the first does not
                n.metadata.labels = few;
            }
            n.metadata.labels->insert({"zone", (i % 2) ? "a" : "b"});
            nodes.push_back(n);
        }
        std::vector<corev1::Pod> pods;
        for (int i = 0; i < 23; ++i) {
            corev1::Pod p = pod_with("pod-" + std::to_string(i), {container((i % 5 == 4) ? "6" : "250m", "1073741824")});
            if (i % 4 == 1) p.spec->node_selector = labels40("v");  // wide: matches nodes 0, 3, 6 (node 3 is too small for nothing here: 250m fits 100m? no -> resources)
            if (i % 4 == 2) {
                corev1::StringMap m = labels40("v");
                m["zone"] = "a";  // 41 keys
                p.spec->node_selector = m;
            }
            if (i % 4 == 3) p.spec->node_selector = corev1::StringMap{{"zone", "b"}};
            if (i == 9) {
                corev1::StringMap m = labels40("v");
                m["wide-120"] = "nobody-has-this";
                p.spec->node_selector = m;
            }
            pods.push_back(p);
        }
        Context ctx = make_ctx(nodes);
        std::vector<const corev1::Pod *> ptrs;
        for (auto &p : pods) ptrs.push_back(&p);
        SplitMixChooser c1(11);
        const BatchSelection sel = select_nodes_for_pods(ptrs, ctx, c1, /*want_rejected=*/true);
        int wide_bound = 0, any_feasible_wide = 0;
        for (size_t i = 0; i < pods.size(); ++i) {
            for (size_t j = 0; j < nodes.size(); ++j) {
                const auto v = check_node_validity(pods[i], nodes[j], ctx);  // per pair (itself group by group for a wide pod)
                const auto b = sel.validity.validity((uint32_t)i, ctx.snapshot->canonical_index((uint32_t)j));
                CHECK(v.has_value() == b.has_value());
                if (v && b) CHECK(*v == *b);
                CHECK(does_node_selector_match(pods[i], nodes[j]) == !(v && *v == InvalidNodeReason::NodeSelectorMismatch) || (v && *v == InvalidNodeReason::NotEnoughResources));
                if (!v && pods[i].spec->node_selector && pods[i].spec->node_selector->size() > KSCHED_MAX_KEYS) ++any_feasible_wide;
            }
            // the binding is the first feasible draw (read from the very masks the groups ANDed into)
            int32_t want = -1;
            for (uint32_t t = 0; t < ATTEMPTS && want < 0; ++t) {
                const uint32_t s = sel.samples[i * ATTEMPTS + t];
                if (s < ctx.snapshot->n() && sel.validity.is_valid((uint32_t)i, s)) want = (int32_t)s;
            }
            CHECK(sel.validity.binding[i] == want);
            if (want >= 0 && pods[i].spec->node_selector && pods[i].spec->node_selector->size() > KSCHED_MAX_KEYS) ++wide_bound;
        }
        CHECK(any_feasible_wide > 0 && wide_bound > 0);
        CHECK(sel.validity.feasible_count(9) == 0);  // a value nobody carries, among forty keys
        // bindings only (what reconcile_batch asks for) + the rejected candidates by ksched_explain: the same lists as from the masks
        SplitMixChooser c2(11);
        const BatchSelection quiet = select_nodes_for_pods(ptrs, ctx, c2);
        CHECK(quiet.validity.binding == sel.validity.binding && quiet.validity.feasible.empty());
        const auto why = explain_rejected(ptrs, ctx, quiet);
        CHECK(why.size() == sel.rejected.size());
        size_t lines = 0;
        for (size_t i = 0; i < why.size(); ++i) {
            CHECK(why[i].size() == sel.rejected[i].size());
            for (size_t k = 0; k < why[i].size() && k < sel.rejected[i].size(); ++k, ++lines)
                CHECK(why[i][k].node_name == sel.rejected[i][k].node_name && why[i][k].reason == sel.rejected[i][k].reason);
        }
        CHECK(lines > 10);
        // with the taint extension on: a node that is tainted AND fails a key of the wide pod's LAST group is a selector mismatch (the reference's
        // order, then E2), whatever the earlier groups say
        {
            std::vector<corev1::Node> tn = nodes;
            for (auto &n : tn) {
                if (!n.spec) n.spec = corev1::NodeSpec{};
                n.spec->taints = std::vector<corev1::Taint>{corev1::Taint{"dedicated", std::string("gpu"), "NoSchedule"}};
            }
            Context tctx = make_ctx(tn);
            std::vector<std::pair<uint32_t, uint32_t>> pairs;
            for (uint32_t j = 0; j < (uint32_t)tn.size(); ++j) pairs.push_back({1u, j});  // pod 1: all forty keys = "v"
            const auto reasons = predicates::explain_pairs(ptrs, tctx, pairs, true);
            tctx.refresh_snapshot();
            const auto &names = tctx.snapshot->columns().names;
            size_t taint_only = 0, mismatch = 0;
            for (uint32_t j = 0; j < (uint32_t)tn.size(); ++j) {
                size_t store = 0;
                while (tn[store].metadata.name.value_or("") != names[j]) ++store;
                const bool sel_ok = does_node_selector_match(pods[1], tn[store]);
                const bool fit_ok = can_pod_fit(pods[1], tn[store], tctx);
                if (!fit_ok) CHECK(reasons[j] && *reasons[j] == InvalidNodeReason::NotEnoughResources);
                else if (!sel_ok) { CHECK(reasons[j] && *reasons[j] == InvalidNodeReason::NodeSelectorMismatch); ++mismatch; }
                else { CHECK(reasons[j] && *reasons[j] == InvalidNodeReason::TaintNotTolerated); ++taint_only; }
            }
            CHECK(taint_only > 0 && mismatch > 0);
        }
        // best fit over a wide pod: the device picks from the combined mask
        const predicates::BatchValidity bf = predicates::check_node_validity_batch(ptrs, ctx, false, KSCHED_PICK_BESTFIT, nullptr, 0, true);
        for (size_t i = 0; i < pods.size(); ++i) {
            if (bf.feasible_count((uint32_t)i) == 0) CHECK(bf.binding[i] == -1);
            else CHECK(bf.binding[i] >= 0 && bf.is_valid((uint32_t)i, (uint32_t)bf.binding[i]));
        }
    });
    run("the reference's WARN line for every rejected candidate (src/main.rs:62) from the batched path: same text, same order, nothing when the level is off", [] {
        std::vector<corev1::Node> nodes = {node_with("node-a", "4", "8589934592"), node_with("node-b", "100m", "1")};
        nodes[0].metadata.labels = corev1::StringMap{{"disk", "ssd"}};
        Context ctx = make_ctx(nodes);
        std::vector<std::string> lines;
        ctx.warn = [&](const std::string &l) { lines.push_back(l); };
        corev1::Pod fits = pod_with("fits", {container("1", "1")});
        corev1::Pod picky = pod_with("picky", {container("50m", "1")});
        picky.spec->node_selector = corev1::StringMap{{"disk", "hdd"}};
        std::vector<const corev1::Pod *> ptrs = {&fits, &picky};
        RecordingSink sink;
        ScriptedChooser c;
        c.script = {1, 1, 0, 0, 0, /* picky: */ 0, 1, 0, 1, 1};  // store indices; fits: node-b twice (too small), then node-a wins
        const auto out = reconcile_batch(ptrs, ctx, c, sink);
        CHECK(out[0].ok && out[0].bound_to && *out[0].bound_to == "node-a");
        CHECK(!out[1].ok && out[1].error == ReconcileError::NoNodeFound);
        const std::vector<std::string> want = {
            "Node node-b failed validity check for pod test/fits: NotEnoughResources",
            "Node node-b failed validity check for pod test/fits: NotEnoughResources",
            "Node node-a failed validity check for pod test/picky: NodeSelectorMismatch",
            "Node node-b failed validity check for pod test/picky: NodeSelectorMismatch",  // (a one-byte node: 50m / 1 byte fit it, the selector does not)
            "Node node-a failed validity check for pod test/picky: NodeSelectorMismatch",
            "Node node-b failed validity check for pod test/picky: NodeSelectorMismatch",
            "Node node-b failed validity check for pod test/picky: NodeSelectorMismatch",
        };
        // ... followed by error_policy's line for the reconcile that found no node (src/main.rs:123)
        std::vector<std::string> with_policy = want;
        with_policy.push_back("reconcile failed on pod test/picky: NoNodeFound");
        CHECK(lines == with_policy);
        if (lines != with_policy)
            for (const auto &l : lines) std::printf("    got: %s\n", l.c_str());
        // the per-pod path of the reference's own shape says the same for the same draws
        std::vector<std::string> one;
        ctx.warn = [&](const std::string &l) { one.push_back(l); };
        ScriptedChooser c2;
        c2.script = {1, 1, 0};
        CHECK(select_node_for_pod(fits, ctx, c2).has_value());
        CHECK(one.size() == 2 && one[0] == want[0]);
        // level off: no lines, and no explain call is made (the evaluator is not even asked)
        lines.clear();
        ctx.warn = nullptr;
        ScriptedChooser c3;
        c3.script = c.script;
        RecordingSink sink2;
        (void)reconcile_batch(ptrs, ctx, c3, sink2);
        CHECK(lines.empty());
    });
    run("reconcile / reconcile_batch (src/main.rs:73-125)", [] {
        std::vector<corev1::Node> nodes = {node_with("node-a", "4", "8589934592"), node_with("node-b", "100m", "1")};
        Context ctx = make_ctx(nodes);
        RecordingSink sink;
        corev1::Pod fits = pod_with("fits", {container("1", "1")});
        corev1::Pod never = pod_with("never", {container("64", "1")});
        corev1::Pod already = pod_with("already", {container("1", "1")}, "node-a");
        ScriptedChooser c;
        c.script = {1, 0};
        ReconcileOutcome r = reconcile(fits, ctx, c, sink);
        CHECK(r.ok && r.action == Action::AwaitChange && r.bound_to && *r.bound_to == "node-a");
        CHECK(sink.posts.size() == 1 && sink.posts[0].first == "test/fits" && sink.posts[0].second == "node-a");
        ScriptedChooser c2;
        c2.script = {0, 1, 0, 1, 0};
        r = reconcile(never, ctx, c2, sink);
        CHECK(!r.ok && r.error == ReconcileError::NoNodeFound && r.action == Action::RequeueAfter5Min);
        ScriptedChooser c3;
        r = reconcile(already, ctx, c3, sink);
        CHECK(r.ok && !r.bound_to && c3.next == 0);  // bound pods are skipped before any draw
        sink.fail = true;
        ScriptedChooser c4;
        c4.script = {0};
        r = reconcile(fits, ctx, c4, sink);
        CHECK(!r.ok && r.error == ReconcileError::CreateBindingFailed);
        sink.fail = false;
        sink.posts.clear();
        SplitMixChooser c5(99);
        std::vector<const corev1::Pod *> batch = {&fits, &never, &already, &fits};
        std::vector<ReconcileOutcome> out = reconcile_batch(batch, ctx, c5, sink);
        CHECK(out.size() == 4);
        CHECK(!out[1].ok && out[1].error == ReconcileError::NoNodeFound);
        CHECK(out[2].ok && !out[2].bound_to);
        for (int i : {0, 3}) CHECK(out[i].ok ? (out[i].bound_to && *out[i].bound_to == "node-a") : out[i].error == ReconcileError::NoNodeFound);
    });

    run("snapshot builder on the device: incremental ksched_update_nodes == fresh ksched_set_nodes (8f n1)", [] {
        // 1500 nodes (two index tiles, the second partial); pods bind to a few of them one event at a time
        std::vector<corev1::Node> nodes;
        for (int i = 0; i < 1500; ++i) {
            char name[32];
            std::snprintf(name, sizeof name, "node-%04d", i);
            nodes.push_back(node_with(name, (i % 3 == 0) ? "4" : "2", "8589934592"));
            nodes.back().metadata.labels = corev1::StringMap{{"zone", (i % 2) ? "a" : "b"}};
        }
        std::vector<corev1::Pod> probes = {pod_with("small", {container("500m", "1")}), pod_with("mid", {container("1800m", "1")}),
                                           pod_with("big", {container("3500m", "1")}), test_pod("zone", "a")};
        std::vector<const corev1::Pod *> pp;
        for (auto &p : probes) pp.push_back(&p);
        Context inc = make_ctx(nodes);
        inc.refresh_snapshot();
        std::vector<corev1::Pod> landed;
        SplitMixChooser rng(5);
        for (int e = 0; e < 40; ++e) {
            char name[32];
            std::snprintf(name, sizeof name, "node-%04d", (int)*rng.choose(1500));
            landed.push_back(pod_with("l" + std::to_string(e), {container((e % 2) ? "700m" : "1500m", "1073741824")}, name));
            CHECK(inc.snapshot->apply_bound_pod(landed.back()));
        }
        CHECK(inc.snapshot->apply_deleted_pod(landed[3]));
        landed.erase(landed.begin() + 3);
        Context full = make_ctx(nodes, landed);  // the same state, re-LISTed and uploaded from scratch
        full.refresh_snapshot();
        const predicates::BatchValidity a = predicates::check_node_validity_batch(pp, inc, false, KSCHED_PICK_BESTFIT);
        const predicates::BatchValidity b = predicates::check_node_validity_batch(pp, full, false, KSCHED_PICK_BESTFIT);
        CHECK(a.feasible == b.feasible && a.fit == b.fit && a.binding == b.binding);
        CHECK(a.feasible_count(2) < a.feasible_count(0));  // the events mattered
        CHECK(std::static_pointer_cast<StaticPodLister>(inc.client)->list_calls == 1500);  // no LIST after the first build
    });

    run("a device failure after the host commit: the snapshot is stale, the next evaluation uploads everything (KSCHED_OPT_FAULT)", [] {
        std::vector<corev1::Node> nodes;
        for (int i = 0; i < 40; ++i) nodes.push_back(node_with("node-" + std::string(i < 10 ? "0" : "") + std::to_string(i), "4", "8589934592"));
        std::vector<corev1::Pod> probes = {pod_with("mid", {container("2500m", "1")}), pod_with("small", {container("500m", "1")})};
        std::vector<const corev1::Pod *> pp = {&probes[0], &probes[1]};
        Context ctx = make_ctx(nodes);
        ctx.refresh_snapshot();
        const predicates::BatchValidity before = predicates::check_node_validity_batch(pp, ctx, false, 0);
        CHECK(before.feasible_count(0) == 40);
        // the watch reports a pod on node-07; the device update fails inside the library (an injected std::bad_alloc -> KSCHED_E_NOMEM)
        corev1::Pod landed = pod_with("l0", {container("2", "1")}, "node-07");
        CHECK(ksched_set_option(ctx.snapshot->device().handle(), KSCHED_OPT_FAULT, 1) == KSCHED_OK);
        CHECK_THROWS(ctx.snapshot->observe_pod(Snapshot::PodEvent::Applied, landed));
        CHECK(ctx.snapshot->device_stale());
        CHECK(ctx.snapshot->counted_pods() == 1);  // the host bookkeeping is committed (the echo of this event will change nothing) ...
        CHECK(ctx.snapshot->observe_pods({{Snapshot::PodEvent::Applied, &landed}}) == 0);
        // ... and the device catches up before the next evaluation: node-07 no longer holds the 2.5-core pod
        const predicates::BatchValidity after = predicates::check_node_validity_batch(pp, ctx, false, 0);
        CHECK(!ctx.snapshot->device_stale());
        CHECK(after.feasible_count(0) == 39 && after.feasible_count(1) == 40);
        Context fresh = make_ctx(nodes, {landed});
        fresh.refresh_snapshot();
        const predicates::BatchValidity want = predicates::check_node_validity_batch(pp, fresh, false, 0);
        CHECK(after.feasible == want.feasible);
        // reconcile_batch keeps its outcomes when the snapshot update after the POSTs fails (the bindings exist by then)
        RecordingSink sink;
        SplitMixChooser rng(9);
        std::vector<corev1::Pod> batch = {pod_with("b0", {container("100m", "1")}), pod_with("b1", {container("100m", "1")})};
        CHECK(ksched_set_option(ctx.snapshot->device().handle(), KSCHED_OPT_FAULT, 1 | (1 << 8)) == KSCHED_OK);  // skip the evaluation's fault point, hit the update's
        const std::vector<ReconcileOutcome> out = reconcile_batch({&batch[0], &batch[1]}, ctx, rng, sink);
        CHECK(out.size() == 2 && out[0].ok && out[1].ok && sink.posts.size() == 2);
        CHECK(ctx.snapshot && ctx.snapshot->device_stale());
        const predicates::BatchValidity later = predicates::check_node_validity_batch(pp, ctx, false, 0);  // uploads, then evaluates
        CHECK(!ctx.snapshot->device_stale() && later.feasible_count(1) == 40);
    });

    run("reconcile_batch_sequential: in-batch capacity accounting never over-commits (8f n3, opt-in)", [] {
        // 6 nodes x 2 CPU; 20 pods x 1 CPU: at most 12 can land.  The reference-faithful batch over-commits.
        std::vector<corev1::Node> nodes;
        for (int i = 0; i < 6; ++i) nodes.push_back(node_with("n" + std::to_string(i), "2", "8589934592"));
        std::vector<corev1::Pod> pods;
        for (int i = 0; i < 20; ++i) pods.push_back(pod_with("p" + std::to_string(i), {container("1", "1073741824")}));
        std::vector<const corev1::Pod *> pp;
        for (auto &p : pods) pp.push_back(&p);
        auto landed_per_node = [&](const std::vector<ReconcileOutcome> &out) {
            std::map<std::string, int> m;
            for (const auto &o : out)
                if (o.ok && o.bound_to) ++m[*o.bound_to];
            return m;
        };
        {
            Context ctx = make_ctx(nodes);
            RecordingSink sink;
            SplitMixChooser c(7);
            int worst = 0, total = 0;
            for (const auto &[node, cnt] : landed_per_node(reconcile_batch(pp, ctx, c, sink))) worst = std::max(worst, cnt), total += cnt;
            CHECK(total == 20 && worst > 2);  // every pod saw the same snapshot: legal for the reference, over-committed
        }
        {
            Context ctx = make_ctx(nodes);
            RecordingSink sink;
            SplitMixChooser c(7);
            SequentialStats st;
            const std::vector<ReconcileOutcome> out = reconcile_batch_sequential(pp, ctx, c, sink, 64, &st);
            int total = 0;
            for (const auto &[node, cnt] : landed_per_node(out)) {
                CHECK(cnt <= 2);
                total += cnt;
            }
            CHECK(total <= 12 && total >= 8);
            CHECK((int)sink.posts.size() == total);
            CHECK(st.rounds >= 2 && st.conflicts > 0);
            int none = 0;
            for (const auto &o : out)
                if (!o.ok) {
                    CHECK(o.error == ReconcileError::NoNodeFound && o.action == Action::RequeueAfter5Min);
                    ++none;
                }
            CHECK(none == 20 - total);
            // the snapshot now carries the batch's bindings: what is left is exactly capacity - landed
            const NodeColumns &cols = ctx.snapshot->columns();
            int64_t left = 0;
            for (int64_t v : cols.avail_cpu_milli) {
                CHECK(v >= 0);
                left += v;
            }
            CHECK(left == 12000 - 1000 * total);
        }
    });
}

// The exchange step behind the C ABI (include/ksched.h "multi-GPU"): bindings produced by ksched_eval_device on a stream are
// all-gathered by ksched_allgather_bindings on the SAME stream -- no host sync, no torch.  One rank is what one GPU allows; both
// communicator constructors are exercised (one process per GPU: unique id + ksched_comm_create; one process, n devices:
// ksched_comm_create_local + the grouped all-gather).
static void comm_tests() {
    run("RCCL all-gather of the bindings through the C ABI (one rank): ksched_comm_create / _create_local", [] {
        const uint32_t n = 300, p = 1000, attempts = ATTEMPTS;
        std::vector<int64_t> ncpu(n), nmem(n), pcpu(p), pmem(p);
        std::vector<uint32_t> samples((size_t)p * attempts);
        SplitMixChooser rng(11);
        for (uint32_t i = 0; i < n; ++i) {
            ncpu[i] = 500 + (int64_t)*rng.choose(8000);
            nmem[i] = (int64_t)1 << (20 + *rng.choose(14));
        }
        for (uint32_t i = 0; i < p; ++i) {
            pcpu[i] = 100 + (int64_t)*rng.choose(6000);
            pmem[i] = (int64_t)1 << (18 + *rng.choose(14));
            for (uint32_t t = 0; t < attempts; ++t) samples[(size_t)i * attempts + t] = (uint32_t)*rng.choose(n);
        }
        ksched_ctx *c = nullptr;
        CHECK(ksched_create(&c, 0) == KSCHED_OK);
        CHECK(ksched_set_nodes(c, n, ncpu.data(), nmem.data(), nullptr, 0, nullptr) == KSCHED_OK);
        std::vector<int32_t> want(p);
        CHECK(ksched_eval(c, p, pcpu.data(), pmem.data(), nullptr, nullptr, samples.data(), attempts, KSCHED_FIT | KSCHED_PICK_SAMPLED, nullptr,
                          nullptr, want.data()) == KSCHED_OK);
        int bound = 0;
        for (int32_t b : want) bound += b >= 0;
        CHECK(bound > 50 && bound < (int)p);  // a non-degenerate batch
        int64_t *d_pcpu = nullptr, *d_pmem = nullptr;
        uint32_t *d_smp = nullptr;
        int32_t *d_local = nullptr, *d_all = nullptr;
        hipStream_t s = nullptr;
        CHECK(hipMalloc((void **)&d_pcpu, p * 8) == hipSuccess && hipMalloc((void **)&d_pmem, p * 8) == hipSuccess);
        CHECK(hipMalloc((void **)&d_smp, samples.size() * 4) == hipSuccess);
        CHECK(hipMalloc((void **)&d_local, p * 4) == hipSuccess && hipMalloc((void **)&d_all, p * 4) == hipSuccess);
        CHECK(hipStreamCreate(&s) == hipSuccess);
        CHECK(hipMemcpy(d_pcpu, pcpu.data(), p * 8, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemcpy(d_pmem, pmem.data(), p * 8, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemcpy(d_smp, samples.data(), samples.size() * 4, hipMemcpyHostToDevice) == hipSuccess);
        auto run_once = [&](const std::function<int()> &gather) {
            CHECK(hipMemsetAsync(d_local, 0x7F, p * 4, s) == hipSuccess);
            CHECK(hipMemsetAsync(d_all, 0x7F, p * 4, s) == hipSuccess);
            CHECK(ksched_eval_device(c, p, d_pcpu, d_pmem, nullptr, nullptr, d_smp, attempts, KSCHED_FIT | KSCHED_PICK_SAMPLED, nullptr, nullptr,
                                     d_local, s) == KSCHED_OK);
            const int rc = gather();  // enqueued behind the pick on the same stream
            if (rc != KSCHED_OK) std::printf("    gather failed: %s / %s\n", ksched_strerror(rc), ksched_comm_last_error());
            CHECK(rc == KSCHED_OK);
            std::vector<int32_t> got(p);
            CHECK(hipMemcpyAsync(got.data(), d_all, p * 4, hipMemcpyDeviceToHost, s) == hipSuccess);
            CHECK(hipStreamSynchronize(s) == hipSuccess);
            CHECK(got == want);
        };
        // one process per GPU
        uint8_t id[KSCHED_COMM_ID_BYTES];
        ksched_comm *q = nullptr;
        CHECK(ksched_comm_unique_id(id) == KSCHED_OK);
        CHECK(ksched_comm_create(c, id, 0, 1, &q) == KSCHED_OK);
        CHECK(ksched_comm_rank(q) == 0 && ksched_comm_size(q) == 1);
        run_once([&] { return ksched_allgather_bindings(q, d_local, d_all, p, s); });
        CHECK(ksched_allgather_bindings(q, nullptr, d_all, p, s) == KSCHED_E_INVAL);
        CHECK(ksched_comm_create(c, id, 1, 1, &q) == KSCHED_E_INVAL);  // rank out of range (q untouched on failure? no: it is reset)
        ksched_comm_destroy(q);
        // one process, n devices (n = 1 here)
        ksched_comm *qs[1] = {nullptr};
        ksched_ctx *ctxs[1] = {c};
        CHECK(ksched_comm_create_local(ctxs, 1, qs) == KSCHED_OK);
        const int32_t *locals[1] = {d_local};
        int32_t *alls[1] = {d_all};
        void *streams[1] = {s};
        run_once([&] { return ksched_allgather_bindings_local(qs, 1, locals, alls, p, streams); });
        ksched_comm_destroy(qs[0]);
        (void)hipFree(d_pcpu); (void)hipFree(d_pmem); (void)hipFree(d_smp); (void)hipFree(d_local); (void)hipFree(d_all);
        CHECK(ksched_forget_stream(c, s) == KSCHED_OK);  // the stream goes away before the ctx does
        (void)hipStreamDestroy(s);
        ksched_destroy(c);
    });
}

// One host process, several devices (host/sharded.hpp; include/ksched.h "one host thread, several devices").  A test box has ONE GPU:
//   * the product path -- RCCL communicator from ksched_comm_create_local, ksched_eval_begin / ksched_allgather_bindings_local /
//     ksched_eval_end -- runs with one device (n = 1) through the mirror's own entry points and must give what the single-device
//     path gives, draw for draw and bit for bit;
//   * the shard arithmetic on the device -- unequal and empty shards, selector columns addressed with the whole batch's stride,
//     masks landing in the right rows -- runs with 2 .. 5 evaluators on the one GPU and the HostCopies exchange (RCCL refuses a
//     communicator that names one device twice), against one plain ksched_eval of the whole batch.
static void sharded_tests() {
    run("sharded host path, one device, RCCL exchange == single-device path (select_nodes_for_pods, masks, reasons)", [] {
        std::vector<corev1::Node> nodes;
        std::vector<corev1::Pod> bound;
        for (int i = 0; i < 37; ++i) {
            corev1::Node n = node_with("node-" + std::to_string(100 + (i * 7) % 37), (i % 3) ? "4" : "2", "8589934592");
            n.metadata.labels = corev1::StringMap{{"zone", (i % 2) ? "a" : "b"}, {"tier", std::to_string(i % 4)}};
            if (i % 5 == 0) n.metadata.labels.reset();
            nodes.push_back(n);
            bound.push_back(pod_with("load-" + std::to_string(i), {container((i % 4) ? "1500m" : "3900m", "1073741824")}, corev1::name_any(n.metadata).c_str()));
        }
        std::vector<corev1::Pod> pods;
        for (int i = 0; i < 301; ++i) {
            corev1::Pod p = pod_with("pod-" + std::to_string(i), {container((i % 3) ? "500m" : "2500m", "2147483648")});
            if (i % 4 == 1) p.spec->node_selector = corev1::StringMap{{"zone", "a"}};
            if (i % 4 == 2) p.spec->node_selector = corev1::StringMap{{"zone", "b"}, {"tier", "2"}};
            if (i % 10 == 3) p.spec->node_selector = corev1::StringMap{{"gpu", "yes"}};
            pods.push_back(p);
        }
        std::vector<const corev1::Pod *> ptrs;
        for (auto &p : pods) ptrs.push_back(&p);
        Context plain = make_ctx(nodes, bound), sharded = make_ctx(nodes, bound);
        plain.snapshot = std::make_shared<Snapshot>(0);
        plain.snapshot->rebuild(plain.node_store, plain.client.get());
        sharded.snapshot = std::make_shared<Snapshot>(std::vector<int>{0}, /*force_sharded=*/true);
        sharded.snapshot->rebuild(sharded.node_store, sharded.client.get());
        CHECK(plain.snapshot->sharded() == nullptr);
        CHECK(sharded.snapshot->sharded() != nullptr && sharded.snapshot->sharded()->size() == 1 &&
              sharded.snapshot->sharded()->exchange() == ShardedContext::Exchange::Rccl);
        for (bool want_rejected : {false, true}) {
            SplitMixChooser c1(7), c2(7);
            const BatchSelection a = select_nodes_for_pods(ptrs, plain, c1, want_rejected), b = select_nodes_for_pods(ptrs, sharded, c2, want_rejected);
            CHECK(a.node_store_index == b.node_store_index);
            CHECK(a.validity.binding == b.validity.binding && a.validity.feasible == b.validity.feasible && a.validity.fit == b.validity.fit);
            int bound_n = 0;
            for (int32_t x : b.node_store_index) bound_n += x >= 0;
            CHECK(bound_n > 30 && bound_n < 301);
            if (want_rejected) {
                CHECK(a.rejected.size() == b.rejected.size());
                for (size_t i = 0; i < a.rejected.size(); ++i) {
                    CHECK(a.rejected[i].size() == b.rejected[i].size());
                    for (size_t k = 0; k < a.rejected[i].size() && k < b.rejected[i].size(); ++k)
                        CHECK(a.rejected[i][k].node_name == b.rejected[i][k].node_name && a.rejected[i][k].reason == b.rejected[i][k].reason);
                }
            }
        }
        CHECK(sharded.snapshot->sharded()->batches() == 2);
        // the callers: a batch reconciled through the sharded context leaves the same POSTs and the same snapshot behind
        RecordingSink s1, s2;
        SplitMixChooser c1(99), c2(99);
        const auto o1 = reconcile_batch(ptrs, plain, c1, s1), o2 = reconcile_batch(ptrs, sharded, c2, s2);
        CHECK(s1.posts == s2.posts && !s1.posts.empty());
        for (size_t i = 0; i < o1.size(); ++i) CHECK(o1[i].ok == o2[i].ok && o1[i].bound_to == o2[i].bound_to);
        CHECK(plain.snapshot->columns().avail_cpu_milli == sharded.snapshot->columns().avail_cpu_milli);
        // ... and the NEXT batch sees the bindings of this one on every replica (ksched_update_nodes went to every device)
        SplitMixChooser d1(5), d2(5);
        CHECK(select_nodes_for_pods(ptrs, plain, d1).node_store_index == select_nodes_for_pods(ptrs, sharded, d2).node_store_index);
    });
    run("sharded host path: 2 .. 5 shards (evaluators on one GPU, HostCopies exchange) == one ksched_eval of the whole batch", [] {
        const uint32_t n = 1500, keys = 3, attempts = ATTEMPTS;
        SplitMixChooser rng(2024);
        NodeColumns nc;
        std::vector<int64_t> ncpu(n), nmem(n);
        std::vector<uint32_t> nlab((size_t)keys * n);
        for (uint32_t i = 0; i < n; ++i) {
            ncpu[i] = 500 + (int64_t)*rng.choose(8000);
            nmem[i] = (int64_t)1 << (20 + *rng.choose(14));
            for (uint32_t k = 0; k < keys; ++k) nlab[(size_t)k * n + i] = (uint32_t)*rng.choose(4 + k);  // 0 = key absent
        }
        const uint32_t W = ksched_mask_words(n);
        for (uint32_t shards : {2u, 3u, 5u})
            for (uint32_t p : {1u, 4u, 1001u}) {  // fewer pods than shards (empty shards), ragged last shard
                PodColumns pc;
                pc.p = p;
                pc.n_keys = keys;
                pc.req_cpu_milli.resize(p);
                pc.req_mem_bytes.resize(p);
                pc.sel_val_ids.assign((size_t)keys * p, 0u);
                std::vector<uint32_t> samples((size_t)p * attempts);
                for (uint32_t i = 0; i < p; ++i) {
                    pc.req_cpu_milli[i] = 100 + (int64_t)*rng.choose(6000);
                    pc.req_mem_bytes[i] = (int64_t)1 << (18 + *rng.choose(14));
                    for (uint32_t k = 0; k < keys; ++k)
                        if (*rng.choose(5) == 0) pc.sel_val_ids[(size_t)k * p + i] = (*rng.choose(20) == 0) ? KSCHED_SEL_NEVER : 1u + (uint32_t)*rng.choose(3 + k);
                    for (uint32_t t = 0; t < attempts; ++t) samples[(size_t)i * attempts + t] = (uint32_t)*rng.choose(n + 2);  // (some draws out of range)
                }
                std::vector<std::shared_ptr<DeviceEvaluator>> devs;
                for (uint32_t r = 0; r < shards; ++r) {
                    devs.push_back(std::make_shared<DeviceEvaluator>(0));
                    CHECK(ksched_set_nodes(devs.back()->handle(), n, ncpu.data(), nmem.data(), nlab.data(), keys, nullptr) == KSCHED_OK);  // replicated
                }
                ShardedContext sh(devs, ShardedContext::Exchange::HostCopies);
                const uint32_t flags = KSCHED_FIT | KSCHED_SEL | KSCHED_PICK_SAMPLED | KSCHED_WANT_FIT_MASK;
                std::vector<uint64_t> feas((size_t)p * W, 0xABull), fit((size_t)p * W, 0xCDull), feas1((size_t)p * W), fit1((size_t)p * W);
                std::vector<int32_t> bind(p, 12345), bind1(p);
                sh.eval(pc, samples.data(), attempts, flags, W, feas.data(), fit.data(), bind.data());
                CHECK(ksched_eval(devs[0]->handle(), p, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(), pc.sel_val_ids.data(), nullptr, samples.data(), attempts, flags,
                                  feas1.data(), fit1.data(), bind1.data()) == KSCHED_OK);
                CHECK(feas == feas1 && fit == fit1 && bind == bind1);
                // bindings only (what reconcile_batch asks for): no mask buffers at all
                std::vector<int32_t> bind2(p, 777);
                sh.eval(pc, samples.data(), attempts, KSCHED_FIT | KSCHED_SEL | KSCHED_PICK_SAMPLED, W, nullptr, nullptr, bind2.data());
                CHECK(bind2 == bind1);
            }
    });
    run("sharded host path: a shard whose evaluation fails inside the library (KSCHED_OPT_FAULT) -> EncodeError, every device drained, the next batch is whole", [] {
        const uint32_t n = 900, p = 2000, attempts = ATTEMPTS;
        SplitMixChooser rng(404);
        std::vector<int64_t> ncpu(n), nmem(n);
        for (uint32_t i = 0; i < n; ++i) {
            ncpu[i] = 500 + (int64_t)*rng.choose(8000);
            nmem[i] = (int64_t)1 << (20 + *rng.choose(14));
        }
        PodColumns pc;
        pc.p = p;
        pc.req_cpu_milli.resize(p);
        pc.req_mem_bytes.resize(p);
        std::vector<uint32_t> samples((size_t)p * attempts);
        for (uint32_t i = 0; i < p; ++i) {
            pc.req_cpu_milli[i] = 100 + (int64_t)*rng.choose(6000);
            pc.req_mem_bytes[i] = (int64_t)1 << (18 + *rng.choose(14));
            for (uint32_t t = 0; t < attempts; ++t) samples[(size_t)i * attempts + t] = (uint32_t)*rng.choose(n);
        }
        std::vector<std::shared_ptr<DeviceEvaluator>> devs;
        for (int r = 0; r < 3; ++r) {
            devs.push_back(std::make_shared<DeviceEvaluator>(0));
            CHECK(ksched_set_nodes(devs.back()->handle(), n, ncpu.data(), nmem.data(), nullptr, 0, nullptr) == KSCHED_OK);
        }
        ShardedContext sh(devs, ShardedContext::Exchange::HostCopies);
        const uint32_t flags = KSCHED_FIT | KSCHED_PICK_SAMPLED;
        std::vector<int32_t> want(p), got(p, 31337);
        CHECK(ksched_eval(devs[0]->handle(), p, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(), nullptr, nullptr, samples.data(), attempts, flags, nullptr, nullptr,
                          want.data()) == KSCHED_OK);
        for (int victim : {0, 1, 2}) {  // the first, a middle and the last shard
            CHECK(ksched_set_option(devs[(size_t)victim]->handle(), KSCHED_OPT_FAULT, 2) == KSCHED_OK);  // the next entry into that ctx throws std::runtime_error
            std::fill(got.begin(), got.end(), 31337);
            CHECK_THROWS(sh.eval(pc, samples.data(), attempts, flags, ksched_mask_words(n), nullptr, nullptr, got.data()));
            for (int32_t b : got) CHECK(b == 31337);  // nothing half-written comes back
            sh.eval(pc, samples.data(), attempts, flags, ksched_mask_words(n), nullptr, nullptr, got.data());  // the contexts work on
            CHECK(got == want);
        }
        CHECK(sh.batches() == 3);
    });
    run("sharded host path: a communicator over one device named twice is refused loudly (no silent stand-in)", [] {
        std::vector<std::shared_ptr<DeviceEvaluator>> devs = {std::make_shared<DeviceEvaluator>(0), std::make_shared<DeviceEvaluator>(0)};
        CHECK_THROWS(ShardedContext(devs));
    });
}

// The exchange with n > 1 on a one-GPU box.  Needs $KSCHED_TEST_HOOKS=1 and $KSCHED_RCCL_LIB=tests/cpp/libfake_rccl.so
// (tests/test_host_mirror.py sets both): the library then loads the TEST-ONLY stand-in of tests/cpp/fake_rccl.cpp, whose ncclCommInitAll
// accepts one device n times and whose grouped ncclAllGather is n x n stream-ordered copies -- so the product sequence
//     ksched_comm_create_local -> ksched_eval_begin x n -> ksched_gather_buffer x n -> ksched_allgather_bindings_local -> ksched_eval_end(gathered_0)
// runs through Exchange::Rccl with n = 2 .. 8, which Exchange::HostCopies bypasses.  Not covered: the real xGMI transport.
static void sharded_rccl_tests() {
    struct Cluster {
        uint32_t n, keys;
        std::vector<int64_t> ncpu, nmem;
        std::vector<uint32_t> nlab;
    };
    auto make_cluster = [](uint32_t n, uint32_t keys, SplitMixChooser &rng) {
        Cluster c{n, keys, std::vector<int64_t>(n), std::vector<int64_t>(n), std::vector<uint32_t>((size_t)keys * n)};
        for (uint32_t i = 0; i < n; ++i) {
            c.ncpu[i] = 500 + (int64_t)*rng.choose(8000);
            c.nmem[i] = (int64_t)1 << (20 + *rng.choose(14));
            for (uint32_t k = 0; k < keys; ++k) c.nlab[(size_t)k * n + i] = (uint32_t)*rng.choose(4 + k);
        }
        return c;
    };
    auto make_pods = [](uint32_t p, const Cluster &c, uint32_t attempts, SplitMixChooser &rng, PodColumns &pc, std::vector<uint32_t> &samples) {
        pc = PodColumns{};
        pc.p = p;
        pc.n_keys = c.keys;
        pc.req_cpu_milli.resize(p);
        pc.req_mem_bytes.resize(p);
        pc.sel_val_ids.assign((size_t)c.keys * p, 0u);
        samples.assign((size_t)p * attempts, 0u);
        for (uint32_t i = 0; i < p; ++i) {
            pc.req_cpu_milli[i] = 100 + (int64_t)*rng.choose(6000);
            pc.req_mem_bytes[i] = (int64_t)1 << (18 + *rng.choose(14));
            for (uint32_t k = 0; k < c.keys; ++k)
                if (*rng.choose(5) == 0) pc.sel_val_ids[(size_t)k * p + i] = (*rng.choose(20) == 0) ? KSCHED_SEL_NEVER : 1u + (uint32_t)*rng.choose(3 + k);
            for (uint32_t t = 0; t < attempts; ++t) samples[(size_t)i * attempts + t] = (uint32_t)*rng.choose(c.n + 2);
        }
    };
    auto replicas = [](uint32_t shards, const Cluster &c) {
        std::vector<std::shared_ptr<DeviceEvaluator>> devs;
        for (uint32_t r = 0; r < shards; ++r) {
            devs.push_back(std::make_shared<DeviceEvaluator>(0));
            CHECK(ksched_set_nodes(devs.back()->handle(), c.n, c.ncpu.data(), c.nmem.data(), c.keys ? c.nlab.data() : nullptr, c.keys, nullptr) == KSCHED_OK);
        }
        return devs;
    };
    run("the stand-in is what the library loaded (and only because the test hooks are switched on)", [] {
        const char *lib = std::getenv("KSCHED_RCCL_LIB"), *hooks = std::getenv("KSCHED_TEST_HOOKS");
        CHECK(lib && *lib && hooks && std::string(hooks) == "1");
        std::vector<std::shared_ptr<DeviceEvaluator>> devs = {std::make_shared<DeviceEvaluator>(0), std::make_shared<DeviceEvaluator>(0)};
        ShardedContext sh(devs);  // the real RCCL refuses this ("a communicator over one device named twice", sharded_tests)
        CHECK(sh.size() == 2 && sh.exchange() == ShardedContext::Exchange::Rccl);
    });
    run("Exchange::Rccl with 2 .. 8 shards on one GPU == one ksched_eval of the whole batch (masks, fit masks, bindings; empty and ragged shards)", [&] {
        SplitMixChooser rng(515);
        const Cluster c = make_cluster(1500, 3, rng);
        const uint32_t W = ksched_mask_words(c.n), attempts = ATTEMPTS;
        for (uint32_t shards : {2u, 3u, 4u, 5u, 8u})
            for (uint32_t p : {1u, 7u, 1001u, 20000u}) {
                PodColumns pc;
                std::vector<uint32_t> samples;
                make_pods(p, c, attempts, rng, pc, samples);
                auto devs = replicas(shards, c);
                ShardedContext sh(devs);
                const uint32_t flags = KSCHED_FIT | KSCHED_SEL | KSCHED_PICK_SAMPLED | KSCHED_WANT_FIT_MASK;
                std::vector<uint64_t> feas((size_t)p * W, 0xABull), fit((size_t)p * W, 0xCDull), feas1((size_t)p * W), fit1((size_t)p * W);
                std::vector<int32_t> bind(p, 12345), bind1(p);
                sh.eval(pc, samples.data(), attempts, flags, W, feas.data(), fit.data(), bind.data());
                CHECK(ksched_eval(devs[0]->handle(), p, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(), pc.sel_val_ids.data(), nullptr, samples.data(), attempts, flags,
                                  feas1.data(), fit1.data(), bind1.data()) == KSCHED_OK);
                CHECK(feas == feas1 && fit == fit1 && bind == bind1);
                std::vector<int32_t> bind2(p, 777);
                sh.eval(pc, samples.data(), attempts, KSCHED_FIT | KSCHED_SEL | KSCHED_PICK_SAMPLED, W, nullptr, nullptr, bind2.data());
                CHECK(bind2 == bind1);
                std::vector<int32_t> bind3(p, 778);  // best fit: its own launches ahead of the gather on each device's stream
                sh.eval(pc, nullptr, 0, KSCHED_FIT | KSCHED_SEL | KSCHED_PICK_BESTFIT, W, nullptr, nullptr, bind3.data());
                CHECK(ksched_eval(devs[0]->handle(), p, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(), pc.sel_val_ids.data(), nullptr, nullptr, 0,
                                  KSCHED_FIT | KSCHED_SEL | KSCHED_PICK_BESTFIT, nullptr, nullptr, bind1.data()) == KSCHED_OK);
                CHECK(bind3 == bind1);
            }
    });
    run("Exchange::Rccl, 4 shards: forty batches back to back through ONE context (stream order between a device's pick, the gather and the next batch's copies in)", [&] {
        SplitMixChooser rng(99);
        const Cluster c = make_cluster(2300, 2, rng);
        const uint32_t W = ksched_mask_words(c.n), attempts = ATTEMPTS;
        auto devs = replicas(4, c);
        ShardedContext sh(devs);
        for (int b = 0; b < 40; ++b) {
            const uint32_t p = 1u + (uint32_t)*rng.choose(6000);
            PodColumns pc;
            std::vector<uint32_t> samples;
            make_pods(p, c, attempts, rng, pc, samples);
            std::vector<int32_t> got(p, -5), want(p);
            sh.eval(pc, samples.data(), attempts, KSCHED_FIT | KSCHED_SEL | KSCHED_PICK_SAMPLED, W, nullptr, nullptr, got.data());
            CHECK(ksched_eval(devs[(size_t)b % 4]->handle(), p, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(), pc.sel_val_ids.data(), nullptr, samples.data(), attempts,
                              KSCHED_FIT | KSCHED_SEL | KSCHED_PICK_SAMPLED, nullptr, nullptr, want.data()) == KSCHED_OK);
            CHECK(got == want);
        }
        CHECK(sh.batches() == 40);
    });
    run("Exchange::Rccl, 3 shards: a shard failing inside the library (KSCHED_OPT_FAULT) -> EncodeError, every device drained, no collective issued, the next batch whole", [&] {
        SplitMixChooser rng(404);
        const Cluster c = make_cluster(900, 0, rng);
        const uint32_t p = 2000, attempts = ATTEMPTS;
        PodColumns pc;
        std::vector<uint32_t> samples;
        make_pods(p, c, attempts, rng, pc, samples);
        auto devs = replicas(3, c);
        ShardedContext sh(devs);
        const uint32_t flags = KSCHED_FIT | KSCHED_PICK_SAMPLED;
        std::vector<int32_t> want(p), got(p, 31337);
        CHECK(ksched_eval(devs[0]->handle(), p, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(), nullptr, nullptr, samples.data(), attempts, flags, nullptr, nullptr,
                          want.data()) == KSCHED_OK);
        for (int victim : {0, 1, 2}) {
            CHECK(ksched_set_option(devs[(size_t)victim]->handle(), KSCHED_OPT_FAULT, 2) == KSCHED_OK);
            std::fill(got.begin(), got.end(), 31337);
            CHECK_THROWS(sh.eval(pc, samples.data(), attempts, flags, ksched_mask_words(c.n), nullptr, nullptr, got.data()));
            for (int32_t b : got) CHECK(b == 31337);
            CHECK(!sh.broken());  // the exchange was never started: the communicator is intact
            sh.eval(pc, samples.data(), attempts, flags, ksched_mask_words(c.n), nullptr, nullptr, got.data());
            CHECK(got == want);
        }
    });
    run("Exchange::Rccl, 3 shards: the collective fails for one rank inside the group -> EncodeError, the clique aborted, nothing hangs, a new context works", [&] {
        SplitMixChooser rng(405);
        const Cluster c = make_cluster(700, 0, rng);
        const uint32_t p = 900, attempts = ATTEMPTS;
        PodColumns pc;
        std::vector<uint32_t> samples;
        make_pods(p, c, attempts, rng, pc, samples);
        auto devs = replicas(3, c);
        const uint32_t flags = KSCHED_FIT | KSCHED_PICK_SAMPLED;
        std::vector<int32_t> want(p), got(p, 31337);
        CHECK(ksched_eval(devs[0]->handle(), p, pc.req_cpu_milli.data(), pc.req_mem_bytes.data(), nullptr, nullptr, samples.data(), attempts, flags, nullptr, nullptr,
                          want.data()) == KSCHED_OK);
        {
            ShardedContext sh(devs);
            sh.eval(pc, samples.data(), attempts, flags, ksched_mask_words(c.n), nullptr, nullptr, got.data());
            CHECK(got == want);
            // the stand-in counts ncclAllGather calls per process: make the SECOND rank's call of the next batch fail (rank 0 has then
            // already put its part of the collective into the group)
            void *fake = dlopen(std::getenv("KSCHED_RCCL_LIB"), RTLD_NOW | RTLD_NOLOAD);
            CHECK(fake != nullptr);
            auto calls = fake ? reinterpret_cast<long (*)()>(dlsym(fake, "fake_rccl_allgather_calls")) : nullptr;
            auto collectives = fake ? reinterpret_cast<long (*)()>(dlsym(fake, "fake_rccl_collectives")) : nullptr;
            CHECK(calls && collectives && collectives() > 0);  // the batches above did go through the stand-in's grouped all-gather
            const long done_before = collectives ? collectives() : 0;
            if (calls) setenv("FAKE_RCCL_FAIL_ALLGATHER", std::to_string(calls() + 2).c_str(), 1);
            bool failed = false;
            std::fill(got.begin(), got.end(), 31337);
            try {
                sh.eval(pc, samples.data(), attempts, flags, ksched_mask_words(c.n), nullptr, nullptr, got.data());
            } catch (const EncodeError &e) {
                failed = true;
                CHECK(std::string(e.what()).find("aborted") != std::string::npos);
                for (int32_t b : got) CHECK(b == 31337);  // nothing half-written comes back
            }
            CHECK(collectives && collectives() == done_before);  // nothing was enqueued for the half-issued collective
            unsetenv("FAKE_RCCL_FAIL_ALLGATHER");
            CHECK(failed && sh.broken());
            CHECK_THROWS(sh.eval(pc, samples.data(), attempts, flags, ksched_mask_words(c.n), nullptr, nullptr, got.data()));  // for good
        }
        ShardedContext again(devs);  // the evaluators themselves are fine: a new clique over them works
        std::fill(got.begin(), got.end(), 31337);
        again.eval(pc, samples.data(), attempts, flags, ksched_mask_words(c.n), nullptr, nullptr, got.data());
        CHECK(got == want);
    });
    run("the mirror's own entry points over THREE replicas of one device: select_nodes_for_pods / reconcile_batch == the single-device path", [] {
        std::vector<corev1::Node> nodes;
        std::vector<corev1::Pod> bound;
        for (int i = 0; i < 41; ++i) {
            corev1::Node n = node_with("node-" + std::to_string(100 + (i * 7) % 41), (i % 3) ? "4" : "2", "8589934592");
            n.metadata.labels = corev1::StringMap{{"zone", (i % 2) ? "a" : "b"}, {"tier", std::to_string(i % 4)}};
            if (i % 5 == 0) n.metadata.labels.reset();
            nodes.push_back(n);
            bound.push_back(pod_with("load-" + std::to_string(i), {container((i % 4) ? "1500m" : "3900m", "1073741824")}, corev1::name_any(n.metadata).c_str()));
        }
        std::vector<corev1::Pod> pods;
        for (int i = 0; i < 333; ++i) {
            corev1::Pod p = pod_with("pod-" + std::to_string(i), {container((i % 3) ? "500m" : "2500m", "2147483648")});
            if (i % 4 == 1) p.spec->node_selector = corev1::StringMap{{"zone", "a"}};
            if (i % 4 == 2) p.spec->node_selector = corev1::StringMap{{"zone", "b"}, {"tier", "2"}};
            if (i % 10 == 3) p.spec->node_selector = corev1::StringMap{{"gpu", "yes"}};
            pods.push_back(p);
        }
        std::vector<const corev1::Pod *> ptrs;
        for (auto &p : pods) ptrs.push_back(&p);
        Context plain = make_ctx(nodes, bound), sharded = make_ctx(nodes, bound);
        plain.snapshot = std::make_shared<Snapshot>(0);
        plain.snapshot->rebuild(plain.node_store, plain.client.get());
        sharded.snapshot = std::make_shared<Snapshot>(std::vector<int>{0, 0, 0});
        sharded.snapshot->rebuild(sharded.node_store, sharded.client.get());
        CHECK(sharded.snapshot->sharded() != nullptr && sharded.snapshot->sharded()->size() == 3);
        for (bool want_rejected : {false, true}) {
            SplitMixChooser c1(7), c2(7);
            const BatchSelection a = select_nodes_for_pods(ptrs, plain, c1, want_rejected), b = select_nodes_for_pods(ptrs, sharded, c2, want_rejected);
            CHECK(a.node_store_index == b.node_store_index);
            CHECK(a.validity.binding == b.validity.binding && a.validity.feasible == b.validity.feasible && a.validity.fit == b.validity.fit);
        }
        RecordingSink s1, s2;
        SplitMixChooser c1(99), c2(99);
        const auto o1 = reconcile_batch(ptrs, plain, c1, s1), o2 = reconcile_batch(ptrs, sharded, c2, s2);
        CHECK(s1.posts == s2.posts && !s1.posts.empty());
        for (size_t i = 0; i < o1.size(); ++i) CHECK(o1[i].ok == o2[i].ok && o1[i].bound_to == o2[i].bound_to);
        CHECK(plain.snapshot->columns().avail_cpu_milli == sharded.snapshot->columns().avail_cpu_milli);
        SplitMixChooser d1(5), d2(5);  // the bindings of that batch reached every replica (ksched_update_nodes on each)
        CHECK(select_nodes_for_pods(ptrs, plain, d1).node_store_index == select_nodes_for_pods(ptrs, sharded, d2).node_store_index);
    });
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "cpu";
    if (mode == "cpu") cpu_tests();
    else if (mode == "gpu") gpu_tests();
    else if (mode == "comm") comm_tests();
    else if (mode == "sharded") sharded_tests();
    else if (mode == "sharded_rccl") sharded_rccl_tests();
    else {
        std::printf("usage: host_tests cpu|gpu|comm|sharded|sharded_rccl\n");
        return 2;
    }
    std::printf("%d test(s), %d failed check(s)\n", g_run, g_fail);
    return g_fail ? 1 : 0;
}
