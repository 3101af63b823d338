#include "util.hpp"

#include <cstdio>
#include <cstdlib>

#include "encoder.hpp"
#include "sharded.hpp"

extern "C" __attribute__((weak)) int ksched_test_hooks_enabled(void);  // defined by the TEST build of libksched_hip.so only (tests/cpp/test_hooks.cpp)

namespace ksched_host {

std::vector<corev1::Pod> StaticPodLister::list_pods_on_node(const std::string &node_name) {
    std::vector<corev1::Pod> out;
    for (const auto &p : pods)
        if (p.spec && p.spec->node_name && *p.spec->node_name == node_name) out.push_back(p);
    return out;
}

PodResources::PodResources() : cpu(ParsedQuantity::try_from("0")), memory(ParsedQuantity::try_from("0")) {}

bool is_pod_bound(const corev1::Pod &pod) { return pod.spec && pod.spec->node_name.has_value(); }

std::string full_name(const corev1::ObjectMeta &meta) {
    if (meta.namespace_) return *meta.namespace_ + "/" + corev1::name_any(meta);
    return corev1::name_any(meta);
}

// Containers only: init containers, overhead and limits never count (src/util.rs:58-69).
PodResources total_pod_resources(const corev1::Pod &pod) {
    PodResources r;
    if (pod.spec) {
        for (const auto &c : pod.spec->containers) {
            if (c.resources && c.resources->requests) {
                const auto &req = *c.resources->requests;
                if (auto it = req.find("cpu"); it != req.end()) r.cpu += ParsedQuantity::try_from(it->second);
                if (auto it = req.find("memory"); it != req.end()) r.memory += ParsedQuantity::try_from(it->second);
            }
        }
    }
    return r;
}

std::function<void(const std::string &)> Context::default_warn_sink() {
    const char *lvl = std::getenv("KSCHED_LOG");
    if (lvl && (std::string(lvl) == "off" || std::string(lvl) == "error")) return nullptr;
    return [](const std::string &line) { std::fprintf(stderr, " WARN %s\n", line.c_str()); };
}

void Context::refresh_snapshot() {
    if (!snapshot) {
        std::vector<int> devs_env = devices.empty() ? devices_from_env(std::getenv("KSCHED_DEVICES"), device) : devices;
        const char *force = std::getenv("KSCHED_SHARDED");
        const bool sharded = force && *force && *force != '0';
        // test hook (only when the evaluator library underneath is its TEST build -- it alone defines ksched_test_hooks_enabled -- and
        // $KSCHED_TEST_HOOKS=1; the exchange then needs the stand-in of $KSCHED_RCCL_LIB): KSCHED_SHARDED=k, k >= 2, with ONE device =
        // k evaluators on that device, so that a one-GPU box runs the mirror through a k-way row shard
        if (sharded && ksched_test_hooks_enabled != nullptr && ksched_test_hooks_enabled() && devs_env.size() == 1) {
            const long k = std::strtol(force, nullptr, 10);
            if (k >= 2 && k <= 64) devs_env.assign((size_t)k, devs_env[0]);
        }
        const std::vector<int> &devs = devs_env;
        snapshot = (devs.size() > 1 || sharded) ? std::make_shared<Snapshot>(devs, sharded) : std::make_shared<Snapshot>(devs[0]);
    }
    snapshot->rebuild(node_store, client.get());
}

}  // namespace ksched_host
