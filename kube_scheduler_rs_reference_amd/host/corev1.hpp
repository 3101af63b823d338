// corev1.hpp -- the subset of k8s_openapi::api::core::v1 the predicate path reads.
//
// The reference manipulates k8s_openapi 0.18 structs (`corev1::Pod`, `corev1::Node`, Cargo.toml:10);
// `Option<T>` there is `std::optional<T>` here and `BTreeMap<String, String>` is `std::map`
// (same ordered iteration, which src/predicates.rs:48 relies on).  Field names follow the Rust
// structs (snake_case), not the JSON spelling.
#pragma once
#include <map>
#include <optional>
#include <string>
#include <vector>

namespace corev1 {

using Quantity = std::string;  // k8s_openapi: pub struct Quantity(pub String)
using StringMap = std::map<std::string, std::string>;

struct ObjectMeta {
    std::optional<std::string> name;
    std::optional<std::string> namespace_;
    std::optional<StringMap> labels;
};

struct ResourceRequirements {
    std::optional<std::map<std::string, Quantity>> requests;
    std::optional<std::map<std::string, Quantity>> limits;
};

struct Container {
    std::string name;
    std::optional<ResourceRequirements> resources;
};

struct Toleration {
    std::optional<std::string> key, operator_, value, effect;
};

struct Taint {
    std::string key;
    std::optional<std::string> value;
    std::string effect;
};

struct PodSpec {
    std::vector<Container> containers;
    std::vector<Container> init_containers;  // read from the wire, ignored by the path (src/util.rs:58)
    std::optional<StringMap> node_selector;
    std::optional<std::string> node_name;
    std::optional<std::vector<Toleration>> tolerations;
};

struct PodStatus {
    std::optional<std::string> phase;
};

struct Pod {
    ObjectMeta metadata;
    std::optional<PodSpec> spec;
    std::optional<PodStatus> status;
};

struct NodeSpec {
    std::optional<std::vector<Taint>> taints;
};

struct NodeStatus {
    std::optional<std::map<std::string, Quantity>> allocatable;
};

struct Node {
    ObjectMeta metadata;
    std::optional<NodeSpec> spec;
    std::optional<NodeStatus> status;
};

// kube::ResourceExt::name_any: metadata.name or "" (used at src/predicates.rs:23)
inline std::string name_any(const ObjectMeta &m) { return m.name.value_or(std::string()); }

}  // namespace corev1
