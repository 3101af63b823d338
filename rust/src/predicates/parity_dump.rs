// src/predicates/parity_dump.rs -- runs THIS crate's own predicates (the reference's, untouched arithmetic in
// kube_quantity) on the ksched repository's golden objects and writes what they answer.  No GPU, no ksched library.
//
//   KSCHED_GOLDEN_DIR=<ksched repo>/tests/golden cargo test --release parity_dump -- --nocapture
//   (rust/pin_parity.sh does exactly that on a patched copy of the reference)
//
// For every <name>_objects.json in the directory ({"pods", "nodes", "bound", "samples"}: Kubernetes JSON objects, nodes in
// canonical order) it writes ref_<name>.json:
//   fit[p][w], sel[p][w], feasible_fit_and_sel[p][w]   pod-major mask rows, bit (n % 64) of word (n / 64), as 16-digit hex
//       fit  = the pure half of can_pod_fit (`fits`) on (pod, node, pods LISTed on that node)
//       sel  = does_node_selector_match(pod, node)
//   sampled[p]   select_node_for_pod's loop with the injected draws of "samples": first draw passing fits && selector, else -1
//   panics       pairs on which a predicate panicked (expect / index), counted as infeasible
// tests/test_reference_fixtures.py (ksched repository) compares these files with its committed fixtures.
use std::collections::BTreeMap;
use std::panic::{catch_unwind, AssertUnwindSafe};

use k8s_openapi::api::core::v1 as corev1;

use super::*;

fn objects<T: serde::de::DeserializeOwned>(doc: &serde_json::Value, key: &str) -> Vec<T> {
    return doc[key]
        .as_array()
        .unwrap_or_else(|| panic!("objects file lacks the array '{}'", key))
        .iter()
        .map(|v| serde_json::from_value(v.clone()).unwrap_or_else(|e| panic!("{}: not a valid object: {}", key, e)))
        .collect();
}

fn hex_rows(rows: &[Vec<u64>]) -> serde_json::Value {
    return serde_json::Value::Array(
        rows.iter().map(|r| serde_json::Value::Array(r.iter().map(|w| serde_json::Value::String(format!("{:016x}", w))).collect())).collect(),
    );
}

fn dump_one(path: &std::path::Path, out_dir: &std::path::Path) {
    let text = std::fs::read_to_string(path).expect("cannot read objects file");
    let doc: serde_json::Value = serde_json::from_str(&text).expect("objects file is not JSON");
    let name = doc["name"].as_str().expect("objects file lacks 'name'").to_string();
    let pods: Vec<corev1::Pod> = objects(&doc, "pods");
    let nodes: Vec<corev1::Node> = objects(&doc, "nodes");
    let bound: Vec<corev1::Pod> = objects(&doc, "bound");
    let samples: Vec<Vec<u64>> = doc["samples"]
        .as_array()
        .expect("objects file lacks 'samples'")
        .iter()
        .map(|r| r.as_array().expect("samples row").iter().map(|x| x.as_u64().expect("sample index")).collect())
        .collect();
    let (p, n) = (pods.len(), nodes.len());
    let words = (n + 63) / 64;

    // the LIST of src/predicates.rs:21-25,34, done once per node: every pod whose spec.nodeName is the node's name, any phase
    let mut lists: BTreeMap<String, Vec<corev1::Pod>> = BTreeMap::new();
    for b in &bound {
        if let Some(corev1::PodSpec { node_name: Some(nn), .. }) = &b.spec {
            lists.entry(nn.clone()).or_default().push(b.clone());
        }
    }
    let empty: Vec<corev1::Pod> = Vec::new();
    let mut fit = vec![vec![0u64; words]; p];
    let mut sel = vec![vec![0u64; words]; p];
    let mut both = vec![vec![0u64; words]; p];
    let mut panics: Vec<serde_json::Value> = Vec::new();
    for (i, pod) in pods.iter().enumerate() {
        for (j, node) in nodes.iter().enumerate() {
            let node_name = node.metadata.name.clone().unwrap_or_default();
            let on_node = lists.get(&node_name).unwrap_or(&empty);
            let f = match catch_unwind(AssertUnwindSafe(|| fits(pod, node, on_node))) {
                Ok(v) => v,
                Err(_) => {
                    panics.push(serde_json::json!({"pod": i, "node": j, "in": "fits"}));
                    false
                },
            };
            let s = does_node_selector_match(pod, node);
            if f {
                fit[i][j / 64] |= 1u64 << (j % 64);
            }
            if s {
                sel[i][j / 64] |= 1u64 << (j % 64);
            }
            if f && s {
                both[i][j / 64] |= 1u64 << (j % 64);
            }
        }
    }
    // select_node_for_pod (src/main.rs:51-71) with the draws injected: first draw whose check_node_validity is Ok wins
    let bit = |rows: &Vec<Vec<u64>>, i: usize, j: usize| (rows[i][j / 64] >> (j % 64)) & 1 == 1;
    let sampled: Vec<i64> = (0..p)
        .map(|i| {
            for &s in &samples[i] {
                let j = s as usize;
                if j < n && bit(&both, i, j) {
                    return j as i64;
                }
            }
            return -1;
        })
        .collect();
    let out = serde_json::json!({
        "name": name, "p": p, "n": n,
        "fit": hex_rows(&fit), "sel": hex_rows(&sel), "feasible_fit_and_sel": hex_rows(&both),
        "sampled": sampled, "panics": panics,
        "reference": "acrlabs/kube-scheduler-rs-reference src/predicates.rs (fits = pure half of can_pod_fit, does_node_selector_match), kube_quantity per Cargo.lock",
    });
    let dest = out_dir.join(format!("ref_{}.json", name));
    std::fs::write(&dest, serde_json::to_string(&out).unwrap() + "\n").expect("cannot write ref file");
    println!("parity_dump: {} pods x {} nodes -> {}", p, n, dest.display());
}

#[test]
fn parity_dump() {
    let dir = match std::env::var("KSCHED_GOLDEN_DIR") {
        Ok(d) => std::path::PathBuf::from(d),
        Err(_) => {
            println!("parity_dump: KSCHED_GOLDEN_DIR not set, nothing to do");
            return;
        },
    };
    let mut done = 0;
    let mut entries: Vec<_> = std::fs::read_dir(&dir).expect("cannot read KSCHED_GOLDEN_DIR").filter_map(|e| e.ok()).map(|e| e.path()).collect();
    entries.sort();
    for path in entries {
        if path.file_name().and_then(|f| f.to_str()).map(|f| f.ends_with("_objects.json")).unwrap_or(false) {
            dump_one(&path, &dir);
            done += 1;
        }
    }
    assert!(done > 0, "no *_objects.json under KSCHED_GOLDEN_DIR");
}
