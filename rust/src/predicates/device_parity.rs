// src/predicates/device_parity.rs -- `cargo test --features ksched device_parity` on an MI355X box: the device evaluator
// (check_node_validity_batch -> ksched_eval) against THIS crate's own predicates on the same golden objects, bit for bit.
// This is BASELINE configs[0] ("100 pods x 20 nodes ... under cargo test") plus the other domain-D fixtures.
// Fails loudly at Evaluator::new without a GPU: the ksched library has no CPU fallback.
use std::collections::BTreeMap;
use std::sync::Arc;

use k8s_openapi::api::core::v1 as corev1;

use super::*;

fn load<T: serde::de::DeserializeOwned>(doc: &serde_json::Value, key: &str) -> Vec<T> {
    return doc[key].as_array().expect("array").iter().map(|v| serde_json::from_value(v.clone()).expect("object")).collect();
}

#[test]
fn device_parity() {
    let dir = std::path::PathBuf::from(std::env::var("KSCHED_GOLDEN_DIR").expect("set KSCHED_GOLDEN_DIR to <ksched repo>/tests/golden"));
    let evaluator = crate::ksched::Devices::new(&[0]).expect("ksched_create (needs an MI355X; there is no CPU fallback)");
    for name in ["c1_100x20", "ragged_70x130_taints", "one_node_33x1", "binsuffix_60x40", "wide_selectors_48x90"] {
        let text = std::fs::read_to_string(dir.join(format!("{}_objects.json", name))).expect("objects file");
        let doc: serde_json::Value = serde_json::from_str(&text).expect("json");
        let pods: Vec<corev1::Pod> = load(&doc, "pods");
        let nodes: Vec<Arc<corev1::Node>> = load::<corev1::Node>(&doc, "nodes").into_iter().rev().map(Arc::new).collect(); // store order != canonical order
        let bound: Vec<corev1::Pod> = load(&doc, "bound");
        let refs: Vec<&corev1::Pod> = pods.iter().collect();
        let (snapshot, validity) = check_node_validity_batch(&refs, &nodes, &bound, &evaluator).expect("device evaluation");
        let mut lists: BTreeMap<String, Vec<corev1::Pod>> = BTreeMap::new();
        for b in &bound {
            if let Some(corev1::PodSpec { node_name: Some(nn), .. }) = &b.spec {
                lists.entry(nn.clone()).or_default().push(b.clone());
            }
        }
        let empty: Vec<corev1::Pod> = Vec::new();
        for (i, pod) in pods.iter().enumerate() {
            for canonical in 0..snapshot.n() {
                let node = &nodes[snapshot.store_index[canonical as usize]];
                let on_node = lists.get(&snapshot.names[canonical as usize]).unwrap_or(&empty);
                let want = if !fits(pod, node, on_node) {
                    Err(InvalidNodeReason::NotEnoughResources)
                } else if !does_node_selector_match(pod, node) {
                    Err(InvalidNodeReason::NodeSelectorMismatch)
                } else {
                    Ok(())
                };
                let got = reason_of(validity.reason(i as u32, canonical));
                assert_eq!(format!("{:?}", got), format!("{:?}", want), "{}: pod {} node {}", name, i, snapshot.names[canonical as usize]);
                assert_eq!(validity.is_valid(i as u32, canonical), want.is_ok());
            }
        }
        println!("device_parity: {} ok ({} pods x {} nodes)", name, pods.len(), nodes.len());
    }
}
