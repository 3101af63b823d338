"""The Rust overlay (rust/) cannot be compiled here (no cargo / rustc).  What can be checked without a toolchain:
  * src/ksched_sys.rs declares exactly the functions include/ksched.h declares (and nothing else), with as many arguments;
  * its constants carry the header's values; its self-check tables list every function and constant it declares;
  * every `sys::` item src/ksched.rs uses is declared there;
  * no `dead_code` allowance anywhere: the batched pick is wired into the running binary, not parked;
  * every file (and every patched file) has balanced brackets once comments, strings and char literals are stripped;
  * every k8s-openapi field the overlay touches is one the reference's own sources use, or is in the short list below of
    fields of the pinned k8s-openapi 0.18 (Cargo.lock:681-682) -- a typo in a field name fails here, not at a maintainer's desk;
  * the wiring itself: under `--features ksched` select_node_for_pod goes through the batch task, main() spawns the batch
    task and the pod watch, Context carries the two fields, reconcile's POST is the reference's own text;
  * the patches apply cleanly to the reference checkout (only where /root/reference exists: this container, not the GPU box).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ksched.h")
SYS_RS = os.path.join(ROOT, "rust", "src", "ksched_sys.rs")
REFERENCE = "/root/reference"


def header_functions():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(ksched_\w+)\s*\(([^;{]*?)\)\s*;", text):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def rust_functions():
    text = re.sub(r"//[^\n]*", "", open(SYS_RS).read())
    out = {}
    for m in re.finditer(r"pub fn (ksched_\w+)\s*\(([^;]*?)\)\s*(?:->[^;]*)?;", text, flags=re.S):
        args = [a for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = len(args)
    return out


def test_rust_binding_declares_exactly_the_header():
    h, r = header_functions(), rust_functions()
    assert set(h) == set(r), f"only in header: {sorted(set(h) - set(r))}; only in ksched_sys.rs: {sorted(set(r) - set(h))}"
    assert {k: (h[k], r[k]) for k in h if h[k] != r[k]} == {}, "argument counts differ"


def test_rust_constants_carry_the_header_values():
    hdr = dict(re.findall(r"#define\s+(KSCHED_\w+)\s+\(?(-?(?:0x[0-9A-Fa-f]+|\d+))u?\)?\s", open(HEADER).read()))
    rs = dict(re.findall(r"pub const (KSCHED_\w+): \w+ = (-?(?:0x[0-9A-Fa-f_]+|\d+));", open(SYS_RS).read()))
    assert len(rs) >= 25
    for k, v in rs.items():
        assert k in hdr, f"{k} is not in include/ksched.h"
        assert int(v.replace("_", ""), 0) == int(hdr[k], 0), k


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="/root/reference is not on this box (GPU box): patch application is checked in the build container")
def test_patches_apply_cleanly_to_the_reference(tmp_path):
    if not shutil.which("patch"):
        pytest.skip("no `patch` binary")
    out = tmp_path / "scheduler-v0"
    subprocess.check_call(["bash", os.path.join(ROOT, "rust", "apply.sh"), REFERENCE, str(out)], stdout=subprocess.DEVNULL)
    pred = (out / "src" / "predicates.rs").read_text()
    assert "pub(crate) fn fits(" in pred and "mod parity_dump;" in pred
    assert "async fn can_pod_fit(pod: &corev1::Pod, node: &corev1::Node, ctx: &Context) -> bool" in pred  # signature kept (src/predicates.rs:20)
    assert "fn does_node_selector_match(pod: &corev1::Pod, node: &corev1::Node) -> bool" in pred          # :45
    assert not list(out.rglob("*.rej")) and not list(out.rglob("*.orig"))
    for f in ("build.rs", "src/ksched.rs", "src/ksched_sys.rs", "src/predicates/parity_dump.rs", "src/predicates/device_parity.rs"):
        assert (out / f).exists(), f


RUST_DIR = os.path.join(ROOT, "rust")


def rust_sources():
    out = {}
    for base, _, files in os.walk(RUST_DIR):
        for f in files:
            if f.endswith((".rs", ".patch", ".sh", ".md")):
                out[os.path.relpath(os.path.join(base, f), RUST_DIR)] = open(os.path.join(base, f)).read()
    return out


def strip_rust(text):
    """comments, string literals and char literals out (lifetimes like <'a> stay: they hold no brackets)"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r'b?"(?:\\.|[^"\\])*"', '""', text, flags=re.S)
    text = re.sub(r"b?'(?:\\.|[^'\\])'", "''", text)
    return text


def balanced(text):
    pairs = {")": "(", "]": "[", "}": "{"}
    stack = []
    for ch in strip_rust(text):
        if ch in "([{":
            stack.append(ch)
        elif ch in pairs:
            if not stack or stack.pop() != pairs[ch]:
                return False
    return not stack


def test_no_dead_code_allowance_anywhere():
    for name, text in rust_sources().items():
        if name.endswith((".rs", ".patch")):
            assert "dead_code" not in text, f"rust/{name} still carries a dead_code allowance"


def test_every_sys_item_used_is_declared_and_listed():
    sys_rs = open(SYS_RS).read()
    declared = set(re.findall(r"pub fn (ksched_\w+)", sys_rs)) | set(re.findall(r"pub const (KSCHED_\w+)", sys_rs)) | \
        set(re.findall(r"pub struct (ksched_\w+)", sys_rs)) | {"symbol_table", "constant_table"}
    used = set(re.findall(r"\bsys::(\w+)", open(os.path.join(RUST_DIR, "src", "ksched.rs")).read()))
    assert used and used <= declared, f"used but not declared in ksched_sys.rs: {sorted(used - declared)}"
    for patch in ("0001-predicates-fits-seam-and-batch.patch",):
        for item in re.findall(r"crate::ksched_sys::(\w+)", open(os.path.join(RUST_DIR, "patches", patch)).read()):
            assert item in declared, item
    # the self-check tables name every function / constant exactly once
    fn_tab = re.search(r"pub fn symbol_table\(\).*?\n    \];", sys_rs, flags=re.S).group(0)
    const_tab = re.search(r"pub fn constant_table\(\).*?\n    \];", sys_rs, flags=re.S).group(0)
    assert sorted(re.findall(r'\("(ksched_\w+)", \1 as usize\)', fn_tab)) == sorted(rust_functions())
    assert sorted(re.findall(r'\("(KSCHED_\w+)", \1 as i64\)', const_tab)) == sorted(re.findall(r"pub const (KSCHED_\w+):", sys_rs))


def test_brackets_balance_in_every_rust_file():
    for name, text in rust_sources().items():
        if name.endswith(".rs"):
            assert balanced(text), f"rust/{name}: unbalanced brackets"


# k8s-openapi 0.18 (v1_26) fields the overlay touches beyond the ones the reference's own sources already use.
PINNED_API_FIELDS = {
    "resource_version": "ObjectMeta::resource_version: Option<String>",
}


def overlay_field_uses(text):
    """every field ACCESS (`.name` not followed by a call) and every field named in a struct pattern (`name: Some(`, `name: None`)"""
    fields = set(re.findall(r"(?<![.\d])\.([a-z_][a-z0-9_]*)\b(?!\s*[(:!])", text))  # (not the `..` of a range)
    fields |= set(re.findall(r"\b([a-z_][a-z0-9_]*):\s*(?:Some\(|None\b)", text))
    return fields - {"await"}


def overlay_own_fields(text):
    """fields of the structs the overlay declares itself"""
    own = set()
    for body in re.findall(r"\bstruct\s+\w+(?:<[^>]*>)?\s*\{(.*?)\n\}", text, flags=re.S):
        own |= set(re.findall(r"(?:pub\s+)?([a-z_][a-z0-9_]*)\s*:", body))
    return own


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="/root/reference is not on this box (GPU box)")
def test_k8s_fields_are_the_reference_own_or_pinned():
    ref_text = "".join(open(os.path.join(base, f)).read() for base, _, files in os.walk(os.path.join(REFERENCE, "src")) for f in files if f.endswith(".rs"))
    srcs = rust_sources()
    used, own = set(), set()
    for name in ("src/ksched.rs", "src/predicates/device_parity.rs", "src/predicates/parity_dump.rs", "patches/0002-main-batch-task-and-pod-watch.patch"):
        text = srcs[name]
        if name.endswith(".patch"):
            text = "\n".join(ln[1:] for ln in text.splitlines() if ln.startswith("+") and not ln.startswith("+++"))
        used |= overlay_field_uses(strip_rust(text))
        own |= overlay_own_fields(strip_rust(text))
    own |= set(re.findall(r"^\+\s+pub (\w+):", srcs["patches/0004-context-ksched-fields.patch"], flags=re.M))  # the fields Context gains
    assert {"node_name", "node_selector", "containers", "requests", "resources", "allocatable", "labels", "name", "namespace", "resource_version"} <= used
    # what is neither a field of the overlay's own structs nor an identifier the reference's sources use must be in the pinned list
    unknown = sorted(f for f in used - own if not re.search(r"\b%s\b" % re.escape(f), ref_text) and f not in PINNED_API_FIELDS)
    assert unknown == [], f"fields neither the overlay's own, nor in the reference's sources, nor in the pinned list: {unknown}"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="/root/reference is not on this box (GPU box)")
def test_the_batched_pick_is_wired_into_the_running_binary(tmp_path):
    """VERDICT r2 "a binding that routes the running binary to the device": with --features ksched the Controller's reconcile gets its node
    from the batch task (ready_chunks -> ClusterState::pick_batch -> ksched_eval), `available` comes from the pod watch, and the POST
    below the pick is the reference's own text."""
    if not shutil.which("patch"):
        pytest.skip("no `patch` binary")
    out = tmp_path / "scheduler-v0"
    subprocess.check_call(["bash", os.path.join(ROOT, "rust", "apply.sh"), REFERENCE, str(out)], stdout=subprocess.DEVNULL)
    main_rs = (out / "src" / "main.rs").read_text()
    util_rs = (out / "src" / "util.rs").read_text()
    for f in ("src/main.rs", "src/util.rs", "src/predicates.rs", "src/ksched.rs", "src/ksched_sys.rs"):
        assert balanced((out / f).read_text()), f
    ref_main = open(os.path.join(REFERENCE, "src", "main.rs")).read()
    # both forms of select_node_for_pod exist, one per feature state; reconcile calls it with the reference's own expression
    assert '#[cfg(not(feature = "ksched"))]\nasync fn select_node_for_pod(pod: &corev1::Pod, ctx: &Context) -> Option<corev1::Node>' in main_rs
    assert '#[cfg(feature = "ksched")]\nasync fn select_node_for_pod(pod: &Arc<corev1::Pod>, ctx: &Context) -> Option<corev1::Node>' in main_rs
    assert strip_rust(main_rs).count("select_node_for_pod(&pod, &ctx).await") == 1 and "select_node_for_pod(&pod, &ctx).await" in ref_main
    # the batch task: ready_chunks -> one device call per batch -> replies; spawned from main(); the pod watch too
    for needle in ("work.ready_chunks(MAX_BATCH)", "state.pick_batch(&mut devices, &nodes_b, &pods_b, &draws, ATTEMPTS, want_rejected)", "tokio::task::spawn_blocking",
                   # the reference's WARN line for every rejected candidate (src/main.rs:62), asked of the device only when the level is on
                   "let want_rejected = tracing::enabled!(tracing::Level::WARN);",
                   'warn!("Node {} failed validity check for pod {}: {:?}", nodes[index].name_any(), full_name(pods[i].deref()), reason);',
                   # at most MAX_BATCH pods per device call, whatever piled up while the watch was syncing
                   "let rest = waiting.split_off(waiting.len().min(MAX_BATCH));",
                   "tokio::spawn(run_pick_batches(work, node_store.clone(), devices));",
                   "tokio::spawn(watch_bound_pods(client.clone(), picker.clone()));", "watcher::Event::Applied(pod)", "watcher::Event::Deleted(pod)",
                   "watcher::Event::Restarted(pods)", "ctx.picker.unbounded_send(Work::Pick(pod.clone(), reply))",
                   # the binding a reconcile has just POSTed is registered through the same channel (no lock shared with the async workers: ADVICE r3)
                   "ctx.picker.unbounded_send(Work::Event(ksched::ClusterEvent::Applied(landed)))",
                   # every device of $KSCHED_DEVICES, not a hard-coded device 0
                   "ksched::Devices::from_env()"):
        assert needle in main_rs, needle
    assert "Evaluator::new(0)" not in main_rs and "std::sync::Mutex" not in main_rs
    assert "picker: futures::channel::mpsc::UnboundedSender<crate::Work>" in util_rs and "Mutex" not in util_rs
    # no LIST per batch any more
    assert "Api::<corev1::Pod>::all" not in main_rs and ".list(" not in main_rs
    # the binding POST of reconcile (src/main.rs:83-103) is the reference's text, line for line
    post = ref_main[ref_main.index("        let pod_name = pod.name_any();"):ref_main.index("                match ctx.client.send(req_body).await {")]
    assert post in main_rs
    # what the overlay's ksched.rs offers is what main.rs calls
    ksched_rs = (out / "src" / "ksched.rs").read_text()
    for item in ("pub fn pick_batch(", "pub fn observe(", "pub fn resync(", "pub fn is_synced(", "pub fn counted_pods(", "pub enum PodEvent", "pub struct ClusterState",
                 "pub enum ClusterEvent", "pub fn apply(", "pub struct Devices", "pub fn from_env(", "pub fn pick_sampled("):
        assert item in ksched_rs, item
    # the reference's own format string, character for character
    assert 'warn!("Node {} failed validity check for pod {}: {:?}", candidate.name_any(), full_name(pod), e);' in ref_main
    # a pod with more selector keys than one device call takes is evaluated group by group, never refused (src/predicates.rs:48-53 has no limit)
    pick = ksched_rs[ksched_rs.index("pub fn pick_batch("):]
    assert "cannot be scheduled: {} nodeSelector keys" not in ksched_rs
    for needle in ("entries.chunks(sys::KSCHED_MAX_KEYS as usize)", "feasible[w] &= f[w];", "devices.pick_from_masks(1, &feasible, &req_mem, &samples, attempts)",
                   "devices.explain(&cols, &pair_pod, &pair_node)", "crate::predicates::reason_of(reasons[k])"):
        assert needle in pick, needle
    pred = (out / "src" / "predicates.rs").read_text()
    assert '#[cfg(feature = "ksched")]\npub fn reason_of(code: i32) -> Result<(), InvalidNodeReason>' in pred
    for call in re.findall(r"\bksched::(\w+)", main_rs):
        assert re.search(r"pub (?:struct|enum|fn|type) %s\b" % call, ksched_rs), call


def test_the_row_shard_over_several_devices_is_behind_the_rust_host():
    """the 8-GPU pod-row shard + RCCL all-gather reachable from the drop-in HOST.  `Devices` owns one Evaluator per
    device of $KSCHED_DEVICES and the communicator over them; a batch goes through ksched_eval_begin on every device, ONE
    ksched_allgather_bindings_local, ksched_eval_end -- the calls of include/ksched.h "one host thread, several devices", in that order --
    and the snapshot uploads are replicated."""
    text = strip_rust(rust_sources()["src/ksched.rs"])
    body = text[text.index("impl Devices {"):text.index("impl Drop for Devices")]
    order = [body.index(n) for n in ("sys::ksched_comm_create_local(", "sys::ksched_set_nodes(", "sys::ksched_update_nodes(", "sys::ksched_eval_begin(",
                                     "sys::ksched_gather_buffer(", "sys::ksched_allgather_bindings_local(", "sys::ksched_eval_end(")]
    assert order == sorted(order), order
    assert "sys::ksched_shard_bounds(" in text and "sys::ksched_device_count()" in text and "sys::ksched_comm_destroy(" in text
    assert 'std::env::var("KSCHED_DEVICES")' in rust_sources()["src/ksched.rs"]  # (strip_rust blanks string literals)
    # uploads go to every device: the snapshot code calls the replicated forms, never a single evaluator's entry point
    snap = text[text.index("impl Snapshot {"):text.index("pub struct ClusterState")]
    assert "devices.set_nodes(" in snap and "devices.update_nodes(" in snap and "sys::ksched_set_nodes(" not in snap and "sys::ksched_update_nodes(" not in snap
    # a failed upload never leaves bookkeeping that vouches for the devices (on_device after a failed ksched_set_nodes)
    up = snap[snap.index("pub fn encode_and_upload"):]
    assert up.index("self.on_device = None;") < up.index("self.encode_labels(&keys)?;") < up.index("devices.set_nodes(") < up.index("self.on_device = Some(wanted);")
    # the key budget: pick_batch walks consecutive ranges instead of failing the whole batch
    pick = text[text.index("pub fn pick_batch("):]
    assert "ranges.push((lo, j));" in pick and "sys::KSCHED_MAX_KEYS" in pick and "for (from, to) in ranges" in pick
