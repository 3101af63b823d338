"""The Rust overlay (rust/) cannot be compiled here (no cargo / rustc).  What can be checked without a toolchain:
  * src/ksched_sys.rs declares exactly the functions include/ksched.h declares (and nothing else), with as many arguments;
  * its constants carry the header's values;
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
