"""ctypes bridge to oracle/liboracle.so (oracle.c) -- TEST INFRASTRUCTURE ONLY.

Converts Kubernetes-shaped dicts into the C object model of oracle.h and exposes the object-level
and encoded-level restatements to tests and to bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "liboracle.so")

FIT, SEL, TAINT, PICK_SAMPLED, PICK_BESTFIT, WANT_FIT_MASK = 1, 2, 4, 8, 16, 32
E_PARSE, E_MISSING_KEY, E_RANGE = -1, -2, -3


class KV(C.Structure):
    _fields_ = [("key", C.c_char_p), ("val", C.c_char_p)]


class Container(C.Structure):
    _fields_ = [("has_resources", C.c_int), ("has_requests", C.c_int), ("cpu", C.c_char_p), ("memory", C.c_char_p)]


class Toleration(C.Structure):
    _fields_ = [("key", C.c_char_p), ("op", C.c_char_p), ("value", C.c_char_p), ("effect", C.c_char_p)]


class Taint(C.Structure):
    _fields_ = [("key", C.c_char_p), ("value", C.c_char_p), ("effect", C.c_char_p)]


class Pod(C.Structure):
    _fields_ = [("ns", C.c_char_p), ("name", C.c_char_p), ("has_spec", C.c_int), ("n_containers", C.c_uint32),
                ("containers", C.POINTER(Container)), ("has_node_selector", C.c_int), ("n_sel", C.c_uint32),
                ("sel", C.POINTER(KV)), ("node_name", C.c_char_p), ("n_tol", C.c_uint32), ("tol", C.POINTER(Toleration))]


class Node(C.Structure):
    _fields_ = [("name", C.c_char_p), ("has_labels", C.c_int), ("n_labels", C.c_uint32), ("labels", C.POINTER(KV)),
                ("has_status", C.c_int), ("has_allocatable", C.c_int), ("alloc_cpu", C.c_char_p),
                ("alloc_memory", C.c_char_p), ("n_taints", C.c_uint32), ("taints", C.POINTER(Taint))]


class Resources(C.Structure):
    # two __int128 as (lo, hi) pairs
    _fields_ = [("cpu_lo", C.c_uint64), ("cpu_hi", C.c_int64), ("mem_lo", C.c_uint64), ("mem_hi", C.c_int64)]
    _align_ = 16


_lib = None


def build(force: bool = False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_DIR, "oracle.c")):
        subprocess.check_call(["make", "-C", os.path.dirname(_DIR), "oracle"], stdout=subprocess.DEVNULL)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        vp, u32 = C.c_void_p, C.c_uint32
        lib.ora_parse_quantity.argtypes = [C.c_char_p, vp]
        lib.ora_q_to_milli.argtypes = [C.c_uint64, C.c_int64, C.POINTER(C.c_int64)]  # __int128 by value = (lo, hi) in SysV
        lib.ora_q_to_units.argtypes = [C.c_uint64, C.c_int64, C.POINTER(C.c_int64)]
        lib.ora_total_pod_resources.argtypes = [C.POINTER(Pod), C.POINTER(Resources)]
        lib.ora_can_pod_fit.argtypes = [C.POINTER(Pod), C.POINTER(Node), C.POINTER(C.POINTER(Pod)), u32, C.POINTER(C.c_int)]
        lib.ora_list_pods_on_node.argtypes = [C.POINTER(Pod), u32, C.c_char_p, C.POINTER(C.POINTER(Pod)), u32]
        lib.ora_list_pods_on_node.restype = u32
        lib.ora_does_node_selector_match.argtypes = [C.POINTER(Pod), C.POINTER(Node)]
        lib.ora_check_node_validity.argtypes = [C.POINTER(Pod), C.POINTER(Node), C.POINTER(C.POINTER(Pod)), u32]
        lib.ora_select_node_for_pod.argtypes = [C.POINTER(Pod), C.POINTER(Node), u32, C.POINTER(Pod), u32, vp, u32]
        lib.ora_tolerates_node_taints.argtypes = [C.POINTER(Pod), C.POINTER(Node)]
        lib.ora_eval_objects.argtypes = [C.POINTER(Pod), u32, C.POINTER(Node), u32, C.POINTER(Pod), u32, u32, vp, vp, C.c_int]
        lib.ora_eval_encoded.argtypes = [u32, vp, vp, vp, u32, vp, u32, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp, C.c_int]
        lib.ora_num_threads.restype = C.c_int
        _lib = lib
    return _lib


def _b(s: Optional[str]):
    return None if s is None else s.encode()


class ObjectSet:
    """Owns the C arrays (and the Python byte strings they point into) for a list of pods or nodes."""

    def __init__(self):
        self.keep: list = []

    def _kvs(self, d: dict):
        items = sorted(d.items())  # BTreeMap order
        arr = (KV * max(len(items), 1))()
        for i, (k, v) in enumerate(items):
            arr[i].key, arr[i].val = _b(k), _b(v)
        self.keep.append(arr)
        return arr, len(items)

    def pods(self, objs: Sequence[dict]):
        arr = (Pod * max(len(objs), 1))()
        for i, o in enumerate(objs):
            md = o.get("metadata") or {}
            p = arr[i]
            p.ns, p.name = _b(md.get("namespace")), _b(md.get("name"))
            spec = o.get("spec")
            p.has_spec = spec is not None
            if spec is None:
                continue
            conts = spec.get("containers") or []
            carr = (Container * max(len(conts), 1))()
            for j, c in enumerate(conts):
                res = c.get("resources")
                req = res.get("requests") if res is not None else None
                carr[j].has_resources = res is not None
                carr[j].has_requests = req is not None
                if req is not None:
                    carr[j].cpu, carr[j].memory = _b(req.get("cpu")), _b(req.get("memory"))
            self.keep.append(carr)
            p.n_containers, p.containers = len(conts), carr
            ns = spec.get("nodeSelector")
            p.has_node_selector = ns is not None
            if ns is not None:
                p.sel, p.n_sel = self._kvs(ns)
            p.node_name = _b(spec.get("nodeName"))
            tols = spec.get("tolerations") or []
            tarr = (Toleration * max(len(tols), 1))()
            for j, t in enumerate(tols):
                tarr[j].key, tarr[j].op, tarr[j].value, tarr[j].effect = _b(t.get("key")), _b(t.get("operator")), _b(t.get("value")), _b(t.get("effect"))
            self.keep.append(tarr)
            p.n_tol, p.tol = len(tols), tarr
        self.keep.append(arr)
        return arr

    def nodes(self, objs: Sequence[dict]):
        arr = (Node * max(len(objs), 1))()
        for i, o in enumerate(objs):
            md = o.get("metadata") or {}
            n = arr[i]
            n.name = _b(md.get("name", ""))
            labels = md.get("labels")
            n.has_labels = labels is not None
            if labels is not None:
                n.labels, n.n_labels = self._kvs(labels)
            status = o.get("status")
            n.has_status = status is not None
            alloc = status.get("allocatable") if status is not None else None
            n.has_allocatable = alloc is not None
            if alloc is not None:
                n.alloc_cpu, n.alloc_memory = _b(alloc.get("cpu")), _b(alloc.get("memory"))
            taints = (o.get("spec") or {}).get("taints") or []
            tarr = (Taint * max(len(taints), 1))()
            for j, t in enumerate(taints):
                tarr[j].key, tarr[j].value, tarr[j].effect = _b(t.get("key")), _b(t.get("value")), _b(t.get("effect"))
            self.keep.append(tarr)
            n.n_taints, n.taints = len(taints), tarr
        self.keep.append(arr)
        return arr


def parse_quantity(s: str) -> int:
    """-> exact nano-units as a Python int; raises ValueError with the ORA_E_* code."""
    buf = (C.c_uint64 * 2)()
    rc = load().ora_parse_quantity(s.encode(), C.cast(buf, C.c_void_p))
    if rc:
        raise ValueError(rc)
    v = buf[0] | (buf[1] << 64)
    return v - (1 << 128) if v >> 127 else v


def _i128(lo: int, hi: int) -> int:
    return (hi << 64) | lo


def total_pod_resources(pod: dict):
    s = ObjectSet()
    r = Resources()
    rc = load().ora_total_pod_resources(s.pods([pod]), C.byref(r))
    if rc:
        raise ValueError(rc)
    return _i128(r.cpu_lo, r.cpu_hi), _i128(r.mem_lo, r.mem_hi)


def does_node_selector_match(pod: dict, node: dict) -> bool:
    s = ObjectSet()
    return bool(load().ora_does_node_selector_match(s.pods([pod]), s.nodes([node])))


def tolerates_node_taints(pod: dict, node: dict) -> bool:
    s = ObjectSet()
    return bool(load().ora_tolerates_node_taints(s.pods([pod]), s.nodes([node])))


def check_node_validity(pod: dict, node: dict, all_pods: Sequence[dict]) -> int:
    """Does the LIST (ora_list_pods_on_node) then ora_check_node_validity; returns ORA_REASON_* or a negative panic code."""
    lib = load()
    s = ObjectSet()
    cp, cn, call = s.pods([pod]), s.nodes([node]), s.pods(all_pods)
    n_all = len(all_pods)
    cnt = lib.ora_list_pods_on_node(call, n_all, cn[0].name, None, 0)
    on = (C.POINTER(Pod) * max(cnt, 1))()
    lib.ora_list_pods_on_node(call, n_all, cn[0].name, on, cnt)
    return lib.ora_check_node_validity(cp, cn, on, cnt)


def select_node_for_pod(pod: dict, nodes: Sequence[dict], all_pods: Sequence[dict], samples: Sequence[int]) -> int:
    s = ObjectSet()
    smp = np.asarray(samples, dtype=np.uint32)
    return load().ora_select_node_for_pod(s.pods([pod]), s.nodes(nodes), len(nodes), s.pods(all_pods), len(all_pods),
                                          smp.ctypes.data_as(C.c_void_p), len(smp))


def eval_objects(pods: Sequence[dict], nodes: Sequence[dict], bound: Sequence[dict], flags: int, want_fit=False, threads=0,
                 prebuilt=None):
    """-> (feasible [P,W] uint64, fit or None).  `prebuilt` = (ObjectSet, cpods, cnodes, cbound) to reuse C arrays."""
    if prebuilt is None:
        s = ObjectSet()
        prebuilt = (s, s.pods(pods), s.nodes(nodes), s.pods(bound))
    _, cp, cn, cb = prebuilt
    P, N = len(pods), len(nodes)
    W = (N + 63) // 64
    feas = np.zeros((P, W), dtype=np.uint64)
    fit = np.zeros((P, W), dtype=np.uint64) if want_fit else None
    rc = load().ora_eval_objects(cp, P, cn, N, cb, len(bound), flags, feas.ctypes.data_as(C.c_void_p),
                                 fit.ctypes.data_as(C.c_void_p) if want_fit else None, threads)
    if rc:
        raise ValueError(rc)
    return feas, fit


def eval_encoded(avail_cpu, avail_mem, label_ids, taints, req_cpu, req_mem, sel_ids, tolerations, samples, flags: int,
                 want_mask=True, threads=0):
    """Encoded-level restatement; same contract as ksched_eval.  -> (feasible, fit, binding)"""
    def arr(a, dt):
        return None if a is None else np.ascontiguousarray(a, dtype=dt)

    def ptr(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    ac, am = arr(avail_cpu, np.int64), arr(avail_mem, np.int64)
    lab, tnt = arr(label_ids, np.uint32), arr(taints, np.uint64)
    rc_, rm = arr(req_cpu, np.int64), arr(req_mem, np.int64)
    sel, tol, smp = arr(sel_ids, np.uint32), arr(tolerations, np.uint64), arr(samples, np.uint32)
    n, p = ac.shape[0], rc_.shape[0]
    n_keys = 0 if lab is None else lab.shape[0]
    W = (n + 63) // 64
    feas = np.zeros((p, W), dtype=np.uint64) if want_mask else None
    fit = np.zeros((p, W), dtype=np.uint64) if flags & WANT_FIT_MASK else None
    binding = np.full((p,), -1, dtype=np.int32) if flags & (PICK_SAMPLED | PICK_BESTFIT) else None
    attempts = smp.shape[1] if (smp is not None and flags & PICK_SAMPLED) else 0
    r = load().ora_eval_encoded(n, ptr(ac), ptr(am), ptr(lab), n_keys, ptr(tnt), p, ptr(rc_), ptr(rm), ptr(sel), ptr(tol),
                                ptr(smp), attempts, flags, ptr(feas), ptr(fit), ptr(binding), threads)
    if r:
        raise ValueError(r)
    return feas, fit, binding


def num_threads() -> int:
    return load().ora_num_threads()
