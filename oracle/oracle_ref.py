"""Pure-Python, exact restatement of the reference's predicate path on Kubernetes-shaped dicts.

TEST INFRASTRUCTURE ONLY -- same rules as oracle/oracle.h: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this; the product never does.

Follows acrlabs/kube-scheduler-rs-reference src/predicates.rs:14-77, src/util.rs:17-36,54-75 and
src/main.rs:49-71 line by line (citations on each function).  Arithmetic is `fractions.Fraction`
(arbitrary precision), so it doubles as an independent check of oracle.c's 128-bit integers.

Pinning: `does_node_selector_match` is pinned by the reference's tests
(src/predicates/test.rs:42-58).  The resource-fit arithmetic (kube_quantity 0.6.1, un-vendored)
has no reference test: parity unpinned there (see oracle.h).

Objects are plain dicts in the API server's JSON shape; an absent key is Rust's `None`.
"""
from __future__ import annotations

import re
from fractions import Fraction
from typing import Iterable, List, Optional, Sequence

ATTEMPTS = 5  # src/main.rs:49


class ReferencePanic(Exception):
    """A situation where the reference would panic (expect / unwrap / map index)."""


# ---- quantities (role of kube_quantity::ParsedQuantity) ------------------------------------------
_DEC = {"n": -9, "u": -6, "m": -3, "": 0, "k": 3, "M": 6, "G": 9, "T": 12, "P": 15, "E": 18}
_BIN = {"Ki": 10, "Mi": 20, "Gi": 30, "Ti": 40, "Pi": 50, "Ei": 60}
_QRE = re.compile(r"^([+-]?)(\d*)(?:\.(\d*))?((?:[eE][+-]?\d+)|[KMGTPE]i|[numkMGTPE]?)$")


def parse_quantity(s: str) -> Fraction:
    """Kubernetes resource.Quantity text -> exact value (cores, bytes)."""
    if not isinstance(s, str):
        raise ReferencePanic(f"quantity is not a string: {s!r}")
    m = _QRE.match(s)
    if not m or not (m.group(2) or m.group(3)):
        raise ReferencePanic(f"invalid quantity {s!r}")  # try_into().expect(...), src/util.rs:65,68
    sign, ip, fp, suf = m.group(1), m.group(2) or "0", m.group(3) or "", m.group(4)
    val = Fraction(int(ip + fp), 10 ** len(fp))
    if suf in _BIN:
        val *= 2 ** _BIN[suf]
    elif suf and suf[0] in "eE" and len(suf) > 1:
        val *= Fraction(10) ** int(suf[1:])
    else:
        val *= Fraction(10) ** _DEC[suf]
    return -val if sign == "-" else val


class PodResources:
    """src/util.rs:17-36"""

    def __init__(self):  # PodResources::new, src/util.rs:22-29
        self.cpu = parse_quantity("0")
        self.memory = parse_quantity("0")

    def sub_assign(self, other: "PodResources"):  # src/util.rs:31-36
        self.cpu -= other.cpu
        self.memory -= other.memory


def total_pod_resources(pod: dict) -> PodResources:
    """src/util.rs:54-75"""
    res = PodResources()  # :55
    spec = pod.get("spec")
    if spec is not None:  # :57
        for c in spec.get("containers") or []:  # :58 -- containers only: no initContainers, no overhead
            resources = c.get("resources")
            requests = resources.get("requests") if resources is not None else None
            if requests is not None:  # :59-63
                if "cpu" in requests:  # :64
                    res.cpu += parse_quantity(requests["cpu"])  # :65
                if "memory" in requests:  # :67
                    res.memory += parse_quantity(requests["memory"])  # :68
    return res


def list_pods_on_node(all_pods: Iterable[dict], node_name: str) -> List[dict]:
    """src/predicates.rs:21-25,34: LIST with field selector spec.nodeName=<node>; every phase counts."""
    return [p for p in all_pods if (p.get("spec") or {}).get("nodeName") == node_name]


def node_name(node: dict) -> str:
    return (node.get("metadata") or {}).get("name", "")


def can_pod_fit(pod: dict, node: dict, pods_on_node: Sequence[dict]) -> bool:
    """src/predicates.rs:20-43 with the LIST result of :34 injected."""
    available = PodResources()  # :27
    status = node.get("status")
    allocatable = status.get("allocatable") if status is not None else None
    if allocatable is not None:  # :28
        if "cpu" not in allocatable or "memory" not in allocatable:
            raise ReferencePanic("allocatable lacks cpu/memory (BTreeMap index panics, src/predicates.rs:29-31)")
        available.cpu = parse_quantity(allocatable["cpu"])  # :29
        available.memory = parse_quantity(allocatable["memory"])  # :30-31
    for p in pods_on_node:  # :36
        available.sub_assign(total_pod_resources(p))  # :37
    pod_requests = total_pod_resources(pod)  # :40
    return pod_requests.cpu <= available.cpu and pod_requests.memory <= available.memory  # :42


def does_node_selector_match(pod: dict, node: dict) -> bool:
    """src/predicates.rs:45-61"""
    matches = True  # :46
    spec = pod.get("spec")
    node_selector = spec.get("nodeSelector") if spec is not None else None
    if node_selector is not None:  # :47
        for pk in sorted(node_selector):  # :48 BTreeMap iteration order
            pv = node_selector[pk]
            labels = (node.get("metadata") or {}).get("labels")
            if labels is not None:  # :49
                if pk not in labels or labels[pk] != pv:  # :50  labels.get(pk) != Some(pv)
                    matches = False
                    break
            else:  # :54-57
                matches = False
                break
    return matches  # :60


def check_node_validity(pod: dict, node: dict, pods_on_node: Sequence[dict]) -> Optional[str]:
    """src/predicates.rs:63-77.  None = Ok(()); otherwise the Debug name of InvalidNodeReason."""
    if not can_pod_fit(pod, node, pods_on_node):  # :68
        return "NotEnoughResources"  # :69
    if not does_node_selector_match(pod, node):  # :72
        return "NodeSelectorMismatch"  # :73
    return None  # :76


def select_node_for_pod(pod: dict, nodes: Sequence[dict], all_pods: Sequence[dict], samples: Sequence[int],
                        attempts: int = ATTEMPTS) -> Optional[int]:
    """src/main.rs:51-71 with the `choose` draws injected; returns the node index or None."""
    chosen = None  # :52
    for i in range(attempts):  # :53
        if len(nodes) == 0:  # :56 choose() on an empty slice
            continue
        s = samples[i]
        if s >= len(nodes):  # not reachable in the reference; the ABI treats it as an infeasible draw
            continue
        candidate = nodes[s]  # :57
        if check_node_validity(pod, candidate, list_pods_on_node(all_pods, node_name(candidate))) is None:  # :61
            chosen = s  # :64
            break  # :65
    return chosen  # :70


# ---- extensions (BASELINE.json config 5; semantics in DESIGN.md) -------------------------------------
def _toleration_matches(t: dict, taint: dict) -> bool:
    if t.get("effect") and t.get("effect") != taint.get("effect", ""):
        return False
    op = t.get("operator") or "Equal"
    if not t.get("key"):
        return op == "Exists"
    if t.get("key") != taint.get("key", ""):
        return False
    if op == "Exists":
        return True
    return op == "Equal" and (t.get("value") or "") == (taint.get("value") or "")


def tolerates_node_taints(pod: dict, node: dict) -> bool:
    tols = (pod.get("spec") or {}).get("tolerations") or []
    for taint in (node.get("spec") or {}).get("taints") or []:
        if taint.get("effect") not in ("NoSchedule", "NoExecute"):
            continue
        if not any(_toleration_matches(t, taint) for t in tols):
            return False
    return True


def available_of(node: dict, all_pods: Sequence[dict]) -> PodResources:
    """allocatable - sum(requests of the node's LIST), the left side of src/predicates.rs:42."""
    available = PodResources()
    status = node.get("status")
    allocatable = status.get("allocatable") if status is not None else None
    if allocatable is not None:
        if "cpu" not in allocatable or "memory" not in allocatable:
            raise ReferencePanic("allocatable lacks cpu/memory")
        available.cpu = parse_quantity(allocatable["cpu"])
        available.memory = parse_quantity(allocatable["memory"])
    for p in list_pods_on_node(all_pods, node_name(node)):
        available.sub_assign(total_pod_resources(p))
    return available


def pick_bestfit(pod: dict, nodes: Sequence[dict], all_pods: Sequence[dict], use_fit=True, use_sel=True,
                 use_taint=False) -> Optional[int]:
    """Extension E1: lexicographic min over feasible nodes of (mem residual, cpu residual, node index)."""
    req = total_pod_resources(pod)
    best = None
    for i, node in enumerate(nodes):
        on = list_pods_on_node(all_pods, node_name(node))
        if use_fit and not can_pod_fit(pod, node, on):
            continue
        if use_sel and not does_node_selector_match(pod, node):
            continue
        if use_taint and not tolerates_node_taints(pod, node):
            continue
        av = available_of(node, all_pods)
        key = (av.memory - req.memory, av.cpu - req.cpu, i)
        if best is None or key < best:
            best = key
    return None if best is None else best[2]


def eval_matrix(pods: Sequence[dict], nodes: Sequence[dict], all_pods: Sequence[dict], use_fit=True, use_sel=True,
                use_taint=False, cache=False):
    """feasible[p][n], fit[p][n] as nested lists of bool, one per-pair reference evaluation each.

    cache=True evaluates the same per-pair predicate with the two pure sub-results memoised -- the left side of
    src/predicates.rs:42 once per node (`available_of`) and `total_pod_resources` once per pod -- instead of re-parsing
    every quantity for every pair.  Same functions, same Fractions, same answers; it only makes clusters of 10^6 pairs
    finish in seconds."""
    lists = [list_pods_on_node(all_pods, node_name(n)) for n in nodes]
    avail = [available_of(n, all_pods) for n in nodes] if (cache and use_fit) else None
    feas, fits = [], []
    for pod in pods:
        frow, rrow = [], []
        req = total_pod_resources(pod) if avail is not None else None
        for j, (node, on) in enumerate(zip(nodes, lists)):
            if not use_fit:
                fit = True
            elif avail is not None:
                fit = req.cpu <= avail[j].cpu and req.memory <= avail[j].memory  # src/predicates.rs:42
            else:
                fit = can_pod_fit(pod, node, on)
            ok = fit
            if ok and use_sel:
                ok = does_node_selector_match(pod, node)
            if ok and use_taint:
                ok = tolerates_node_taints(pod, node)
            frow.append(ok)
            rrow.append(fit)
        feas.append(frow)
        fits.append(rrow)
    return feas, fits


# ---- the callers either side of the pick ---------------------------------------
class SplitMixChooser:
    """The injected stand-in for `SliceRandom::choose(&mut thread_rng())` (src/main.rs:56): a SplitMix64 stream, the same
    one the host mirror's SplitMixChooser and the generator use.  choose(n) -> index in [0, n) or None for an empty slice."""
    M = (1 << 64) - 1

    def __init__(self, seed: int):
        self.state = seed & self.M

    def choose(self, n: int) -> Optional[int]:
        if n == 0:
            return None
        self.state = (self.state + 0x9E3779B97F4A7C15) & self.M
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.M
        z ^= z >> 31
        return z % n


def is_pod_bound(pod: dict) -> bool:
    """src/util.rs:38-45"""
    spec = pod.get("spec")
    return spec is not None and spec.get("nodeName") is not None


def _draws(chooser, n_store: int, attempts: int) -> List[Optional[int]]:
    return [chooser.choose(n_store) for _ in range(attempts)]


def _select_with_draws(pod: dict, store: Sequence[dict], all_pods: Sequence[dict], draws: Sequence[Optional[int]]) -> Optional[int]:
    """src/main.rs:51-71 on the store's own ordering: first drawn candidate whose check_node_validity is Ok(())."""
    for s in draws:  # :53
        if s is None:  # :56 empty store
            continue
        candidate = store[s]  # :57
        if check_node_validity(pod, candidate, list_pods_on_node(all_pods, node_name(candidate))) is None:  # :61
            return s  # :64-65
    return None  # :70


def _post(outcomes, i, pod, node, sink_state):
    """src/main.rs:94-108: the binding POST; sink_state = [calls, fail_every, posted]."""
    sink_state[0] += 1
    if sink_state[1] and sink_state[0] % sink_state[1] == 0:
        outcomes[i] = {"ok": False, "error": "create-binding-failed", "bound_to": None}  # :105-108
        return False
    md = pod.get("metadata") or {}
    sink_state[2].append((f"{md.get('namespace')}/{md.get('name')}", node_name(node)))
    outcomes[i] = {"ok": True, "error": None, "bound_to": node_name(node)}  # :119
    return True


def reconcile_batch(pods: Sequence[dict], store: Sequence[dict], all_pods: Sequence[dict], chooser, fail_every: int = 0,
                    attempts: int = ATTEMPTS):
    """The batching reconciler (8f n2) as a legal execution of the reference: every pending pod is reconciled
    (src/main.rs:73-120) against the SAME API-server state -- the reference has no assume/reserve step (:78-119), so
    reconciles racing on one state all see it unchanged.  Bound pods return Ok at once (:74-76).  Each pending pod, in batch
    order, takes ATTEMPTS draws from the chooser (all of them: the draws after its first success cannot change its outcome);
    the POSTs then go out in batch order.  -> (outcomes, posted)"""
    outcomes = [None] * len(pods)
    pending = [i for i, p in enumerate(pods) if not is_pod_bound(p)]
    for i in range(len(pods)):
        if i not in pending:
            outcomes[i] = {"ok": True, "error": None, "bound_to": None}  # :74-76
    picks = [_select_with_draws(pods[i], store, all_pods, _draws(chooser, len(store), attempts)) for i in pending]
    sink = [0, fail_every, []]
    for i, s in zip(pending, picks):
        if s is None:
            outcomes[i] = {"ok": False, "error": "no-node-found", "bound_to": None}  # :116-118
        else:
            _post(outcomes, i, pods[i], store[s], sink)
    return outcomes, sink[2]


def reconcile_batch_sequential(pods: Sequence[dict], store: Sequence[dict], all_pods: Sequence[dict], chooser, fail_every: int = 0,
                               max_rounds: int = 64, attempts: int = ATTEMPTS):
    """In-batch capacity accounting (8f n3; NOT reference behaviour -- builder-defined): rounds.  In a round
    every still-pending pod is picked against the state at the START of the round (fresh ATTEMPTS draws each); per node only the
    first pod of the round (batch order) is accepted and POSTed, its requests then count against the node (it joins the pods
    the LIST returns); later pods that drew the same node go to the next round.  A pod with no feasible draw gets NoNodeFound;
    so do pods still colliding after max_rounds.  -> (outcomes, posted, rounds, conflicts, final list of bound pods)"""
    outcomes = [None] * len(pods)
    pending = []
    for i, p in enumerate(pods):
        if is_pod_bound(p):
            outcomes[i] = {"ok": True, "error": None, "bound_to": None}
        else:
            pending.append(i)
    state = list(all_pods)
    sink = [0, fail_every, []]
    rounds = conflicts = 0
    while pending and rounds < max_rounds:
        rounds += 1
        picks = [_select_with_draws(pods[i], store, state, _draws(chooser, len(store), attempts)) for i in pending]
        taken, nxt, landed = set(), [], []
        for i, s in zip(pending, picks):
            if s is None:
                outcomes[i] = {"ok": False, "error": "no-node-found", "bound_to": None}
            elif s in taken:
                conflicts += 1
                nxt.append(i)
            elif _post(outcomes, i, pods[i], store[s], sink):
                taken.add(s)
                p = dict(pods[i])
                p["spec"] = dict(p.get("spec") or {})
                p["spec"]["nodeName"] = node_name(store[s])
                landed.append(p)
        state += landed
        pending = nxt
    for i in pending:
        outcomes[i] = {"ok": False, "error": "no-node-found", "bound_to": None}
    return outcomes, sink[2], rounds, conflicts, state


# ---- second reading: kube_quantity 0.6.1 AS RECALLED (SURVEY.md section 8c hazard list) ----------------------------------------------
#
# NOT a restatement of code on disk: the crate is an un-vendored dependency (Cargo.lock:787-797) and nothing here can build or run it.
# This class writes down what the surveyor RECALLS of its internals, so that the unpinned half of the parity claim is BRACKETED: for
# every quantity spelling, tests/test_quantity_readings.py states whether exact Kubernetes semantics (parse_quantity above: what the
# product and both oracles implement) and this reading agree.  Where they agree, a run of the real reference (rust/pin_parity.sh)
# cannot tell them apart and the parity claim holds under either; where they differ, the committed fixtures carry BOTH expectations
# and the day someone runs the reference the answer is a one-line diff.
#
# The recollection (crate `kube_quantity`, struct ParsedQuantity { value: rust_decimal::Decimal, scale, format }):
#   * parse: sign? digits [. digits] suffix; suffix in  n u m "" k M G T P E  -> DecimalSI, Ki Mi Gi Ti Pi Ei -> BinarySI, scale index
#     n=-3 u=-2 m=-1 ""=0 k/Ki=1 M/Mi=2 G/Gi=3 T/Ti=4 P/Pi=5 E/Ei=6.  Exponent forms (1e3, 1E3) are NOT accepted by 0.6.1.
#   * a += b, a -= b, a <= b all first bring b to a's FORMAT -- b.value *= Decimal::from_f32((1024/1000)^b.scale) (DecimalSI <- BinarySI) or
#     (1000/1024)^b.scale (BinarySI <- DecimalSI), the power computed in f32 -- then both to the SMALLER scale, multiplying the larger-scale
#     side by Decimal::from_f32(base^diff), base 1000 or 1024 by that side's format, again in f32; the result keeps a's format.
#   * Decimal::from_f32 keeps 7 significant decimal digits (rust_decimal's FromPrimitive for f32).
# Consequences: Ki (1.024) and Mi (1.048576) factors survive f32 + 7 digits exactly; Gi (1.073741824 -> 1.073742), Ti, Pi, Ei do not,
# and neither do decimal scale steps of 10^12 and beyond.  Because PodResources::new() seeds "0" (DecimalSI, src/util.rs:25-26), every
# BinarySI REQUEST is converted on its way into the sum; a node's allocatable is ASSIGNED (src/predicates.rs:29-31) and keeps its own
# format, so `available` may be BinarySI and the pods subtracted from it are converted the other way.
import decimal as _decimal
import struct as _struct

_KQ_CTX = _decimal.Context(prec=60)  # (rust_decimal: 96-bit mantissa, 28 fractional digits; nothing here comes near either limit)
_KQ_SCALE = {"n": -3, "u": -2, "m": -1, "": 0, "k": 1, "M": 2, "G": 3, "T": 4, "P": 5, "E": 6, "Ki": 1, "Mi": 2, "Gi": 3, "Ti": 4, "Pi": 5, "Ei": 6}
_KQ_RE = re.compile(r"^([+-]?)(\d+)(?:\.(\d+))?(Ki|Mi|Gi|Ti|Pi|Ei|[numkMGTPE]?)$")


def _f32(x: float) -> float:
    return _struct.unpack("f", _struct.pack("f", x))[0]


def _powi_f32(base: float, n: int) -> float:
    """f32::powi: repeated multiplication in f32 (what llvm.powi lowers to for small integer exponents); negative n = 1 / powi(-n)."""
    r = _f32(1.0)
    b = _f32(base)
    for _ in range(abs(n)):
        r = _f32(r * b)
    return _f32(1.0 / r) if n < 0 else r


def _decimal_from_f32(x: float) -> "_decimal.Decimal":
    """rust_decimal Decimal::from_f32: the f32's value rounded to 7 significant decimal digits."""
    if x == 0:
        return _decimal.Decimal(0)
    return _KQ_CTX.create_decimal(_decimal.Context(prec=7, rounding=_decimal.ROUND_HALF_EVEN).create_decimal(repr(float(x))))


class KubeQuantity061:
    """One ParsedQuantity as recalled: (value, scale index, binary format?)."""

    __slots__ = ("value", "scale", "binary")

    def __init__(self, value, scale: int, binary: bool):
        self.value, self.scale, self.binary = _decimal.Decimal(value), scale, binary

    @staticmethod
    def parse(s: str) -> "KubeQuantity061":
        if not isinstance(s, str):
            raise ReferencePanic(f"quantity is not a string: {s!r}")
        m = _KQ_RE.match(s)
        if not m:
            raise ReferencePanic(f"kube_quantity 0.6.1 (as recalled) does not parse {s!r}")  # try_into().expect(...)
        sign, ip, fp, suf = m.group(1), m.group(2), m.group(3) or "", m.group(4)
        v = _decimal.Decimal((1 if sign == "-" else 0, tuple(int(c) for c in (ip + fp).lstrip("0") or "0"), -len(fp)))
        return KubeQuantity061(v, _KQ_SCALE[suf], suf.endswith("i"))

    def copy(self) -> "KubeQuantity061":
        return KubeQuantity061(self.value, self.scale, self.binary)

    # -- the two normalisations every operator runs on (clones of) its operands
    @staticmethod
    def _normalize_formats(lhs: "KubeQuantity061", rhs: "KubeQuantity061"):
        if lhs.binary == rhs.binary:
            return
        ratio = _f32(1000.0) / _f32(1024.0) if lhs.binary else _f32(1024.0) / _f32(1000.0)
        rhs.value = _KQ_CTX.multiply(rhs.value, _decimal_from_f32(_powi_f32(_f32(ratio), rhs.scale)))
        rhs.binary = lhs.binary

    @staticmethod
    def _normalize_scales(lhs: "KubeQuantity061", rhs: "KubeQuantity061"):
        if lhs.scale == rhs.scale:
            return
        big, small = (rhs, lhs) if rhs.scale > lhs.scale else (lhs, rhs)
        base = 1024.0 if big.binary else 1000.0
        big.value = _KQ_CTX.multiply(big.value, _decimal_from_f32(_powi_f32(base, big.scale - small.scale)))
        big.scale = small.scale

    def _with(self, other: "KubeQuantity061"):
        lhs, rhs = self.copy(), other.copy()
        KubeQuantity061._normalize_formats(lhs, rhs)
        KubeQuantity061._normalize_scales(lhs, rhs)
        return lhs, rhs

    def add_assign(self, other: "KubeQuantity061"):
        lhs, rhs = self._with(other)
        self.value, self.scale, self.binary = _KQ_CTX.add(lhs.value, rhs.value), lhs.scale, lhs.binary

    def sub_assign(self, other: "KubeQuantity061"):
        lhs, rhs = self._with(other)
        self.value, self.scale, self.binary = _KQ_CTX.subtract(lhs.value, rhs.value), lhs.scale, lhs.binary

    def le(self, other: "KubeQuantity061") -> bool:
        lhs, rhs = self._with(other)
        return lhs.value <= rhs.value

    def in_units(self) -> Fraction:
        """the quantity's value in cores / bytes under THIS reading's own conversion to scale "" (for printing and comparing)"""
        probe = KubeQuantity061(0, 0, False)
        probe.add_assign(self)  # = how a request enters the accumulator seeded "0" (src/util.rs:25-26,65,68)
        return Fraction(str(probe.value)) * Fraction(1000) ** probe.scale


def kq061_total_pod_resources(pod: dict):
    """src/util.rs:54-75 under the recalled arithmetic: two accumulators seeded "0" (DecimalSI)."""
    cpu, mem = KubeQuantity061.parse("0"), KubeQuantity061.parse("0")
    spec = pod.get("spec")
    if spec is not None:
        for c in spec.get("containers") or []:
            resources = c.get("resources")
            requests = resources.get("requests") if resources is not None else None
            if requests is not None:
                if "cpu" in requests:
                    cpu.add_assign(KubeQuantity061.parse(requests["cpu"]))
                if "memory" in requests:
                    mem.add_assign(KubeQuantity061.parse(requests["memory"]))
    return cpu, mem


def kq061_can_pod_fit(pod: dict, node: dict, pods_on_node: Sequence[dict]) -> bool:
    """src/predicates.rs:20-43 under the recalled arithmetic (allocatable ASSIGNED, so `available` keeps the node's format)."""
    acpu, amem = KubeQuantity061.parse("0"), KubeQuantity061.parse("0")
    status = node.get("status")
    allocatable = status.get("allocatable") if status is not None else None
    if allocatable is not None:
        if "cpu" not in allocatable or "memory" not in allocatable:
            raise ReferencePanic("allocatable lacks cpu/memory (BTreeMap index panics, src/predicates.rs:29-31)")
        acpu, amem = KubeQuantity061.parse(allocatable["cpu"]), KubeQuantity061.parse(allocatable["memory"])
    for p in pods_on_node:
        c, m = kq061_total_pod_resources(p)
        acpu.sub_assign(c)
        amem.sub_assign(m)
    rc, rm = kq061_total_pod_resources(pod)
    return rc.le(acpu) and rm.le(amem)


def kq061_fit_matrix(pods: Sequence[dict], nodes: Sequence[dict], all_pods: Sequence[dict]):
    """fit bit of every (pod, node) pair under the recalled reading; a pair on which it would panic (an unparsable spelling) is None."""
    lists = [list_pods_on_node(all_pods, node_name(n)) for n in nodes]
    out = []
    for p in pods:
        for i, n in enumerate(nodes):
            try:
                out.append(kq061_can_pod_fit(p, n, lists[i]))
            except ReferencePanic:
                out.append(None)
    return out
