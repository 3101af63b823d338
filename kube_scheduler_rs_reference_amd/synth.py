"""Seeded synthetic pod/node sets -- the single source of inputs for the
oracle, the CPU baseline and the GPU path.

The generator works on primitive integer columns (numpy, SplitMix64 counter streams; never
`thread_rng`-style global state) and can materialise two views of the SAME cluster:

  * `columns`  -- the encoded SoA columns the C ABI takes (milli-CPU / bytes as int64, label value
                  ids, taint bit sets), for any size up to millions of pods;
  * `objects`  -- Kubernetes-shaped dicts (Pod / Node as the API server would serve them, quantity
                  strings and label strings), for the object-level oracle and the host encoder.
                  Meant for small and medium cases.

Quantities stay inside the canonical domain D of SURVEY.md section 8c: CPU as "<n>m" or integer
cores, memory as plain integer byte strings (optionally Ki/Mi when `binary_suffixes=True`).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
SEL_NEVER = 0xFFFFFFFF
KEY_CARDINALITY = (2, 3, 4, 8, 16, 32, 64, 128,  # keys k0..k7
                   3, 5, 2, 7, 4, 9, 6, 11)       # keys k8..k15 (clusters with more than eight label keys; tests)
NODE_CORES = (4, 8, 16, 32, 64, 96, 128)
GIB_PER_CORE = (2, 4, 8)
MIB = 1 << 20
GIB = 1 << 30


def _mix(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def stream(seed: int, stream_id: int, n: int, start: int = 0) -> np.ndarray:
    """outputs [start, start + n) of the counter stream (seed, stream_id): SplitMix64 of a counter, so any window of a stream can be made
    without the outputs before it"""
    with np.errstate(over="ignore"):
        base = _mix(np.array([np.uint64(seed & 0xFFFFFFFFFFFFFFFF)]) * GOLDEN + np.uint64(stream_id))[0]
        ctr = (np.arange(start + 1, start + n + 1, dtype=np.uint64) * GOLDEN) + base
    return _mix(ctr)


def _below(x: np.ndarray, m: int) -> np.ndarray:
    return (x % np.uint64(m)).astype(np.int64)


def _bern(x: np.ndarray, p: float) -> np.ndarray:
    return (x >> np.uint64(11)) < np.uint64(int(p * (1 << 53)))


@dataclass
class Cluster:
    seed: int
    P: int
    N: int
    n_keys: int
    n_taints: int
    attempts: int
    # node columns
    alloc_cpu_milli: np.ndarray = None   # [N] int64
    alloc_mem_bytes: np.ndarray = None   # [N] int64
    bound_cnt: np.ndarray = None         # [N] pods already bound to the node (1..3)
    bound_cpu: np.ndarray = None         # [N,3] int64 requests of those pods (0 beyond bound_cnt)
    bound_mem: np.ndarray = None         # [N,3]
    avail_cpu: np.ndarray = None         # [N] = alloc - sum(bound)   (may be negative)
    avail_mem: np.ndarray = None
    node_labels: np.ndarray = None       # [n_keys, N] uint32 value ids, 0 = absent
    node_taints: np.ndarray = None       # [N] uint64 bit sets
    # pod columns
    pod_ncont: np.ndarray = None         # [P] 1..3 containers
    pod_has_req: np.ndarray = None       # [P] bool: containers carry resources.requests
    cont_cpu: np.ndarray = None          # [P,3] int64 per-container milli-CPU (0 beyond ncont)
    cont_mem: np.ndarray = None          # [P,3]
    req_cpu: np.ndarray = None           # [P] totals
    req_mem: np.ndarray = None
    pod_sel: np.ndarray = None           # [n_keys, P] uint32, 0 = unconstrained, SEL_NEVER = outside the domain
    pod_tol: np.ndarray = None           # [P] uint64
    samples: np.ndarray = None           # [P, attempts] uint32
    binary_suffixes: bool = False

    # ---- encoded view -----------------------------------------------------------------------
    def node_columns(self):
        return dict(avail_cpu_milli=self.avail_cpu, avail_mem_bytes=self.avail_mem,
                    label_val_ids=self.node_labels if self.n_keys else None,
                    taints=self.node_taints if self.n_taints else None)

    def pod_columns(self, lo: int = 0, hi: Optional[int] = None):
        hi = self.P if hi is None else hi
        return dict(req_cpu_milli=self.req_cpu[lo:hi], req_mem_bytes=self.req_mem[lo:hi],
                    sel_val_ids=np.ascontiguousarray(self.pod_sel[:, lo:hi]) if self.n_keys else None,
                    tolerations=self.pod_tol[lo:hi] if self.n_taints else None,
                    samples=self.samples[lo:hi])

    # ---- object view ------------------------------------------------------------------------
    @staticmethod
    def node_name(i: int) -> str:
        return f"node-{i:06d}"

    @staticmethod
    def label_key(k: int) -> str:
        return f"k{k}"

    @staticmethod
    def label_val(v: int) -> str:
        return f"v{v}"

    @staticmethod
    def taint_key(t: int) -> str:
        return f"t{t}"

    def _cpu_str(self, milli: int, salt: int) -> str:
        # both DecimalSI spellings of the canonical domain: "<n>m" and integer cores
        if milli % 1000 == 0 and (salt & 1):
            return str(milli // 1000)
        return f"{milli}m"

    def _mem_str(self, b: int, salt: int) -> str:
        if self.binary_suffixes and b > 0:
            if b % MIB == 0 and (salt & 2):
                return f"{b // MIB}Mi"
            if b % 1024 == 0 and (salt & 4):
                return f"{b // 1024}Ki"
        return str(b)

    def node_objects(self) -> List[dict]:
        out = []
        for i in range(self.N):
            labels = {self.label_key(k): self.label_val(int(self.node_labels[k, i]))
                      for k in range(self.n_keys) if self.node_labels[k, i] != 0}
            node = {"metadata": {"name": self.node_name(i)},
                    "status": {"allocatable": {"cpu": self._cpu_str(int(self.alloc_cpu_milli[i]), i),
                                               "memory": self._mem_str(int(self.alloc_mem_bytes[i]), i),
                                               "pods": "110"}}}
            if labels or (i % 7):  # some label-less nodes carry `labels: None`, others an empty map
                node["metadata"]["labels"] = labels
            taints = [{"key": self.taint_key(t), "value": "true", "effect": "NoSchedule"}
                      for t in range(self.n_taints) if (int(self.node_taints[i]) >> t) & 1]
            if taints:
                node["spec"] = {"taints": taints}
            out.append(node)
        return out

    def bound_pod_objects(self) -> List[dict]:
        """The pods the API server would LIST per node (reference src/predicates.rs:22-25,34):
        every phase counts, including Succeeded ones."""
        out = []
        for i in range(self.N):
            for b in range(int(self.bound_cnt[i])):
                cpu, mem = int(self.bound_cpu[i, b]), int(self.bound_mem[i, b])
                # split each bound pod over two containers to exercise the sum (src/util.rs:58-69)
                c0, m0 = cpu // 2, mem // 2
                conts = [{"name": "a", "resources": {"requests": {"cpu": self._cpu_str(c0, i + b), "memory": self._mem_str(m0, b)}}},
                         {"name": "b", "resources": {"requests": {"cpu": self._cpu_str(cpu - c0, i), "memory": self._mem_str(mem - m0, i)}}}]
                out.append({"metadata": {"name": f"bound-{i:06d}-{b}", "namespace": "kube-system"},
                            "spec": {"nodeName": self.node_name(i), "containers": conts},
                            "status": {"phase": "Succeeded" if (i + b) % 5 == 0 else "Running"}})
        return out

    def pod_objects(self, lo: int = 0, hi: Optional[int] = None) -> List[dict]:
        hi = self.P if hi is None else hi
        out = []
        for p in range(lo, hi):
            conts = []
            for c in range(int(self.pod_ncont[p])):
                cont: dict = {"name": f"c{c}"}
                if self.pod_has_req[p]:
                    cont["resources"] = {"requests": {"cpu": self._cpu_str(int(self.cont_cpu[p, c]), p + c),
                                                      "memory": self._mem_str(int(self.cont_mem[p, c]), p + c)},
                                         "limits": {"cpu": "64", "memory": "1Ti"}}
                elif c == 1:
                    cont["resources"] = {}  # resources present, requests absent
                conts.append(cont)
            spec: dict = {"containers": conts,
                          # ignored by the reference (src/util.rs:58): must not change any result
                          "initContainers": [{"name": "init", "resources": {"requests": {"cpu": "64", "memory": "1Ti"}}}]}
            sel = {}
            for k in range(self.n_keys):
                v = int(self.pod_sel[k, p])
                if v == SEL_NEVER:
                    sel[self.label_key(k)] = "outside-the-domain"
                elif v != 0:
                    sel[self.label_key(k)] = self.label_val(v)
            if sel or p % 3 == 0:
                spec["nodeSelector"] = sel  # includes explicit empty maps
            tol = [{"key": self.taint_key(t), "operator": "Equal", "value": "true", "effect": "NoSchedule"}
                   for t in range(self.n_taints) if (int(self.pod_tol[p]) >> t) & 1]
            if tol:
                spec["tolerations"] = tol
            out.append({"metadata": {"name": f"pod-{p:07d}", "namespace": "ns"}, "spec": spec,
                        "status": {"phase": "Pending"}})
        return out


def make_cluster(P: int, N: int, n_keys: int = 8, n_taints: int = 0, seed: int = 0x5EED0000, attempts: int = 5,
                 binary_suffixes: bool = False, hostname_key: Optional[int] = None, pod_offset: int = 0) -> Cluster:
    """Build the cluster of SURVEY.md section 8d for P pending pods and N nodes.
    hostname_key = k: label key k is kubernetes.io/hostname-like -- every node carries its own value (cardinality N), and the pods
    that constrain it (same 15 %) each name one node.
    pod_offset = o: the pods are pods [o, o + P) of the (unbounded) pod sequence of this seed -- the same pods a cluster of o + P pods holds in
    rows [o, o + P) -- so a rank can make ITS rows of the b-th batch without generating everybody's (bench.py: input batches, pod-row shards)."""
    if not (0 <= n_keys <= len(KEY_CARDINALITY)):
        raise ValueError("n_keys must be 0..16")
    if not (0 <= n_taints <= 64):
        raise ValueError("n_taints must be 0..64")
    c = Cluster(seed=seed, P=P, N=N, n_keys=n_keys, n_taints=n_taints, attempts=attempts, binary_suffixes=binary_suffixes)
    S = lambda sid, n: stream(seed, sid, n)  # noqa: E731

    # ---- nodes ---------------------------------------------------------------------------------
    cores = np.array(NODE_CORES, dtype=np.int64)[_below(S(1, N), len(NODE_CORES))]
    gpc = np.array(GIB_PER_CORE, dtype=np.int64)[_below(S(2, N), len(GIB_PER_CORE))]
    c.alloc_cpu_milli = cores * 1000
    c.alloc_mem_bytes = cores * gpc * GIB
    # already-bound load: uniform 0..90 % of allocatable; ~1 % of nodes over-committed (100..120 %)
    over = _bern(S(3, N), 0.01)
    permille_cpu = np.where(over, 1000 + _below(S(4, N), 200), _below(S(4, N), 901))
    permille_mem = np.where(over, 1000 + _below(S(5, N), 200), _below(S(5, N), 901))
    load_cpu = c.alloc_cpu_milli * permille_cpu // 1000
    load_mem = c.alloc_mem_bytes // 1000 * permille_mem + _below(S(6, N), 4096)  # odd byte counts on purpose
    c.bound_cnt = 1 + _below(S(7, N), 3)
    c.bound_cpu = np.zeros((N, 3), dtype=np.int64)
    c.bound_mem = np.zeros((N, 3), dtype=np.int64)
    f1, f2 = _below(S(8, N), 1001), _below(S(9, N), 1001)
    for b in range(3):
        live = c.bound_cnt > b
        last = c.bound_cnt == b + 1
        rem_cpu = load_cpu - c.bound_cpu.sum(axis=1)
        rem_mem = load_mem - c.bound_mem.sum(axis=1)
        frac = (f1 if b == 0 else f2)
        c.bound_cpu[:, b] = np.where(live, np.where(last, rem_cpu, rem_cpu * frac // 1000), 0)
        c.bound_mem[:, b] = np.where(live, np.where(last, rem_mem, rem_mem // 1000 * frac), 0)
    c.avail_cpu = c.alloc_cpu_milli - c.bound_cpu.sum(axis=1)
    c.avail_mem = c.alloc_mem_bytes - c.bound_mem.sum(axis=1)

    c.node_labels = np.zeros((n_keys, N), dtype=np.uint32)
    for k in range(n_keys):
        if hostname_key is not None and k == hostname_key:
            c.node_labels[k] = (1 + np.argsort(S(40 + k, N), kind="stable")).astype(np.uint32)  # a permutation: one value per node
            continue
        has = _bern(S(20 + k, N), 0.9)
        val = 1 + _below(S(40 + k, N), KEY_CARDINALITY[k])
        c.node_labels[k] = np.where(has, val, 0).astype(np.uint32)
    c.node_taints = np.zeros(N, dtype=np.uint64)
    for t in range(n_taints):
        c.node_taints |= (_bern(S(100 + t, N), 0.05).astype(np.uint64) << np.uint64(t))

    # ---- pods ----------------------------------------------------------------------------------
    SP = lambda sid, n: stream(seed, sid, n, start=pod_offset)  # noqa: E731
    c.pod_ncont = 1 + _below(SP(200, P), 3)
    c.pod_has_req = ~_bern(SP(201, P), 0.05)
    c.cont_cpu = np.zeros((P, 3), dtype=np.int64)
    c.cont_mem = np.zeros((P, 3), dtype=np.int64)
    for j in range(3):
        live = (c.pod_ncont > j) & c.pod_has_req
        # log-uniform by octave: cpu 50m..4000m per pod, memory 64 MiB..16 GiB per pod, split over containers
        oc = _below(SP(210 + j, P), 7)
        cpu = np.minimum(50 * (1 << oc) + _below(SP(220 + j, P), 1 << 30) % (50 * (1 << oc)), 4000)
        om = _below(SP(230 + j, P), 8)
        mem = (64 * MIB) * (1 << om) + _below(SP(240 + j, P), 1 << 62) % ((64 * MIB) * (1 << om))
        c.cont_cpu[:, j] = np.where(live, np.maximum(cpu // c.pod_ncont, 1), 0)
        c.cont_mem[:, j] = np.where(live, np.maximum(mem // c.pod_ncont, 1), 0)
    c.req_cpu = c.cont_cpu.sum(axis=1)
    c.req_mem = c.cont_mem.sum(axis=1)

    c.pod_sel = np.zeros((n_keys, P), dtype=np.uint32)
    for k in range(n_keys):
        con = _bern(SP(300 + k, P), 0.15)
        val = 1 + _below(SP(320 + k, P), N if (hostname_key is not None and k == hostname_key and N > 0) else KEY_CARDINALITY[k])
        never = _bern(SP(340 + k, P), 0.01)
        c.pod_sel[k] = np.where(con, np.where(never, SEL_NEVER, val), 0).astype(np.uint32)
    c.pod_tol = np.zeros(P, dtype=np.uint64)
    for t in range(n_taints):
        c.pod_tol |= (_bern(SP(400 + t, P), 0.3).astype(np.uint64) << np.uint64(t))
    if N > 0:
        c.samples = _below(stream(seed, 500, P * attempts, start=pod_offset * attempts), N).astype(np.uint32).reshape(P, attempts)
    else:
        c.samples = np.zeros((P, attempts), dtype=np.uint32)
    return c


# BASELINE.json configs
CONFIGS: Dict[str, dict] = {
    "C1": dict(P=100, N=20, n_keys=8, n_taints=0, flags=("FIT", "SEL")),
    "C2": dict(P=10_000, N=1_000, n_keys=0, n_taints=0, flags=("FIT",)),
    "C3": dict(P=100_000, N=5_000, n_keys=8, n_taints=0, flags=("FIT", "SEL")),
    "C4": dict(P=1_000_000, N=10_000, n_keys=8, n_taints=0, flags=("FIT", "SEL")),
    "C5": dict(P=1_000_000, N=50_000, n_keys=8, n_taints=16, flags=("FIT", "SEL", "TAINT")),
    # C3 with its eighth label key replaced by a hostname-like key (5 000 values): the high-cardinality case (not a BASELINE config)
    "C3h": dict(P=100_000, N=5_000, n_keys=8, n_taints=0, flags=("FIT", "SEL"), hostname_key=7),
    "C5h": dict(P=1_000_000, N=50_000, n_keys=8, n_taints=16, flags=("FIT", "SEL", "TAINT"), hostname_key=7),
}


def make_config(name: str, P: Optional[int] = None, N: Optional[int] = None, pod_offset: int = 0) -> Cluster:
    cfg = CONFIGS[name]
    idx = list(CONFIGS).index(name)
    return make_cluster(P if P is not None else cfg["P"], N if N is not None else cfg["N"], n_keys=cfg["n_keys"],
                        n_taints=cfg["n_taints"], seed=0x5EED0000 + idx, hostname_key=cfg.get("hostname_key"), pod_offset=pod_offset)
