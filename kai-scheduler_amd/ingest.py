"""Snapshot ingest: the reference's snapshot.json / snapshot.zip (plugins/snapshot/snapshot.go:33-66, cmd/snapshot-tool) → `abi.Snapshot`.

`ingest_json` / `ingest_file` call the native packer (csrc/kai_ingest.cpp, C ABI in include/kai_ingest.h) and copy its arrays into the
numpy-backed `abi.Snapshot` the rest of the host side uses.  `export_snapshot_json` is the inverse used by tests and benchmarks: it writes
a structure-of-arrays snapshot as reference-schema Kubernetes objects (what `cmd/snapshot-tool` would be fed), so that BASELINE config 1
("16-node / 64-pod snapshot via cmd/snapshot-tool") is an actual snapshot file.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass, field

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libkai_ingest.so")
EXPORTS = ["kai_ingest_parse", "kai_ingest_load", "kai_ingest_snapshot", "kai_ingest_config", "kai_ingest_actions", "kai_ingest_name",
           "kai_ingest_warnings", "kai_ingest_decisions_json", "kai_ingest_free", "kai_ingest_last_error", "kai_quantity_milli", "kai_quantity_value"]
ACTION_NAMES = ["allocate", "consolidation", "reclaim", "preempt"]


class KaiIngestOptions(C.Structure):
    _fields_ = [("scheduler_name", C.c_char_p), ("now_ns", C.c_int64), ("reserved", C.c_int32 * 4)]


_lib = None


def load_ingest_library(path: str = LIB_PATH):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing — build it with __graft_entry__.build()")
    lib = C.CDLL(path)
    lib.kai_ingest_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(KaiIngestOptions), C.POINTER(C.c_void_p)]
    lib.kai_ingest_load.argtypes = [C.c_char_p, C.POINTER(KaiIngestOptions), C.POINTER(C.c_void_p)]
    lib.kai_ingest_snapshot.argtypes = [C.c_void_p]; lib.kai_ingest_snapshot.restype = C.POINTER(abi.KaiSnapshotSoA)
    lib.kai_ingest_config.argtypes = [C.c_void_p]; lib.kai_ingest_config.restype = C.POINTER(abi.KaiConfig)
    lib.kai_ingest_actions.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    lib.kai_ingest_name.argtypes = [C.c_void_p, C.c_int, C.c_int]; lib.kai_ingest_name.restype = C.c_char_p
    lib.kai_ingest_warnings.argtypes = [C.c_void_p]; lib.kai_ingest_warnings.restype = C.c_char_p
    lib.kai_ingest_free.argtypes = [C.c_void_p]; lib.kai_ingest_free.restype = None
    lib.kai_ingest_decisions_json.argtypes = [C.c_void_p, C.POINTER(abi.KaiOp), C.c_int64, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.kai_ingest_last_error.restype = C.c_char_p
    lib.kai_quantity_milli.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    lib.kai_quantity_value.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    _lib = lib
    return lib


class IngestError(ValueError):
    pass


@dataclass
class Ingested:
    snapshot: abi.Snapshot
    config: abi.KaiConfig
    actions: list
    warnings: list
    resource_names: list = field(default_factory=list)
    _handle: object = None  # the native handle: keeps the name tables the decision writer needs

    def decisions_json(self, ops) -> str:
        """The committed operations (records of `Session.execute`, or (kind, pod, node, job) tuples) as the BindRequests / evictions the
        reference's cache would create (cache/cache.go:216-330) — one document for the whole batch."""
        lib = load_ingest_library()
        n = len(ops)
        arr = (abi.KaiOp * max(n, 1))()
        for i, o in enumerate(ops):
            kind, pod, node, job = (int(o["kind"]), int(o["pod"]), int(o["node"]), int(o["job"])) if not isinstance(o, tuple) else o
            arr[i].seq, arr[i].kind, arr[i].pod, arr[i].node, arr[i].job = i, kind, pod, node, job
        need = C.c_size_t(0)
        rc = lib.kai_ingest_decisions_json(self._handle, arr, n, None, 0, C.byref(need))
        if rc != -4:
            raise IngestError(lib.kai_ingest_last_error().decode())
        buf = C.create_string_buffer(need.value + 1)
        if lib.kai_ingest_decisions_json(self._handle, arr, n, buf, need.value + 1, C.byref(need)) != 0:
            raise IngestError(lib.kai_ingest_last_error().decode())
        return buf.value.decode()

    def close(self):
        if self._handle is not None:
            load_ingest_library().kai_ingest_free(self._handle); self._handle = None

    def __del__(self):
        try: self.close()
        except Exception: pass


def _copy(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def _from_handle(lib, h) -> Ingested:
    s = lib.kai_ingest_snapshot(h).contents
    R, N, P, S, J, Q = s.n_res, s.n_nodes, s.n_pods, s.n_podsets, s.n_jobs, s.n_queues
    dims = {"N": N, "P": P, "S": S, "J": J, "Q": Q}
    a = {}
    for name, dt, kind in abi._SPEC:
        ptr = getattr(s, name)
        if kind == "RN": a[name] = _copy(ptr, R * N, dt).reshape(R, N)
        elif kind == "RP": a[name] = _copy(ptr, R * P, dt).reshape(R, P)
        elif kind == "3Q": a[name] = _copy(ptr, 3 * Q, dt).reshape(3, Q) if ptr else np.zeros((3, Q), dt)
        elif kind == "CF": a[name] = _copy(ptr, s.n_pod_classes * s.n_node_classes, dt).reshape(s.n_pod_classes, s.n_node_classes)
        else: a[name] = _copy(ptr, dims[kind], dt)
    T, TL, D, G = s.n_topologies, s.n_topo_levels, s.n_domains, s.n_groups
    if T > 0:
        a["topo_level_off"] = _copy(s.topo_level_off, T + 1, np.int32)
        a["node_domain"] = _copy(s.node_domain, TL * N, np.int32).reshape(TL, N)
        a["domain_level"] = _copy(s.domain_level, D, np.int32); a["domain_parent"] = _copy(s.domain_parent, D, np.int32); a["domain_id_rank"] = _copy(s.domain_id_rank, D, np.uint32)
    if G > 0:
        for name, n in (("group_job", G), ("group_parent", G), ("group_name_rank", G), ("group_topology", G), ("group_required_level", G), ("group_preferred_level", G),
                        ("job_root_group", J), ("podset_group", S), ("podset_topology", S), ("podset_required_level", S), ("podset_preferred_level", S)):
            a[name] = _copy(getattr(s, name), n, dict(abi._SPEC_OPT)[name])
    a["job_signature"] = _copy(s.job_signature, J, np.int64); a["job_last_start_ns"] = _copy(s.job_last_start_ns, J, np.int64)
    a["queue_preempt_min_runtime_ns"] = _copy(s.queue_preempt_min_runtime_ns, Q, np.int64); a["queue_reclaim_min_runtime_ns"] = _copy(s.queue_reclaim_min_runtime_ns, Q, np.int64)
    if s.pod_gpu_portion:
        a["pod_gpu_portion"] = _copy(s.pod_gpu_portion, P, np.float64); a["pod_gpu_group"] = _copy(s.pod_gpu_group, P, np.int32)
    if s.pod_gpu_memory:
        a["pod_gpu_memory"] = _copy(s.pod_gpu_memory, P, np.int64)
    if s.res_mig_gpus:
        a["res_mig_gpus"] = _copy(s.res_mig_gpus, int(s.n_res), np.int32); a["res_mig_memory"] = _copy(s.res_mig_memory, int(s.n_res), np.int64)
    if s.node_gpu_memory and N:
        a["node_gpu_memory"] = _copy(s.node_gpu_memory, N, np.int64)
    snap = abi.Snapshot(n_res=R, arrays=a)
    names = lambda kind, n: [lib.kai_ingest_name(h, kind, i).decode() for i in range(n)]
    snap.node_names, snap.pod_names, snap.job_names, snap.queue_names, snap.podset_names = names(0, N), names(1, P), names(2, J), names(3, Q), names(4, S)
    snap.finalize()
    cfg = abi.KaiConfig(); C.memmove(C.byref(cfg), lib.kai_ingest_config(h), C.sizeof(cfg))
    buf = (C.c_int32 * 16)(); n = lib.kai_ingest_actions(h, buf, 16)
    if n < 0:
        raise IngestError(lib.kai_ingest_last_error().decode())
    warnings = [w for w in lib.kai_ingest_warnings(h).decode().split("\n") if w]
    return Ingested(snap, cfg, [ACTION_NAMES[buf[i]] for i in range(min(n, 16))], warnings, names(5, R), h)


def _options(scheduler_name, now_ns):
    o = KaiIngestOptions(); o.scheduler_name = scheduler_name.encode() if scheduler_name else None; o.now_ns = int(now_ns)
    return o


def ingest_json(text, scheduler_name: str | None = None, now_ns: int = 0) -> Ingested:
    lib = load_ingest_library()
    data = text.encode() if isinstance(text, str) else bytes(text)
    h = C.c_void_p(); o = _options(scheduler_name, now_ns)
    if lib.kai_ingest_parse(data, len(data), C.byref(o), C.byref(h)) != 0:
        raise IngestError(lib.kai_ingest_last_error().decode())
    try:
        return _from_handle(lib, h)
    except Exception:
        lib.kai_ingest_free(h)
        raise


def ingest_file(path: str, scheduler_name: str | None = None, now_ns: int = 0) -> Ingested:
    lib = load_ingest_library()
    h = C.c_void_p(); o = _options(scheduler_name, now_ns)
    if lib.kai_ingest_load(os.fsencode(path), C.byref(o), C.byref(h)) != 0:
        raise IngestError(lib.kai_ingest_last_error().decode())
    try:
        return _from_handle(lib, h)
    except Exception:
        lib.kai_ingest_free(h)
        raise


# ------------------------------------------------------------------------------------------------------------------------------
# export: structure-of-arrays snapshot → reference-schema objects
def _ts(ns: int) -> str:
    """RFC 3339 with nanoseconds (metav1.Time itself only carries seconds; the ingest reads fractions, which the tests use to keep
    the synthetic creation order exact)."""
    import datetime
    sec, frac = divmod(int(ns), 1_000_000_000)
    base = datetime.datetime(1970, 1, 1, tzinfo=datetime.timezone.utc) + datetime.timedelta(seconds=sec)
    return base.strftime("%Y-%m-%dT%H:%M:%S") + (f".{frac:09d}" if frac else "") + "Z"


def _dur(ns: int) -> str:
    return f"{int(ns)}ns"


_STATUS = abi.POD_STATUS
_EPOCH_NS = 1_700_000_000 * 1_000_000_000  # synthetic snapshots count time from 0: shifted so every timestamp is a valid date


def export_snapshot_json(snap: abi.Snapshot, cfg: abi.KaiConfig | None = None, actions=("allocate",), scheduler_name="kai-scheduler") -> dict:
    """Write `snap` as the objects `plugins/snapshot` serialises.  Objects are named by zero-padded rank so the byte-wise name order is
    the rank order of the arrays; predicate classes become a node label + a required node-affinity term; topology domains become level
    labels.  Raises on what the schema cannot express (fractional quantities, usage history)."""
    cfg = cfg or abi.default_config()
    a = snap.arrays
    R, N, P, S, J, Q = snap.n_res, snap.n_nodes, snap.n_pods, snap.n_podsets, snap.n_jobs, snap.n_queues
    if R != 4:
        raise ValueError("export: only the four core resources")
    w = lambda n: max(1, len(str(max(n - 1, 1))))
    node_name = [f"node-{int(r):0{w(N)}d}" for r in a["node_name_rank"]]
    queue_name = [f"queue-{int(r):0{w(Q)}d}" for r in a["queue_uid_rank"]]
    job_name = [f"job-{int(r):0{w(J)}d}" for r in a["job_uid_rank"]]
    pod_uid = [f"uid-{int(r):0{w(P)}d}" for r in a["pod_uid_rank"]]
    T = int(a["topo_level_off"].shape[0]) - 1 if "topo_level_off" in a else 0
    topo_levels = [[f"kai.test/t{t}-l{l}" for l in range(int(a["topo_level_off"][t + 1] - a["topo_level_off"][t]))] for t in range(T)]
    PC, NC = a["class_fit"].shape

    def qty(v):
        if float(v) != int(v):
            raise ValueError("export: non-integer quantity")
        return str(int(v))

    nodes = []
    dom_label = {}
    if T:
        D = a["domain_level"].shape[0]
        width = w(D)
        for d in range(D):
            dom_label[d] = f"d{int(a['domain_id_rank'][d]):0{width}d}"
    for i in range(N):
        labels = {"kai.test/nclass": f"c{int(a['node_class'][i])}"}
        f = int(a["node_flags"][i])
        if int(a["node_gpu_count"][i]) >= 0: labels["nvidia.com/gpu.count"] = str(int(a["node_gpu_count"][i]))
        if f & abi.NODE_GPU_WORKER: labels["node-role.kubernetes.io/gpu-worker"] = ""
        if f & abi.NODE_CPU_WORKER: labels["node-role.kubernetes.io/cpu-worker"] = ""
        if f & abi.NODE_MIG_ENABLED: labels["node-role.kubernetes.io/mig-enabled"] = "true"
        if f & abi.NODE_MIG_MIXED: labels["nvidia.com/mig.strategy"] = "mixed"
        for t in range(T):
            for l, key in enumerate(topo_levels[t]):
                d = int(a["node_domain"][int(a["topo_level_off"][t]) + l, i])
                if d >= 0: labels[key] = dom_label[d]
        alloc = {"cpu": f"{qty(a['node_allocatable'][0, i])}m", "memory": qty(a["node_allocatable"][1, i]), "nvidia.com/gpu": qty(a["node_allocatable"][2, i]),
                 "pods": qty(a["node_allocatable"][3, i])}
        nodes.append({"metadata": {"name": node_name[i], "labels": labels}, "spec": {"unschedulable": True} if f & abi.NODE_NOT_READY else {},
                      "status": {"allocatable": alloc, "conditions": [{"type": "Ready", "status": "True"}]}})

    prio_classes = sorted({int(p) for p in a["job_priority"]})
    pods, bind_requests = [], []
    podset_label = [f"ps-{int(a['podset_name_rank'][s]):03d}" for s in range(S)]
    has_groups = "group_job" in a
    job_simple = []  # one pod-set directly under an unconstrained position: expressed as spec.minMember, pods carry no sub-group label
    for j in range(J):
        s0, ns = int(a["job_first_podset"][j]), int(a["job_n_podsets"][j])
        job_simple.append(ns == 1 and (not has_groups or (int(a["podset_group"][s0]) == int(a["job_root_group"][j]) and int(a["podset_topology"][s0]) == -1)))
    for p in range(P):
        j, s, st = int(a["pod_job"][p]), int(a["pod_podset"][p]), int(a["pod_status"][p])
        md = {"name": f"pod-{p}", "namespace": "ns", "uid": pod_uid[p], "creationTimestamp": _ts(_EPOCH_NS + int(a["pod_created_ns"][p])), "labels": {}, "annotations": {}}
        spec = {"schedulerName": "default-scheduler" if int(a["pod_flags"][p]) & abi.POD_FOREIGN_SCHEDULER else scheduler_name,
                "containers": [{"name": "c", "resources": {"requests": {"cpu": f"{qty(a['pod_req'][0, p])}m", "memory": qty(a["pod_req"][1, p]), "nvidia.com/gpu": qty(a["pod_req"][2, p])}}}]}
        status = {"phase": "Pending"}
        if j >= 0:
            md["annotations"]["pod-group-name"] = job_name[j]
            if not job_simple[j]:
                md["labels"]["kai.scheduler/subgroup-name"] = podset_label[s]
        if int(a["pod_flags"][p]) & abi.POD_HAS_TASK_PRIORITY: md["labels"]["kai.scheduler/task-priority"] = str(int(a["pod_task_priority"][p]))
        if int(a["pod_flags"][p]) & abi.POD_CPU_FALLBACK: md["annotations"]["gpu-fraction"] = "0.5"
        nd = int(a["pod_node"][p])
        if st == _STATUS["Running"]: status["phase"] = "Running"; spec["nodeName"] = node_name[nd]
        elif st == _STATUS["Releasing"]: status["phase"] = "Running"; spec["nodeName"] = node_name[nd]; md["deletionTimestamp"] = _ts(_EPOCH_NS)
        elif st == _STATUS["Bound"]: spec["nodeName"] = node_name[nd]
        elif st == _STATUS["Binding"]: bind_requests.append({"metadata": {"name": f"br-{p}", "namespace": "ns"}, "spec": {"podName": f"pod-{p}", "selectedNode": node_name[nd]}})
        elif st == _STATUS["Gated"]: spec["schedulingGates"] = [{"name": "gate"}]
        elif st == _STATUS["Succeeded"]: status["phase"] = "Succeeded"
        elif st == _STATUS["Failed"]: status["phase"] = "Failed"
        elif st == _STATUS["Unknown"]: status["phase"] = "Unknown"
        elif st != _STATUS["Pending"]: raise ValueError(f"export: pod status {st} is a session status, not a cluster state")
        if st in (_STATUS["Succeeded"], _STATUS["Failed"], _STATUS["Unknown"]) and nd >= 0: spec["nodeName"] = node_name[nd]
        if int(a["pod_nominated_node"][p]) >= 0: status["nominatedNodeName"] = node_name[int(a["pod_nominated_node"][p])]
        fits = [c for c in range(NC) if a["class_fit"][int(a["pod_class"][p]), c]]
        if len(fits) < NC:
            spec["affinity"] = {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": [{"matchExpressions": [
                {"key": "kai.test/nclass", "operator": "In", "values": [f"c{c}" for c in fits]} if fits else {"key": "kai.test/nclass", "operator": "DoesNotExist"}]}]}}}
        pods.append({"metadata": md, "spec": spec, "status": status})

    def tc(topo, req, pref):
        if topo < 0: return None
        out = {"topology": f"topology-{topo}"}
        if req >= 0: out["requiredTopologyLevel"] = topo_levels[topo][req]
        if pref >= 0: out["preferredTopologyLevel"] = topo_levels[topo][pref]
        return out

    pod_groups = []
    for j in range(J):
        q = int(a["job_queue"][j])
        spec = {"queue": queue_name[q] if q >= 0 else "no-such-queue", "priorityClassName": f"prio-{int(a['job_priority'][j])}",
                "preemptibility": "preemptible" if int(a["job_preemptible"][j]) else "non-preemptible"}
        md = {"name": job_name[j], "namespace": "ns", "uid": f"pg-{j}", "creationTimestamp": _ts(_EPOCH_NS + int(a["job_created_ns"][j])), "annotations": {}}
        if "job_last_start_ns" in a and int(a["job_last_start_ns"][j]) != 0: md["annotations"]["kai.scheduler/last-start-timestamp"] = _ts(_EPOCH_NS + int(a["job_last_start_ns"][j]))
        s0, ns = int(a["job_first_podset"][j]), int(a["job_n_podsets"][j])
        simple = job_simple[j]
        if has_groups:
            rg = int(a["job_root_group"][j]); c = tc(int(a["group_topology"][rg]), int(a["group_required_level"][rg]), int(a["group_preferred_level"][rg]))
            if c: spec["topologyConstraint"] = c
        if simple:
            spec["minMember"] = int(a["podset_min_available"][s0])
        else:
            subs = []
            gname = {}
            if has_groups:
                rg = int(a["job_root_group"][j]); gname = {rg: None}
                for g in range(a["group_job"].shape[0]):
                    if int(a["group_job"][g]) == j and g != rg: gname[g] = f"grp-{int(a['group_name_rank'][g]):03d}"
                for g, nm in gname.items():
                    if nm is None: continue
                    e = {"name": nm}; par = gname[int(a["group_parent"][g])]
                    if par: e["parent"] = par
                    c = tc(int(a["group_topology"][g]), int(a["group_required_level"][g]), int(a["group_preferred_level"][g]))
                    if c: e["topologyConstraint"] = c
                    subs.append(e)
            for s in range(s0, s0 + ns):
                e = {"name": podset_label[s], "minMember": int(a["podset_min_available"][s])}
                if has_groups:
                    par = gname[int(a["podset_group"][s])]
                    if par: e["parent"] = par
                    c = tc(int(a["podset_topology"][s]), int(a["podset_required_level"][s]), int(a["podset_preferred_level"][s]))
                    if c: e["topologyConstraint"] = c
                subs.append(e)
            spec["subGroups"] = subs
        pod_groups.append({"metadata": md, "spec": spec})

    queues = []
    for q in range(Q):
        res = {k: {"quota": float(a["queue_deserved"][i, q]), "limit": float(a["queue_limit"][i, q]), "overQuotaWeight": float(a["queue_oqw"][i, q])} for i, k in enumerate(("cpu", "memory", "gpu"))}
        spec = {"resources": res, "priority": int(a["queue_priority"][q])}
        if int(a["queue_parent"][q]) >= 0: spec["parentQueue"] = queue_name[int(a["queue_parent"][q])]
        if "queue_preempt_min_runtime_ns" in a and int(a["queue_preempt_min_runtime_ns"][q]) >= 0: spec["preemptMinRuntime"] = _dur(a["queue_preempt_min_runtime_ns"][q])
        if "queue_reclaim_min_runtime_ns" in a and int(a["queue_reclaim_min_runtime_ns"][q]) >= 0: spec["reclaimMinRuntime"] = _dur(a["queue_reclaim_min_runtime_ns"][q])
        queues.append({"metadata": {"name": queue_name[q], "creationTimestamp": _ts(_EPOCH_NS + int(a["queue_created_ns"][q]))}, "spec": spec})
    if "queue_usage" in a and np.any(a["queue_usage"] != 0):
        raise ValueError("export: queue usage history is not part of the snapshot schema")

    plugin_names = [n for n, bit in abi.PLUGINS.items() if int(cfg.plugins) & bit]
    plugins = []
    for n in plugin_names:
        e = {"name": n}
        if n == "nodeplacement": e["arguments"] = {"gpu": "spread" if cfg.gpu_strategy == abi.SPREAD else "binpack", "cpu": "spread" if cfg.cpu_strategy == abi.SPREAD else "binpack"}
        if n == "proportion": e["arguments"] = {"kValue": repr(float(cfg.k_value)), "relcaimerSaturationMultiplier": repr(float(cfg.reclaimer_saturation_multiplier))}
        if n == "minruntime": e["arguments"] = {"defaultPreemptMinRuntime": _dur(cfg.default_preempt_min_runtime_ns), "defaultReclaimMinRuntime": _dur(cfg.default_reclaim_min_runtime_ns),
                                               "reclaimResolveMethod": "queue" if cfg.reclaim_resolve_method == 1 else "lca"}
        plugins.append(e)
    depth = {ACTION_NAMES[i]: int(cfg.queue_depth[i]) for i in range(4) if int(cfg.queue_depth[i]) >= 0}
    config = {"actions": ", ".join(list(actions) + ["stalegangeviction"]), "tiers": [{"plugins": plugins}]}
    if depth: config["queueDepthPerAction"] = depth
    params = {"schedulerName": scheduler_name, "restrictSchedulingNodes": bool(cfg.restrict_node_scheduling), "maxNumberConsolidationPreemptees": int(cfg.max_consolidation_preemptees),
              "useSchedulingSignatures": bool(cfg.use_scheduling_signatures), "fullHierarchyFairness": True, "allowConsolidatingReclaim": bool(cfg.allow_consolidating_reclaim)}
    return {"config": config, "schedulerParams": params,
            "rawObjects": {"pods": pods, "nodes": nodes, "queues": queues, "podGroups": pod_groups, "bindRequests": bind_requests,
                           "priorityClasses": [{"metadata": {"name": f"prio-{v}"}, "value": v} for v in prio_classes],
                           "configMaps": [], "persistentVolumeClaims": [], "csiStorageCapacities": [], "storageClasses": [], "csiDrivers": [], "resourceClaims": [],
                           "resourceSlices": [], "deviceClasses": [],
                           "topologies": [{"metadata": {"name": f"topology-{t}"}, "spec": {"levels": [{"nodeLabel": k} for k in topo_levels[t]]}} for t in range(T)]}}


def write_snapshot_zip(path: str, doc: dict) -> None:
    """snapshot.zip as the snapshot plugin serves it (one deflated member, snapshot.json)."""
    import zipfile
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("snapshot.json", json.dumps(doc))
