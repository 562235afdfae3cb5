"""ctypes mirror of include/kai_core.h and the structure-of-arrays session snapshot.

`Snapshot` is the host-side twin of the reference's `api.ClusterInfo` (pkg/scheduler/api/cluster_info.go:43-64)
already flattened to the arrays `kai_snapshot_soa` carries.  Strings never cross the ABI: names are ranked
once here (byte-wise order, like Go's string compare) because the reference's tie-breaks are string compares
(framework/session.go:480-485, session_plugins.go:227-260).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

KAI_ABI_VERSION = 5
RES_CPU, RES_MEM, RES_GPU, RES_PODS = 0, 1, 2, 3
MAX_RES = 8
Q_CPU, Q_MEM, Q_GPU = 0, 1, 2
UNLIMITED = -1.0

# pod status bit-set (api/pod_status/pod_status.go:25-71)
POD_STATUS = {
    "Pending": 1 << 0, "Gated": 1 << 1, "Allocated": 1 << 2, "Pipelined": 1 << 3, "Binding": 1 << 4, "Bound": 1 << 5,
    "Running": 1 << 6, "Releasing": 1 << 7, "Succeeded": 1 << 8, "Failed": 1 << 9, "Unknown": 1 << 10, "Deleted": 1 << 11,
}
POD_STATUS_NAME = {v: k for k, v in POD_STATUS.items()}
ACTIVE_USED = sum(POD_STATUS[s] for s in ("Allocated", "Pipelined", "Binding", "Bound", "Running", "Releasing"))

NODE_NOT_READY, NODE_MIG_ENABLED, NODE_MIG_MIXED, NODE_HAS_DRA_GPUS, NODE_GPU_WORKER, NODE_CPU_WORKER, NODE_MIG_SINGLE = 1, 2, 4, 8, 16, 32, 64
POD_FOREIGN_SCHEDULER, POD_HAS_TASK_PRIORITY, POD_CPU_FALLBACK, POD_GPU_UNMODELLED, POD_LEGACY_MIG = 1, 2, 4, 8, 16

ACTIONS = {"allocate": 0, "consolidation": 1, "reclaim": 2, "preempt": 3}
OP_KIND = {0: "allocate", 1: "pipeline", 2: "evict"}
BINPACK, SPREAD = 0, 1
PLUGINS = {"predicates": 0x001, "proportion": 0x002, "priority": 0x004, "elastic": 0x008, "nodeavailability": 0x010,
           "resourcetype": 0x020, "subgrouporder": 0x040, "taskorder": 0x080, "nominatednode": 0x100, "nodeplacement": 0x200,
           "minruntime": 0x400, "topology": 0x800, "gpusharingorder": 0x1000, "gpupack": 0x2000, "gpuspread": 0x4000}
PLUGIN_PREDICATES, PLUGIN_PROPORTION, PLUGIN_PRIORITY, PLUGIN_ELASTIC, PLUGIN_NODEAVAILABILITY, PLUGIN_RESOURCETYPE = 0x001, 0x002, 0x004, 0x008, 0x010, 0x020
PLUGIN_SUBGROUPORDER, PLUGIN_TASKORDER, PLUGIN_NOMINATEDNODE, PLUGIN_NODEPLACEMENT, PLUGIN_MINRUNTIME, PLUGIN_TOPOLOGY = 0x040, 0x080, 0x100, 0x200, 0x400, 0x800
PLUGIN_ALL = 0x3FFF  # the default tier list (conf_util/scheduler_conf_util.go:36-61): everything except gpuspread

STATUS_TEXT = {0: "ok", -1: "invalid argument", -2: "no HIP device", -3: "HIP runtime error", -4: "output capacity",
               -5: "unsupported snapshot feature", -6: "call order", -7: "device engine fault", -8: "multi-GPU exchange"}


class KaiConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("gpu_strategy", C.c_int32), ("cpu_strategy", C.c_int32), ("k_value", C.c_double),
        ("reclaimer_saturation_multiplier", C.c_double), ("plugins", C.c_uint32), ("restrict_node_scheduling", C.c_int32),
        ("max_consolidation_preemptees", C.c_int32), ("use_scheduling_signatures", C.c_int32), ("allow_consolidating_reclaim", C.c_int32),
        ("full_hierarchy_fairness", C.c_int32), ("min_node_gpu_memory", C.c_int64), ("queue_depth", C.c_int32 * 4),
        ("engine_mode", C.c_int32), ("reserved", C.c_int32 * 7),
        ("now_ns", C.c_int64), ("default_preempt_min_runtime_ns", C.c_int64), ("default_reclaim_min_runtime_ns", C.c_int64),
        ("reclaim_resolve_method", C.c_int32), ("pad0", C.c_int32),
    ]


def default_config(**kw) -> KaiConfig:
    """Defaults of conf_util/scheduler_conf_util.go:36-61 + conf/scheduler_conf.go:31-46."""
    c = KaiConfig()
    c.abi_version = KAI_ABI_VERSION
    c.gpu_strategy = BINPACK
    c.cpu_strategy = BINPACK
    c.k_value = 1.0
    c.reclaimer_saturation_multiplier = 1.0
    c.plugins = PLUGIN_ALL
    c.restrict_node_scheduling = 0
    c.max_consolidation_preemptees = 16
    c.use_scheduling_signatures = 1
    c.allow_consolidating_reclaim = 1
    c.full_hierarchy_fairness = 1
    c.min_node_gpu_memory = 100
    for i in range(4):
        c.queue_depth[i] = -1
    for k, v in kw.items():
        setattr(c, k, v)
    return c


_P = C.POINTER


class KaiSnapshotSoA(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("n_res", C.c_int32),
        ("n_nodes", C.c_int32), ("node_allocatable", _P(C.c_double)), ("node_flags", _P(C.c_uint32)), ("node_gpu_count", _P(C.c_int32)),
        ("node_name_rank", _P(C.c_uint32)), ("node_class", _P(C.c_int32)),
        ("n_pods", C.c_int32), ("pod_req", _P(C.c_double)), ("pod_job", _P(C.c_int32)), ("pod_podset", _P(C.c_int32)),
        ("pod_status", _P(C.c_int32)), ("pod_node", _P(C.c_int32)), ("pod_flags", _P(C.c_uint32)), ("pod_task_priority", _P(C.c_int32)),
        ("pod_created_ns", _P(C.c_int64)), ("pod_uid_rank", _P(C.c_uint32)), ("pod_class", _P(C.c_int32)), ("pod_nominated_node", _P(C.c_int32)),
        ("n_podsets", C.c_int32), ("podset_job", _P(C.c_int32)), ("podset_min_available", _P(C.c_int32)), ("podset_name_rank", _P(C.c_uint32)),
        ("n_jobs", C.c_int32), ("job_queue", _P(C.c_int32)), ("job_priority", _P(C.c_int32)), ("job_preemptible", _P(C.c_int32)),
        ("job_created_ns", _P(C.c_int64)), ("job_uid_rank", _P(C.c_uint32)), ("job_first_pod", _P(C.c_int32)), ("job_n_pods", _P(C.c_int32)),
        ("job_first_podset", _P(C.c_int32)), ("job_n_podsets", _P(C.c_int32)),
        ("n_queues", C.c_int32), ("queue_parent", _P(C.c_int32)), ("queue_priority", _P(C.c_int32)), ("queue_created_ns", _P(C.c_int64)),
        ("queue_uid_rank", _P(C.c_uint32)), ("queue_deserved", _P(C.c_double)), ("queue_limit", _P(C.c_double)), ("queue_oqw", _P(C.c_double)),
        ("queue_usage", _P(C.c_double)),
        ("n_pod_classes", C.c_int32), ("n_node_classes", C.c_int32), ("class_fit", _P(C.c_uint8)),
        ("n_topologies", C.c_int32), ("topo_level_off", _P(C.c_int32)), ("n_topo_levels", C.c_int32), ("node_domain", _P(C.c_int32)),
        ("n_domains", C.c_int32), ("domain_level", _P(C.c_int32)), ("domain_parent", _P(C.c_int32)), ("domain_id_rank", _P(C.c_uint32)),
        ("n_groups", C.c_int32), ("group_job", _P(C.c_int32)), ("group_parent", _P(C.c_int32)), ("group_name_rank", _P(C.c_uint32)),
        ("group_topology", _P(C.c_int32)), ("group_required_level", _P(C.c_int32)), ("group_preferred_level", _P(C.c_int32)),
        ("job_root_group", _P(C.c_int32)), ("podset_group", _P(C.c_int32)), ("podset_topology", _P(C.c_int32)),
        ("podset_required_level", _P(C.c_int32)), ("podset_preferred_level", _P(C.c_int32)),
        ("job_signature", _P(C.c_int64)),
        ("job_last_start_ns", _P(C.c_int64)), ("queue_preempt_min_runtime_ns", _P(C.c_int64)), ("queue_reclaim_min_runtime_ns", _P(C.c_int64)),
        ("pod_gpu_portion", _P(C.c_double)), ("pod_gpu_group", _P(C.c_int32)), ("node_gpu_memory", _P(C.c_int64)), ("pod_gpu_memory", _P(C.c_int64)),
        ("res_mig_gpus", _P(C.c_int32)), ("res_mig_memory", _P(C.c_int64)),
    ]


class KaiOp(C.Structure):
    _fields_ = [("seq", C.c_int64), ("kind", C.c_int32), ("pod", C.c_int32), ("node", C.c_int32), ("job", C.c_int32), ("stmt", C.c_int32), ("pad", C.c_int32)]


class KaiQueueShare(C.Structure):
    _fields_ = [(n, C.c_double * 3) for n in ("fair_share", "allocated", "allocated_non_preemptible", "request", "deserved", "max_allowed")]


class KaiNodeState(C.Structure):
    _fields_ = [("idle", C.c_double * MAX_RES), ("releasing", C.c_double * MAX_RES), ("used", C.c_double * MAX_RES)]


class KaiActionStats(C.Structure):
    _fields_ = [("decisions", C.c_int64), ("node_scans", C.c_int64), ("nodes_scanned", C.c_int64), ("jobs_attempted", C.c_int64),
                ("jobs_committed", C.c_int64), ("rollbacks", C.c_int64), ("kernel_ms", C.c_double), ("upload_ms", C.c_double),
                ("reserved", C.c_int64 * 8)]


def rank_strings(names) -> np.ndarray:
    """Rank of each string in byte-wise ascending order (Go's `<` on strings); ties keep first-seen order."""
    order = sorted(range(len(names)), key=lambda i: (names[i].encode("utf-8"), i))
    rank = np.empty(len(names), dtype=np.uint32)
    for r, i in enumerate(order):
        rank[i] = r
    return rank


_SPEC = [  # (field, dtype, shape-kind)
    ("node_allocatable", np.float64, "RN"), ("node_flags", np.uint32, "N"), ("node_gpu_count", np.int32, "N"),
    ("node_name_rank", np.uint32, "N"), ("node_class", np.int32, "N"),
    ("pod_req", np.float64, "RP"), ("pod_job", np.int32, "P"), ("pod_podset", np.int32, "P"), ("pod_status", np.int32, "P"),
    ("pod_node", np.int32, "P"), ("pod_flags", np.uint32, "P"), ("pod_task_priority", np.int32, "P"), ("pod_created_ns", np.int64, "P"),
    ("pod_uid_rank", np.uint32, "P"), ("pod_class", np.int32, "P"), ("pod_nominated_node", np.int32, "P"),
    ("podset_job", np.int32, "S"), ("podset_min_available", np.int32, "S"), ("podset_name_rank", np.uint32, "S"),
    ("job_queue", np.int32, "J"), ("job_priority", np.int32, "J"), ("job_preemptible", np.int32, "J"), ("job_created_ns", np.int64, "J"),
    ("job_uid_rank", np.uint32, "J"), ("job_first_pod", np.int32, "J"), ("job_n_pods", np.int32, "J"), ("job_first_podset", np.int32, "J"),
    ("job_n_podsets", np.int32, "J"),
    ("queue_parent", np.int32, "Q"), ("queue_priority", np.int32, "Q"), ("queue_created_ns", np.int64, "Q"), ("queue_uid_rank", np.uint32, "Q"),
    ("queue_deserved", np.float64, "3Q"), ("queue_limit", np.float64, "3Q"), ("queue_oqw", np.float64, "3Q"), ("queue_usage", np.float64, "3Q"),
    ("class_fit", np.uint8, "CF"),
]
# optional topology + sub-group-tree arrays (kai_core.h): present only when the snapshot carries them
_SPEC_OPT = [
    ("topo_level_off", np.int32), ("node_domain", np.int32), ("domain_level", np.int32), ("domain_parent", np.int32), ("domain_id_rank", np.uint32),
    ("group_job", np.int32), ("group_parent", np.int32), ("group_name_rank", np.uint32), ("group_topology", np.int32),
    ("group_required_level", np.int32), ("group_preferred_level", np.int32), ("job_root_group", np.int32), ("podset_group", np.int32),
    ("podset_topology", np.int32), ("podset_required_level", np.int32), ("podset_preferred_level", np.int32),
    ("job_signature", np.int64), ("job_last_start_ns", np.int64), ("queue_preempt_min_runtime_ns", np.int64), ("queue_reclaim_min_runtime_ns", np.int64),
    ("pod_gpu_portion", np.float64), ("pod_gpu_group", np.int32), ("node_gpu_memory", np.int64), ("pod_gpu_memory", np.int64), ("res_mig_gpus", np.int32), ("res_mig_memory", np.int64),
]


@dataclass
class Snapshot:
    """numpy-backed kai_snapshot_soa.  2-D arrays are resource-major ([R][N])."""
    n_res: int = 4
    arrays: dict = field(default_factory=dict)
    # optional name tables (never cross the ABI; for tests / reporting)
    node_names: list = field(default_factory=list)
    pod_names: list = field(default_factory=list)
    job_names: list = field(default_factory=list)
    queue_names: list = field(default_factory=list)
    podset_names: list = field(default_factory=list)

    def __getattr__(self, k):
        arrays = self.__dict__.get("arrays", {})
        if k in arrays:
            return arrays[k]
        raise AttributeError(k)

    @property
    def n_nodes(self): return int(self.arrays["node_flags"].shape[0])
    @property
    def n_pods(self): return int(self.arrays["pod_job"].shape[0])
    @property
    def n_podsets(self): return int(self.arrays["podset_job"].shape[0])
    @property
    def n_jobs(self): return int(self.arrays["job_queue"].shape[0])
    @property
    def n_queues(self): return int(self.arrays["queue_parent"].shape[0])
    @property
    def n_pod_classes(self): return int(self.arrays["class_fit"].shape[0])
    @property
    def n_node_classes(self): return int(self.arrays["class_fit"].shape[1])

    def finalize(self):
        """Coerce dtypes / contiguity and fill optional arrays with their neutral defaults."""
        a = self.arrays
        N, P = self.n_nodes, self.n_pods
        a.setdefault("node_gpu_count", np.full(N, -1, np.int32))
        a.setdefault("node_class", np.zeros(N, np.int32))
        a.setdefault("pod_flags", np.zeros(P, np.uint32))
        a.setdefault("pod_task_priority", np.zeros(P, np.int32))
        a.setdefault("pod_created_ns", np.zeros(P, np.int64))
        a.setdefault("pod_class", np.zeros(P, np.int32))
        a.setdefault("pod_nominated_node", np.full(P, -1, np.int32))
        a.setdefault("class_fit", np.ones((1, 1), np.uint8))
        a.setdefault("queue_usage", np.zeros((3, self.n_queues), np.float64))
        for name, dt, _ in _SPEC:
            a[name] = np.ascontiguousarray(a[name], dtype=dt)
        for name, dt in _SPEC_OPT:
            if name in a:
                a[name] = np.ascontiguousarray(a[name], dtype=dt)
        assert a["node_allocatable"].shape == (self.n_res, N), a["node_allocatable"].shape
        assert a["pod_req"].shape == (self.n_res, P), a["pod_req"].shape
        return self

    def as_struct(self) -> KaiSnapshotSoA:
        a = self.arrays
        s = KaiSnapshotSoA()
        s.abi_version = KAI_ABI_VERSION
        s.n_res = self.n_res
        s.n_nodes, s.n_pods, s.n_podsets, s.n_jobs, s.n_queues = self.n_nodes, self.n_pods, self.n_podsets, self.n_jobs, self.n_queues
        s.n_pod_classes, s.n_node_classes = self.n_pod_classes, self.n_node_classes
        ctype = {np.dtype(np.float64): C.c_double, np.dtype(np.int32): C.c_int32, np.dtype(np.uint32): C.c_uint32,
                 np.dtype(np.int64): C.c_int64, np.dtype(np.uint8): C.c_uint8}
        for name, dt, _ in _SPEC:
            arr = a[name]
            setattr(s, name, arr.ctypes.data_as(_P(ctype[np.dtype(dt)])))
        for name, dt in _SPEC_OPT:
            if name in a:
                setattr(s, name, a[name].ctypes.data_as(_P(ctype[np.dtype(dt)])))
        s.n_topologies = int(a["topo_level_off"].shape[0]) - 1 if "topo_level_off" in a else 0
        s.n_topo_levels = int(a["topo_level_off"][-1]) if "topo_level_off" in a else 0
        s.n_domains = int(a["domain_level"].shape[0]) if "domain_level" in a else 0
        s.n_groups = int(a["group_job"].shape[0]) if "group_job" in a else 0
        s._keepalive = a  # the struct borrows the numpy buffers
        return s
