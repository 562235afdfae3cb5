"""Test infrastructure shared by the CPU and GPU suites.

 * `pkg`               — the product's Python host package (directory `kai-scheduler_amd/`, not importable by name)
 * `Oracle`            — ctypes wrapper of oracle/liboracle.so (the CPU restatement; checker only)
 * `case_to_snapshot`  — restates the reference's test fixture builders
                         (pkg/scheduler/test_utils/test_utils_builder.go:94-289, jobs_fake/jobs.go:51-330,
                          nodes_fake/nodes.go:52-287, resources_fake/resources.go:32-77) so a golden table
                         transcribed by tools/go_fixtures.py becomes the same session the Go test builds.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import json
import math
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _load_pkg():
    name = "kai_scheduler_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "kai-scheduler_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


pkg = _load_pkg()
abi = pkg.abi


# ------------------------------------------------------------------------------------------------ oracle
class Oracle:
    """CPU oracle (oracle/liboracle.so).  Builds it on first use; it is test infrastructure, never shipped."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            so = os.environ.get("KAI_ORACLE_SO") or os.path.join(ROOT, "oracle", "liboracle.so")  # override: a -DORC_TRACE build while debugging
            srcs = [os.path.join(ROOT, "oracle", f) for f in ("kai_oracle.cpp", "oracle_model.hpp", "oracle_session.hpp", "oracle_solver.hpp")]
            if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
                subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
            lib = C.CDLL(so)
            lib.kai_oracle_run.restype = C.c_int
            lib.kai_oracle_pack_score.restype = C.c_double
            lib.kai_oracle_pack_score.argtypes = [C.c_double] * 4
            lib.kai_oracle_set_resources_share.restype = C.c_int
            cls._lib = lib
        return cls._lib

    @classmethod
    def run(cls, snap, cfg=None, actions=("allocate",), threads=1):
        """threads > 1: the node-scoring fan-out of OrderedNodesByTask on that many threads (the reference scores nodes on goroutines); same results."""
        lib = cls.lib()
        if hasattr(lib, "kai_oracle_set_threads"): lib.kai_oracle_set_threads(int(threads))  # (the host twin shares this wrapper and has no such knob)
        cfg = cfg or abi.default_config()
        s = snap.as_struct()
        acts = (C.c_int * len(actions))(*[abi.ACTIONS[a] for a in actions])
        P, Q, N = snap.n_pods, snap.n_queues, snap.n_nodes
        cap = max(16, 4 * P)
        ops = (abi.KaiOp * cap)()
        n_ops = C.c_int64(0)
        status = np.zeros(P, np.int32); node = np.zeros(P, np.int32)
        sh_open = (abi.KaiQueueShare * max(Q, 1))(); sh_fin = (abi.KaiQueueShare * max(Q, 1))()
        nodes = (abi.KaiNodeState * max(N, 1))()
        stats = abi.KaiActionStats(); ms = C.c_double(0)
        rc = lib.kai_oracle_run(C.byref(cfg), C.byref(s), acts, len(actions), ops, C.c_int64(cap), C.byref(n_ops),
                                status.ctypes.data_as(C.POINTER(C.c_int32)), node.ctypes.data_as(C.POINTER(C.c_int32)),
                                sh_open, sh_fin, nodes, C.byref(stats), C.byref(ms))
        if rc != 0:
            raise RuntimeError(f"oracle rc={rc}")
        groups = np.full(P, -1, np.int32)
        if hasattr(lib, "kai_oracle_last_gpu_groups"):
            lib.kai_oracle_last_gpu_groups(groups.ctypes.data_as(C.POINTER(C.c_int32)), P)
        return Result(gpu_groups=groups, ops=[(o.kind, o.pod, o.node, o.job) for o in ops[: n_ops.value]], stmts=[o.stmt for o in ops[: n_ops.value]], pod_status=status, pod_node=node,
                      shares_open=shares_to_np(sh_open, Q), shares_final=shares_to_np(sh_fin, Q), nodes=nodes_to_np(nodes, N, snap.n_res),
                      stats=stats, elapsed_ms=ms.value)


class Result:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def ops_sha256(ops):
    """SHA-256 of a committed operation stream [(kind, pod, node, job), ...] as little-endian int32 quadruples in commit order — what profiles/full_size_pins.json
    holds for the oracle's full-size runs (tools/pin_full_sizes.py) and bench.py prints for the MI355X's."""
    import hashlib
    return hashlib.sha256(np.asarray([tuple(o)[:4] for o in ops], dtype="<i4").reshape(-1, 4).tobytes()).hexdigest()


def state_sha256(res):
    """SHA-256 of the final pod statuses and nodes plus the node accounting and the final queue shares of a result."""
    import hashlib
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(res.pod_status, dtype="<i4").tobytes()); h.update(np.ascontiguousarray(res.pod_node, dtype="<i4").tobytes())
    for k in ("idle", "releasing", "used"): h.update(np.ascontiguousarray(res.nodes[k], dtype="<f8").tobytes())
    for k in ("fair_share", "allocated", "allocated_non_preemptible", "request"): h.update(np.ascontiguousarray(res.shares_final[k], dtype="<f8").tobytes())
    return h.hexdigest()


def shares_to_np(sh, Q):
    out = {}
    for f in ("fair_share", "allocated", "allocated_non_preemptible", "request", "deserved", "max_allowed"):
        out[f] = np.array([[getattr(sh[q], f)[r] for r in range(3)] for q in range(Q)], dtype=np.float64).reshape(Q, 3)
    return out


def nodes_to_np(nodes, N, R):
    out = {}
    for f in ("idle", "releasing", "used"):
        out[f] = np.array([[getattr(nodes[n], f)[r] for r in range(R)] for n in range(N)], dtype=np.float64).reshape(N, R)
    return out


# ------------------------------------------------------------------------------------------------ fixtures
class Unsupported(Exception):
    """The golden case needs a feature outside the path built so far (documented in DESIGN.md)."""


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def _milli(v):  # resource.MustParse(FormatFloat(v)).MilliValue(): cores → milli-cores, rounded up
    return float(math.ceil(round(v * 1000.0, 6)))


def _value(v):  # Quantity.Value(): rounded up to an integer
    return float(math.ceil(v))


def _flatten_subgroups(root):
    """SubGroupSet.GetAllPodSets (subgroup_info/subgroupset.go:57-69) — returns [(name, minAvailable, constraint, parents)]."""
    out = []

    def walk(g, parents):
        for ps in g.get("PodSets", []):
            out.append((ps["Name"], int(ps["MinAvailable"]), ps.get("TopologyConstraint"), parents))
        for sg in g.get("SubGroups", []):
            walk(sg, parents + [sg.get("Name", "")])

    walk(root, [])
    return out


def _expand_nodes(nodes):
    """actions/allocate/allocateTopology_test.go:3162-3188 buildEvenlyDistributedTopologyNodes(zones, spines/zone, racks/spine, nodes/rack, gpus)"""
    if not (isinstance(nodes, dict) and nodes.get("_call") == "buildEvenlyDistributedTopologyNodes"):
        return nodes
    nz, ns, nr, nn, gpus = [int(x) for x in nodes["args"]]
    out, node_id = {}, 0
    for z in range(1, nz + 1):
        for sp in range(1, ns + 1):
            for r in range(1, nr + 1):
                for _ in range(nn):
                    out[f"node{node_id}"] = {"GPUs": gpus, "Labels": {"k8s.io/zone": f"zone{z}", "k8s.io/spine": f"spine{sp + (z - 1) * ns}",
                                                                      "k8s.io/rack": f"rack{r + (sp - 1) * nr + (z - 1) * ns * nr}"}}
                    node_id += 1
    return out


def case_to_snapshot(case, actions=("allocate",), fractions=False):
    """→ (Snapshot, KaiConfig, meta).  Raises Unsupported for features outside the built path.  `fractions`: accept tasks that ask for a fraction of
    one GPU (ABI v4 arrays) — restated by the oracle for the allocate action only, so only its golden test asks for them."""
    S = abi.POD_STATUS
    case = dict(case); case["Nodes"] = _expand_nodes(case.get("Nodes") or {})
    topologies = case.get("Topologies") or []
    topo_names = [t["ObjectMeta"]["Name"] for t in topologies]
    topo_levels = [[lv["NodeLabel"] for lv in t["Spec"]["Levels"]] for t in topologies]
    mocks = case.get("Mocks") or {}
    cfg = abi.default_config(max_consolidation_preemptees=-1)  # test_utils_builder.go:73-79: SchedulerParams carries only the queue label key
    cfg.use_scheduling_signatures = 0
    # addSessionPlugins (test_utils_builder.go:297-321): "predicates" is skipped unless a cache mock exists
    cache_mock = bool(mocks) and mocks.get("CacheRequirements") is not None and mocks.get("Cache") is None
    plugins = abi.PLUGIN_ALL
    if mocks.get("SchedulerConf"):
        conf = mocks["SchedulerConf"]
        plugins = 0
        for tier in conf.get("Tiers", []):
            for pl in tier.get("Plugins", []):
                nm = pl.get("Name")
                if nm in abi.PLUGINS:
                    plugins |= abi.PLUGINS[nm]
                args = pl.get("Arguments") or {}
                if nm == "nodeplacement":
                    strat = {"binpack": abi.BINPACK, "spread": abi.SPREAD}
                    for k, v in args.items():
                        key = {"constants.GPUResource": "gpu", "constants.CPUResource": "cpu"}.get(k, k)
                        val = {"constants.SpreadStrategy": "spread", "constants.BinpackStrategy": "binpack"}.get(v, v)
                        if key == "gpu": cfg.gpu_strategy = strat[val]
                        if key == "cpu": cfg.cpu_strategy = strat[val]
                if nm == "proportion" and "kValue" in args:
                    cfg.k_value = float(args["kValue"])
                if nm in ("gpupack", "gpuspread", "gpusharingorder", "kubeflow", "ray", "dynamicresources", "snapshot", "podaffinity"):
                    pass
    if not cache_mock:
        plugins &= ~abi.PLUGINS["predicates"]
    cfg.plugins = plugins

    # ---- queues + departments (test_utils_builder.go:94-233)
    queues = [dict(q) for q in case.get("Queues", [])]
    departments = [dict(d) for d in case.get("Departments", [])]
    if not departments and not case.get("DisableDefaultDepartment"):
        for q in queues:
            q["ParentQueue"] = "default"
        departments = [{"Name": "default", "DeservedGPUs": -1.0, "MaxAllowedGPUs": -1.0}]
    qnames, qrec = [], []
    for i, q in enumerate(queues):
        if q.get("V1"):
            raise Unsupported("v1 queue")
        rec = dict(name=q["Name"], parent=q.get("ParentQueue", ""), priority=q.get("Priority"), created=i * 60_000_000_000 + i,
                   deserved=[-1.0, -1.0, float(q.get("DeservedGPUs", 0))],
                   limit=[-1.0, -1.0, float(q["MaxAllowedGPUs"]) if q.get("MaxAllowedGPUs", 0) != 0 else -1.0],
                   oqw=[1.0, 1.0, float(q.get("GPUOverQuotaWeight", 0))])
        if q.get("DeservedCPUs") is not None: rec["deserved"][0] = float(q["DeservedCPUs"])
        if q.get("DeservedMemory") is not None: rec["deserved"][1] = float(q["DeservedMemory"])
        if q.get("MaxAllowedCPUs") is not None: rec["limit"][0] = float(q["MaxAllowedCPUs"])
        if q.get("MaxAllowedMemory") is not None: rec["limit"][1] = float(q["MaxAllowedMemory"])
        qnames.append(q["Name"]); qrec.append(rec)
    for i, d in enumerate(departments):
        rec = dict(name=d["Name"], parent="", priority=None, created=i * 60_000_000_000 + 1000 + i,
                   deserved=[-1.0, -1.0, float(d.get("DeservedGPUs", 0))],
                   limit=[-1.0, -1.0, float(d["MaxAllowedGPUs"]) if d.get("MaxAllowedGPUs", 0) != 0 else -1.0],
                   oqw=[1.0, 1.0, float(d.get("DeservedGPUs", 0))])
        if d.get("MaxAllowedCPUs") is not None: rec["limit"][0] = float(d["MaxAllowedCPUs"])
        if d.get("MaxAllowedMemory") is not None: rec["limit"][1] = float(d["MaxAllowedMemory"])
        if d["Name"] in qnames:  # mergeQueues: departments override same-named queues
            qrec[qnames.index(d["Name"])] = rec
        else:
            qnames.append(d["Name"]); qrec.append(rec)
    # UpdateQueueHierarchy: orphans (missing parent) are dropped with their subtree (cache/cluster_info/queue.go:105-129)
    alive = {r["name"] for r in qrec}
    changed = True
    while changed:
        changed = False
        for r in qrec:
            if r["name"] in alive and r["parent"] and r["parent"] not in alive:
                alive.discard(r["name"]); changed = True
    qrec = [r for r in qrec if r["name"] in alive]
    qnames = [r["name"] for r in qrec]
    Q = len(qrec)
    qidx = {n: i for i, n in enumerate(qnames)}

    # ---- nodes (nodes_fake/nodes.go:52-287)
    node_names = sorted(case.get("Nodes", {}).keys())
    nidx = {n: i for i, n in enumerate(node_names)}
    N = len(node_names)
    # MIG instance types of the case (nodes' MigInstances, tasks' RequiredMigInstances): one resource row each, after the four base rows (ABI v5 res_mig_*)
    mig_names = sorted({k for nd in case.get("Nodes", {}).values() if isinstance(nd, dict) for k in (nd.get("MigInstances") or {})} |
                       {k for job in case.get("Jobs", []) if isinstance(job, dict) for t in (job.get("Tasks") or []) if isinstance(t, dict) for k in (t.get("RequiredMigInstances") or {})})
    if mig_names and not fractions:
        raise Unsupported("MIG instances (oracle only, allocate action)")
    if len(mig_names) > 4:
        raise Unsupported("more MIG profiles than resource rows")
    import re as _re
    mig_row = {k: 4 + i for i, k in enumerate(mig_names)}
    R = 4 + len(mig_names)
    alloc = np.zeros((R, N)); nflags = np.zeros(N, np.uint32); gpu_count = np.zeros(N, np.int32)
    for i, nm in enumerate(node_names):
        nd = case["Nodes"][nm]
        for k, v in (nd.get("MigInstances") or {}).items():
            alloc[mig_row[k], i] = int(v)  # BuildResourceList (resources_fake/resources.go:37-81): the instance count
        gpus = int(nd.get("GPUs", 0))
        mig = nd.get("MigStrategy", "")
        alloc[abi.RES_CPU, i] = _milli(nd["CPUMillis"]) if nd.get("CPUMillis", 0) > 0 else 20000.0 * 1000.0
        alloc[abi.RES_MEM, i] = _value(nd["CPUMemory"]) if nd.get("CPUMemory", 0) > 0 else 20e9
        alloc[abi.RES_GPU, i] = 0 if mig == "mixed" else gpus
        alloc[abi.RES_PODS, i] = nd["MaxTaskNum"] if nd.get("MaxTaskNum") is not None else 110
        gpu_count[i] = gpus
        if mig not in ("", None):  # migEnabledLabel "true" (nodes.go:203-206)
            nflags[i] |= abi.NODE_MIG_ENABLED
            if mig == "mixed":
                nflags[i] |= abi.NODE_MIG_MIXED
            if mig == "single":
                nflags[i] |= abi.NODE_MIG_SINGLE
        if nd.get("GpuMemorySynced") is not None or nd.get("GPUMemory"):
            pass  # only read by gpu-memory requests (unsupported here)

    # ---- topology domain tables (plugins/topology/topology_plugin.go:57-110, topology_structs.go:94-101)
    topo_level_off = [0]
    for lv in topo_levels:
        topo_level_off.append(topo_level_off[-1] + len(lv))
    node_domain = np.full((topo_level_off[-1], N), -1, np.int32)
    domain_level, domain_parent, domain_ids = [], [], []
    for t, levels in enumerate(topo_levels):
        ids = {}
        for i, nm in enumerate(node_names):
            labels = case["Nodes"][nm].get("Labels") or {}
            if not all(lv in labels for lv in levels):  # isNodePartOfTopology (common.go:70-77)
                continue
            parent = -1
            for l, lv in enumerate(levels):
                did = ".".join(labels[x] for x in levels[: l + 1])
                if (l, did) not in ids:
                    ids[(l, did)] = len(domain_level)
                    domain_level.append(topo_level_off[t] + l); domain_parent.append(parent); domain_ids.append((t, did))
                node_domain[topo_level_off[t] + l, i] = ids[(l, did)]
                parent = ids[(l, did)]
    domain_id_rank = np.zeros(len(domain_ids), np.uint32)
    for t in range(len(topo_levels)):
        idxs = [k for k, (tt, _) in enumerate(domain_ids) if tt == t]
        for k, r in zip(idxs, abi.rank_strings([domain_ids[k][1] for k in idxs])):
            domain_id_rank[k] = r

    def constraint(tc):
        """api/topology_info.TopologyConstraintInfo → (topology index | -1 none | -2 missing, required level, preferred level)"""
        if not tc or not tc.get("Topology"):
            return (-1, -1, -1)
        if tc["Topology"] not in topo_names:
            return (-2, -1, -1)
        t = topo_names.index(tc["Topology"])
        lv = lambda name: (topo_levels[t].index(name) if name in topo_levels[t] else 10 ** 6) if name else -1  # unknown level name ⇒ beyond the last level
        return (t, lv(tc.get("RequiredLevel")), lv(tc.get("PreferredLevel")))

    group_job, group_parent, group_names, group_tc, job_root_group, podset_group, podset_tc = [], [], [], [], [], [], []

    # ---- jobs & tasks (jobs_fake/jobs.go:51-330)
    if not isinstance(case.get("Jobs", []), list) or any(not isinstance(j, dict) for j in case.get("Jobs", [])):
        raise Unsupported("fixture built by Go code, not a literal")
    jobs = sorted(enumerate(case.get("Jobs", [])), key=lambda t: -int(t[1].get("Priority", 0)))  # SliceStable by priority desc
    jobs = [j for _, j in jobs]
    J = len(jobs)
    pod_names, pod_job, pod_podset, pod_status, pod_node, pod_flags, pod_prio, pod_aff = [], [], [], [], [], [], [], []
    pod_gpu_portion, pod_gpu_group, pod_gpu_memory = [], [], []
    req_rows = []
    podset_job, podset_min, podset_names_l, job_first_podset, job_n_podsets, job_first_pod, job_n_pods = [], [], [], [], [], [], []
    job_names, job_queue, job_priority, job_preempt, job_created = [], [], [], [], []
    trees = {}
    for ji, job in enumerate(jobs):
        tasks = job.get("Tasks", []) or []
        gmem = int(job.get("RequiredGpuMemory", 0) or 0)  # annotation gpu-memory (jobs.go:277): MiB of one device
        if gmem and not fractions:
            raise Unsupported("gpu memory request (oracle only, allocate action)")
        if job.get("RequiredMultiFractionDevicesPerTask") is not None:
            raise Unsupported("multi-fraction")
        g = float(job.get("RequiredGPUsPerTask", 0))
        frac = 0.0 < g < 1.0  # a fraction of one device (jobs.go:278-283 → annotation gpu-fraction)
        if gmem and g != 0:
            raise Unsupported("gpu memory request beside a GPU count")
        if g != int(g) and not frac:
            raise Unsupported("fractional gpu above one device")
        if frac and not fractions:
            raise Unsupported("fractional gpu (oracle only, allocate action)")
        job_names.append(job["Name"])
        job_queue.append(qidx.get(job.get("QueueName", ""), -1))
        if job_queue[-1] >= 0:  # input_jobs.go:53-59: the queue's parent must exist too (already pruned above)
            pass
        prio = int(job.get("Priority", 0))
        job_priority.append(prio)
        pre = job.get("Preemptibility", "")
        job_preempt.append(1 if pre == "preemptible" else 0 if pre == "non-preemptible" else (1 if prio < 100 else 0))
        age = int(job.get("JobAgeInMinutes", 0))
        job_created.append((-(age if age != 0 else (J - ji)) * 60_000_000_000) + ji)
        # sub-group tree: RootSubGroupSet (subgroup_info/subgroupset.go) or, by default, a root without constraint (jobs.go:116-123)
        root = job.get("RootSubGroupSet")
        sets = []  # (name, minAvailable, constraint, parent group)
        root_g = len(group_job)
        job_root_group.append(root_g)
        group_job.append(ji); group_parent.append(-1); group_names.append((ji, "")); group_tc.append(constraint(root.get("TopologyConstraint") if root else None))

        def walk(g, gi):
            for ps in g.get("PodSets", []) or []:
                sets.append((ps["Name"], int(ps["MinAvailable"]), constraint(ps.get("TopologyConstraint")), gi))
            for sg in g.get("SubGroups", []) or []:
                ci = len(group_job)
                group_job.append(ji); group_parent.append(gi); group_names.append((ji, sg.get("Name", ""))); group_tc.append(constraint(sg.get("TopologyConstraint")))
                walk(sg, ci)
        if root:
            walk(root, root_g)
            trees[job["Name"]] = root
        names_in_sets = [x[0] for x in sets]
        if any(not t.get("SubGroupName") for t in tasks) and "default" not in names_in_sets:
            sets.append(("default", len(tasks), (-1, -1, -1), root_g))  # jobs.go:116-123
        job_first_podset.append(len(podset_job)); job_n_podsets.append(len(sets))
        set_index = {}
        for n, m, tc, gi in sets:
            set_index[n] = len(podset_job)
            podset_job.append(ji); podset_min.append(int(m)); podset_names_l.append((ji, n)); podset_group.append(gi); podset_tc.append(tc)
        job_first_pod.append(len(pod_names)); job_n_pods.append(len(tasks))
        for ti, t in enumerate(tasks):
            if (t.get("RequiredMigInstances") or t.get("IsLegacyMigTask")) and not fractions:
                raise Unsupported("MIG task (oracle only, allocate action)")
            if t.get("PodAffinityLabels") or t.get("PodAffinityTopologyKey") or t.get("PodAntiAffinityTopologyKey"):
                raise Unsupported("inter-pod affinity")
            if t.get("ResourceClaimNames") or t.get("ResourceClaimTemplates"):
                raise Unsupported("DRA")
            groups = t.get("GPUGroups") or []
            shared = frac or gmem > 0
            if groups and not shared:
                groups = []  # a group label on a task that does not share a device: nothing reads it (node_info.go:457-493 looks at groups of fraction allocations only)
            if len(groups) > 1 or (groups and not shared) or any(not str(x).lstrip("-").isdigit() for x in groups):
                raise Unsupported("shared gpu groups beyond one numeric group of a fraction task")
            pod_gpu_portion.append(g if frac else 0.0); pod_gpu_memory.append(gmem)
            if frac and t.get("NodeName"):  # resourceFractionCalc (jobs.go:318-330): a placed fraction task without a group gets one of its own, named by a fresh UUID
                pod_gpu_group.append(int(groups[0]) if groups else NEW_GPU_GROUP + len(pod_gpu_group))
            elif gmem and t.get("NodeName"):  # a gpu-memory task keeps the groups its fixture names (jobs.go:277: the whole-GPU branch, no UUID)
                if not groups: raise Unsupported("placed gpu-memory task without a group")
                pod_gpu_group.append(int(groups[0]))
            else:
                pod_gpu_group.append(-1)
            pod_names.append(f"{job['Name']}-{ti}")
            pod_job.append(ji)
            sg = t.get("SubGroupName") or "default"
            if sg not in set_index:
                raise Unsupported("task sub-group missing")
            pod_podset.append(set_index[sg])
            st = t.get("State", "Pending")
            pod_status.append(S[st])
            nn = t.get("NodeName", "")
            pod_node.append(nidx.get(nn, -1) if nn else -1)
            fl = 0
            if t.get("IsLegacyMigTask"):
                fl |= abi.POD_LEGACY_MIG  # jobs_fake/jobs.go:213
            if t.get("Priority") is not None:
                fl |= abi.POD_HAS_TASK_PRIORITY
            pod_flags.append(fl); pod_prio.append(int(t["Priority"]) if t.get("Priority") is not None else 0)
            pod_aff.append(tuple(sorted(t.get("NodeAffinityNames") or [])))
            # CalcJobAndPodResources (jobs.go:216-246) + BuildResourceList (resources_fake/resources.go:32-68)
            if job.get("IsBestEffortJob"):
                cpu = mem = gp = 0.0
            else:
                cpu = _milli(job["RequiredCPUsPerTask"]) if job.get("RequiredCPUsPerTask", 0) != 0 else 1000.0
                mem = _value(job["RequiredMemoryPerTask"]) if job.get("RequiredMemoryPerTask", 0) != 0 else 1e9
                gp = g if frac else float(int(g))
            if t.get("RequiredGPUs") is not None:
                gp = float(int(t["RequiredGPUs"]))  # jobs.go:309-312
            req_rows.append((cpu, mem, gp, 1.0) + tuple(float((t.get("RequiredMigInstances") or {}).get(k, 0)) for k in mig_names))
    P = len(pod_names)
    pod_req = np.zeros((R, P))
    for p, row in enumerate(req_rows):
        pod_req[:, p] = row

    # ---- static predicate classes from NodeAffinityNames (tasks_fake/tasks.go:104-122: label kai.scheduler/type == node name)
    aff_sets = sorted(set(pod_aff))
    if aff_sets == [()] or not aff_sets:
        class_fit = np.ones((1, 1), np.uint8); pod_class = np.zeros(P, np.int32); node_class = np.zeros(N, np.int32)
    else:
        if () not in aff_sets:
            aff_sets = [()] + aff_sets
        class_fit = np.zeros((len(aff_sets), max(N, 1)), np.uint8)
        for ci, names in enumerate(aff_sets):
            for i, nm in enumerate(node_names):
                label = (case["Nodes"][nm].get("Labels") or {}).get("tasks_fake.NodeAffinityKey", nm)  # nodes_fake/nodes.go:187-191
                class_fit[ci, i] = 1 if (not names or label in names) else 0
        pod_class = np.array([aff_sets.index(a) for a in pod_aff], np.int32)
        node_class = np.arange(N, dtype=np.int32)

    # pod-set name rank inside the job
    ps_rank = np.zeros(len(podset_job), np.uint32)
    for ji in range(J):
        idxs = [k for k in range(len(podset_job)) if podset_job[k] == ji]
        ranks = abi.rank_strings([podset_names_l[k][1] for k in idxs])
        for k, r in zip(idxs, ranks):
            ps_rank[k] = r

    snap = abi.Snapshot(n_res=R)
    snap.node_names, snap.pod_names, snap.job_names, snap.queue_names = node_names, pod_names, job_names, qnames
    snap.podset_names = [n for (_, n) in podset_names_l]
    a = snap.arrays
    a["node_allocatable"] = alloc; a["node_flags"] = nflags; a["node_gpu_count"] = gpu_count
    a["node_name_rank"] = abi.rank_strings(node_names); a["node_class"] = node_class
    a["pod_req"] = pod_req; a["pod_job"] = np.array(pod_job, np.int32); a["pod_podset"] = np.array(pod_podset, np.int32)
    a["pod_status"] = np.array(pod_status, np.int32); a["pod_node"] = np.array(pod_node, np.int32)
    a["pod_flags"] = np.array(pod_flags, np.uint32); a["pod_task_priority"] = np.array(pod_prio, np.int32)
    a["pod_created_ns"] = np.zeros(P, np.int64); a["pod_uid_rank"] = abi.rank_strings(pod_names); a["pod_class"] = pod_class
    a["pod_nominated_node"] = np.full(P, -1, np.int32)
    if mig_names:  # nvidia.com/mig-<g>g.<m>gb (api/common_info/resources/mig.go:13-33)
        mg = np.zeros(R, np.int32); mm = np.zeros(R, np.int64)
        for k, r in mig_row.items():
            m = _re.match(r"^nvidia.com/mig-(\d+)g\.(\d+)gb$", k)
            if not m: raise Unsupported("MIG profile name")
            mg[r], mm[r] = int(m.group(1)), int(m.group(2))
        a["res_mig_gpus"] = mg; a["res_mig_memory"] = mm
    if any(x > 0 for x in pod_gpu_portion) or any(pod_gpu_memory):  # shared GPUs (ABI v4; gpu-memory requests: v5)
        a["pod_gpu_portion"] = np.array(pod_gpu_portion, np.float64); a["pod_gpu_group"] = np.array(pod_gpu_group, np.int32)
        if any(pod_gpu_memory): a["pod_gpu_memory"] = np.array(pod_gpu_memory, np.int64)
        gm = []
        for nm in node_names:  # nodes_fake/nodes.go:184-194 + getNodeGpuMemory (node_info.go:673-687): MiB, floored to a multiple of 100
            v = int(case["Nodes"][nm].get("GPUMemory") or 100); gm.append(v - v % 100)
        a["node_gpu_memory"] = np.array(gm, np.int64)
    a["podset_job"] = np.array(podset_job, np.int32); a["podset_min_available"] = np.array(podset_min, np.int32); a["podset_name_rank"] = ps_rank
    a["job_queue"] = np.array(job_queue, np.int32); a["job_priority"] = np.array(job_priority, np.int32)
    a["job_preemptible"] = np.array(job_preempt, np.int32); a["job_created_ns"] = np.array(job_created, np.int64)
    a["job_uid_rank"] = abi.rank_strings(job_names)
    a["job_first_pod"] = np.array(job_first_pod, np.int32); a["job_n_pods"] = np.array(job_n_pods, np.int32)
    a["job_first_podset"] = np.array(job_first_podset, np.int32); a["job_n_podsets"] = np.array(job_n_podsets, np.int32)
    a["queue_parent"] = np.array([qidx.get(r["parent"], -1) if r["parent"] else -1 for r in qrec], np.int32)
    a["queue_priority"] = np.array([r["priority"] if r["priority"] is not None else 100 for r in qrec], np.int32)  # constants.DefaultQueuePriority
    a["queue_created_ns"] = np.array([r["created"] for r in qrec], np.int64)
    a["queue_uid_rank"] = abi.rank_strings(qnames)
    a["queue_deserved"] = np.array([[r["deserved"][k] for r in qrec] for k in range(3)], np.float64).reshape(3, Q)
    a["queue_limit"] = np.array([[r["limit"][k] for r in qrec] for k in range(3)], np.float64).reshape(3, Q)
    a["queue_oqw"] = np.array([[r["oqw"][k] for r in qrec] for k in range(3)], np.float64).reshape(3, Q)
    a["class_fit"] = class_fit
    # topology + sub-group tree (ABI v2)
    G = len(group_job)
    g_rank = np.zeros(G, np.uint32)
    for ji in range(J):
        idxs = [k for k in range(G) if group_job[k] == ji]
        for k, r in zip(idxs, abi.rank_strings([group_names[k][1] for k in idxs])):
            g_rank[k] = r
    a["topo_level_off"] = np.array(topo_level_off, np.int32); a["node_domain"] = node_domain
    a["domain_level"] = np.array(domain_level, np.int32); a["domain_parent"] = np.array(domain_parent, np.int32); a["domain_id_rank"] = domain_id_rank
    a["group_job"] = np.array(group_job, np.int32); a["group_parent"] = np.array(group_parent, np.int32); a["group_name_rank"] = g_rank
    a["group_topology"] = np.array([t[0] for t in group_tc], np.int32); a["group_required_level"] = np.array([t[1] for t in group_tc], np.int32)
    a["group_preferred_level"] = np.array([t[2] for t in group_tc], np.int32); a["job_root_group"] = np.array(job_root_group, np.int32)
    a["podset_group"] = np.array(podset_group, np.int32); a["podset_topology"] = np.array([t[0] for t in podset_tc], np.int32)
    a["podset_required_level"] = np.array([t[1] for t in podset_tc], np.int32); a["podset_preferred_level"] = np.array([t[2] for t in podset_tc], np.int32)
    snap.finalize()
    meta = dict(name=case.get("Name"), line=case.get("_line"), expected_jobs=case.get("JobExpectedResults") or {},
                expected_tasks=case.get("TaskExpectedResults") or {}, expected_nodes=case.get("ExpectedNodesResources") or {})
    return snap, cfg, meta


NEW_GPU_GROUP = 1 << 20  # oracle: ids from here on are groups created by the run (a UUID in the reference)


def check_expectations(snap, meta, pod_status, pod_node, nodes=None, gpu_groups=None):
    """test_utils.MatchExpectedAndRealTasks (test_utils/test_utils.go:121-314): status + node per task. → list of mismatches.
    GPU groups (:150-175): the reference maps every expected group name of a node to the actual group the first time it sees it and requires
    consistency afterwards — and only looks at all when expected and actual names are literally equal, which a freshly drawn UUID never is.
    Checked here a little more strictly: a task that lands on a group of the fixture must be on the one named; new groups must map to the
    expected names consistently per node."""
    S = abi.POD_STATUS
    errs = []
    seen = {}  # (node, expected name) → actual id
    if gpu_groups is not None and "pod_gpu_group" in snap.arrays:
        for p in range(snap.n_pods):
            g = int(snap.pod_gpu_group[p])
            if g >= 0 and snap.pod_node[p] >= 0: seen[(int(snap.pod_node[p]), str(g))] = g
    # An expectation whose own labels overbook a device cannot be met by anything (consolidationGpuMemory_test.go:38-117 names group "0" for 30 + 80 of
    # a 100 MiB device); the reference's literal-equality rule lets it through, and so does this check — for that scenario only.
    labels_ok = True
    if gpu_groups is not None and "pod_gpu_portion" in snap.arrays:
        load = {}
        for jname, exp in meta["expected_jobs"].items():
            if jname not in snap.job_names or not exp.get("GPUGroups") or not exp.get("NodeName") or exp["NodeName"] not in snap.node_names: continue
            want = S[exp.get("Status", "Pending")] if isinstance(exp.get("Status", "Pending"), str) else exp.get("Status")
            if not (want & abi.ACTIVE_USED) or want == S["Releasing"]: continue  # a pipelined task may share a device with the releasing one it waits for
            j = snap.job_names.index(jname); ni = snap.node_names.index(exp["NodeName"])
            for p in range(int(snap.job_first_pod[j]), int(snap.job_first_pod[j]) + int(snap.job_n_pods[j])):
                f = float(snap.pod_gpu_portion[p])
                if "pod_gpu_memory" in snap.arrays and snap.pod_gpu_memory[p] > 0 and snap.node_gpu_memory[ni] > 0: f = float(snap.pod_gpu_memory[p]) / float(snap.node_gpu_memory[ni])
                if f > 0: load[(ni, str(exp["GPUGroups"][0]))] = load.get((ni, str(exp["GPUGroups"][0])), 0.0) + f
        labels_ok = all(v <= 1.0 + 1e-9 for v in load.values())
        # … the one scenario named above, nothing else: an overbooked expectation anywhere else is a broken fixture
        if not labels_ok and meta.get("name") != "consolidate two memory jobs to free gpu":
            errs.append("the expectations book more than one device's worth on a shared GPU group: " + ", ".join(f"{k}: {v:.2f}" for k, v in load.items() if v > 1.0 + 1e-9))
    for jname, exp in meta["expected_jobs"].items():
        if jname not in snap.job_names:
            errs.append(f"job {jname} missing"); continue
        j = snap.job_names.index(jname)
        first, n = int(snap.job_first_pod[j]), int(snap.job_n_pods[j])
        gsum = 0.0
        for p in range(first, first + n):
            want = S[exp.get("Status", "Pending")] if isinstance(exp.get("Status", "Pending"), str) else exp.get("Status")
            if int(pod_status[p]) != want:
                errs.append(f"{snap.pod_names[p]}: status {abi.POD_STATUS_NAME.get(int(pod_status[p]))} want {exp.get('Status')}")
            if exp.get("NodeName"):
                got = snap.node_names[pod_node[p]] if pod_node[p] >= 0 else ""
                if got != exp["NodeName"]:
                    errs.append(f"{snap.pod_names[p]}: node {got!r} want {exp['NodeName']!r}")
            gsum += snap.pod_req[abi.RES_GPU, p]
            if gpu_groups is not None and labels_ok and exp.get("GPUGroups") and not exp.get("DontValidateGPUGroup") and (int(pod_status[p]) & abi.ACTIVE_USED) and "pod_gpu_portion" in snap.arrays and (snap.pod_gpu_portion[p] > 0 or ("pod_gpu_memory" in snap.arrays and snap.pod_gpu_memory[p] > 0)):
                name, actual, key = str(exp["GPUGroups"][0]), int(gpu_groups[p]), (int(pod_node[p]), str(exp["GPUGroups"][0]))
                if actual < NEW_GPU_GROUP:  # landed on a group of the fixture: it must be the one named
                    if str(actual) != name: errs.append(f"{snap.pod_names[p]}: gpu group {actual} want {name!r}")
                elif key not in seen: seen[key] = actual  # a new group: the same one every time this name appears on the node …
                elif seen[key] >= NEW_GPU_GROUP and seen[key] != actual: errs.append(f"{snap.pod_names[p]}: gpu group {actual} want {name!r} = {seen[key]}")
                # … unless the name is a group of the fixture: the reference's check compares literal names only, so a fresh group passes it (e.g. line 1369)
        if abs(gsum - float(exp.get("GPUsRequired", 0))) > 1e-9:
            errs.append(f"{jname}: GPUsRequired {gsum} want {exp.get('GPUsRequired', 0)}")
    for tname, exp in meta["expected_tasks"].items():
        if tname not in snap.pod_names:
            continue
        p = snap.pod_names.index(tname)
        if int(pod_status[p]) != S[exp.get("Status", "Pending")]:
            errs.append(f"{tname}: status {abi.POD_STATUS_NAME.get(int(pod_status[p]))} want {exp.get('Status')}")
        if exp.get("NodeName"):
            got = snap.node_names[pod_node[p]] if pod_node[p] >= 0 else ""
            if got != exp["NodeName"]:
                errs.append(f"{tname}: node {got!r} want {exp['NodeName']!r}")
    if nodes is not None:
        for nname, exp in meta["expected_nodes"].items():
            if nname not in snap.node_names:
                errs.append(f"node {nname} missing"); continue
            i = snap.node_names.index(nname)
            if nodes["releasing"][i, abi.RES_GPU] != float(exp.get("ReleasingGPUs", 0)):
                errs.append(f"{nname}: releasing gpus {nodes['releasing'][i, abi.RES_GPU]} want {exp.get('ReleasingGPUs', 0)}")
            if nodes["idle"][i, abi.RES_GPU] != float(exp.get("IdleGPUs", 0)):
                errs.append(f"{nname}: idle gpus {nodes['idle'][i, abi.RES_GPU]} want {exp.get('IdleGPUs', 0)}")
    return errs


# ------------------------------------------------------------------------------------------------ integration tests (several rounds)
CYCLE = ("allocate", "consolidation", "reclaim", "preempt")  # the default action list minus stalegangeviction (checked to be a no-op below)


def _stale_gang(snap, pod_status):
    """PodGroupInfo.IsStale (api/podgroup_info/job_info.go:417-432): some pods active, a pod-set below minAvailable, nothing Succeeded."""
    S = abi.POD_STATUS
    active = S["Allocated"] | S["Pipelined"] | S["Binding"] | S["Bound"] | S["Running"] | S["Releasing"]
    for j in range(snap.n_jobs):
        b, n = int(snap.job_first_pod[j]), int(snap.job_n_pods[j])
        st = pod_status[b:b + n]
        if (st == S["Succeeded"]).any() or not ((st & active) != 0).any():
            continue
        for k in range(int(snap.job_first_podset[j]), int(snap.job_first_podset[j] + snap.job_n_podsets[j])):
            used = int((((st & active) != 0) & (snap.pod_podset[b:b + n] == k)).sum())
            if used < int(snap.podset_min_available[k]):
                return True
    return False


def run_integration(case, run_fn, rounds_after=None, fractions=False):
    """actions/integration_tests/integration_tests_utils/integration_tests_utils.go:40-156: `RoundsUntilMatch` scheduling cycles, each on a
    session rebuilt from the fixture with the previous cycle's outcome fed back (Binding -> Running, Pipelined / Releasing -> Pending),
    then the expectations on a rebuilt session and after each of `RoundsAfterMatch` further cycles.  Returns the list of mismatches."""
    import copy
    case = copy.deepcopy(case)

    def one_round():
        snap, cfg, meta = case_to_snapshot(case, CYCLE, fractions)
        res = run_fn(snap, cfg, CYCLE)
        if _stale_gang(snap, res.pod_status):
            raise Unsupported("stalegangeviction would act (not a placement action)")
        for job in case["Jobs"]:
            for ti, t in enumerate(job.get("Tasks", []) or []):
                p = snap.pod_names.index(f"{job['Name']}-{ti}")
                st = abi.POD_STATUS_NAME[int(res.pod_status[p])]
                node = snap.node_names[res.pod_node[p]] if res.pod_node[p] >= 0 else ""
                if st == "Releasing":
                    if job.get("DeleteJobInTest"):
                        t["NodeName"] = node; t["State"] = "Releasing"
                    else:
                        t["NodeName"] = ""; t["State"] = "Pending"
                elif st == "Pipelined":
                    t["NodeName"] = ""; t["State"] = "Pending"
                elif st == "Binding":
                    t["State"] = "Running"; t["NodeName"] = node
                else:
                    t["State"] = st; t["NodeName"] = node
                if "pod_gpu_portion" in snap.arrays and (snap.pod_gpu_portion[p] > 0 or ("pod_gpu_memory" in snap.arrays and snap.pod_gpu_memory[p] > 0)):  # the shared-GPU group travels with the placed pod (label runai-gpu-group)
                    g = int(getattr(res, "gpu_groups", np.full(snap.n_pods, -1))[p])
                    if t["NodeName"] and g >= 0: t["GPUGroups"] = [str(g)]
                    else: t.pop("GPUGroups", None)
        return snap, meta, res

    for _ in range(int(case.get("_RoundsUntilMatch", 2))):
        one_round()
    snap, cfg, meta = case_to_snapshot(case, CYCLE, fractions)  # prepareSessionForMatch: a rebuilt session, nothing run on it
    res = run_fn(snap, cfg, ())
    errs = check_expectations(snap, meta, res.pod_status, res.pod_node, res.nodes)
    for _ in range(int(case.get("_RoundsAfterMatch", 5)) if rounds_after is None else rounds_after):
        snap, meta, res = one_round()
        errs += check_expectations(snap, meta, res.pod_status, res.pod_node, res.nodes)
    return errs


def broad_case(seed):
    """One seed of the broad randomized campaign (the one that found the staged-path / canonical-node-order / stale-cache / leaf-heap
    divergences): a crowded cluster with every optional feature drawn at random, and a baseline-style snapshot, each under several action orders."""
    synth = pkg.synth
    rng = np.random.default_rng(seed)
    cases = []
    s1 = synth.make_crowded_snapshot(4 + seed % 29, 5000 + seed, fill=0.7 + 0.25 * rng.random(), queue_levels=((2, 2), (3,), (2, 2, 2), (4, 3))[seed % 4], hog_frac=rng.random(),
                                     elastic_frac=rng.random() * 0.6, nonpreempt_frac=rng.random() * 0.3, cpu_only_frac=0.3 if seed % 4 == 0 else 0.0, n_pending_jobs=5 + seed % 25,
                                     minruntime=seed % 3 == 0, two_podsets_frac=0.5 if seed % 2 else 0.0)
    if seed % 2:
        synth.add_replica_topology(s1, seed, zones=1 + seed % 3, nodes_per_rack=2 + seed % 3)
    c1 = abi.default_config(max_consolidation_preemptees=(-1, 16, 2, 0)[seed % 4], gpu_strategy=(abi.BINPACK, abi.SPREAD)[seed % 2], cpu_strategy=(abi.BINPACK, abi.SPREAD)[(seed // 2) % 2],
                            k_value=(0.0, 0.5, 1.0)[seed % 3])
    c1.use_scheduling_signatures = seed % 2; c1.allow_consolidating_reclaim = int(seed % 3 != 0); c1.reclaimer_saturation_multiplier = (1.0, 1.2, 2.0)[seed % 3]
    c1.now_ns = synth.NOW_NS; c1.default_preempt_min_runtime_ns = (0, 900 * 10**9)[seed % 2]; c1.default_reclaim_min_runtime_ns = (0, 600 * 10**9)[(seed // 2) % 2]; c1.reclaim_resolve_method = seed % 2
    for acts in (("allocate", "consolidation", "reclaim", "preempt"), ("reclaim", "preempt", "consolidation", "allocate"), ("preempt",), ("consolidation",)):
        cases.append((s1, c1, acts))
    s2 = synth.make_snapshot(int(rng.integers(2, 60)), int(rng.integers(0, 400)), 7000 + seed, queue_levels=((2, 3), (3, 2, 2), (1, 4))[seed % 3], prefill=float(rng.random()) * 0.9,
                             gpu_mix=((8, .5), (4, .3), (0, .2)), cpu_only_frac=0.3, zipf=True, limits_frac=0.3, queue_prios=(100, 200), oqws=(1.0, 2.0), nonpreempt_frac=0.2,
                             usage_max=0.2, lexi_names=bool(seed % 2), elastic_frac=0.3, multi_podset_frac=0.3, task_prio_frac=0.2)
    if seed % 2 == 0:
        synth.add_topology(s2, seed, zones=1 + seed % 4, racks_per_zone=1 + seed % 5, req_rack_frac=0.4, pref_rack_frac=0.3)
    c2 = abi.default_config(gpu_strategy=(abi.BINPACK, abi.SPREAD)[seed % 2], k_value=float(seed % 3) * 0.5, max_consolidation_preemptees=8)
    cases.append((s2, c2, ("allocate",))); cases.append((s2, c2, ("allocate", "consolidation", "reclaim", "preempt")))
    return cases
