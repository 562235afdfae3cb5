"""Snapshot ingest (SURVEY.md §8f n1 + n4): snapshot.json / snapshot.zip → kai_snapshot_soa.

Pinned on the reference's own tests where they exist (transcribed by hand, source lines cited), on known answers of
k8s.io/apimachinery's resource.Quantity, and on round trips: every synthetic snapshot written as reference-schema objects and ingested
again must give back the same arrays and the same scheduling result.
"""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import kai_testlib as T

pkg = T.pkg
ing = pkg.ingest
abi = pkg.abi
E = ing._EPOCH_NS
ST = abi.POD_STATUS


# ------------------------------------------------------------------------------------------------ ABI
def test_ingest_library_exports_header_symbols():
    hdr = open(os.path.join(T.ROOT, "include", "kai_ingest.h")).read()
    syms = sorted(set(re.findall(r"^(?:int|void|const char\*|const kai_[a-z_]+\*)\s+(kai_[a-z_]+)\s*\(", hdr, flags=re.M)))
    lib = ing.load_ingest_library()
    assert set(syms) == set(ing.EXPORTS), (syms, ing.EXPORTS)
    for s in syms:
        assert hasattr(lib, s), s


# ------------------------------------------------------------------------------------------------ resource.Quantity
# k8s.io/apimachinery pkg/api/resource: MilliValue() / Value() round up (quantity.go); suffix table of suffix.go
QUANTITIES = [
    ("0", 0, 0), ("1", 1000, 1), ("100m", 100, 1), ("1500m", 1500, 2), ("0.1", 100, 1), ("2.5", 2500, 3), ("1k", 1_000_000, 1000),
    ("20000m", 20000, 20), ("1G", 10**12, 10**9), ("20G", 2 * 10**13, 2 * 10**10), ("1Gi", 2**30 * 1000, 2**30), ("1.5Gi", 1610612736000, 1610612736),
    ("128974848", 128974848000, 128974848), ("129e6", 129 * 10**9, 129 * 10**6), ("123Mi", 123 * 2**20 * 1000, 123 * 2**20), ("1e3", 10**6, 1000),
    ("1E3", 10**6, 1000), ("5Ki", 5120000, 5120), ("1u", 1, 1), ("1n", 1, 1), ("999u", 1, 1), ("1001u", 2, 1), ("0.0005", 1, 1), ("1T", 10**15, 10**12),
    ("1Ti", 2**40 * 1000, 2**40), ("12e-1", 1200, 2), ("+3", 3000, 3), ("1.000", 1000, 1), ("110", 110000, 110),
]


@pytest.mark.parametrize("text,milli,value", QUANTITIES)
def test_quantity_known_answers(text, milli, value):
    lib = ing.load_ingest_library()
    out = C.c_int64()
    assert lib.kai_quantity_milli(text.encode(), C.byref(out)) == 0 and out.value == milli
    assert lib.kai_quantity_value(text.encode(), C.byref(out)) == 0 and out.value == value


@pytest.mark.parametrize("text", ["", "abc", "1.2.3", "1Zi", "1e", "--1", "1 Gi"])
def test_quantity_rejects_malformed(text):
    lib = ing.load_ingest_library()
    out = C.c_int64()
    assert lib.kai_quantity_milli(text.encode(), C.byref(out)) == -1


# ------------------------------------------------------------------------------------------------ document helpers
def doc(pods=(), nodes=(), queues=(), pod_groups=(), params=None, config=None, **raw):
    p = {"schedulerName": "kai-scheduler", "fullHierarchyFairness": True, "useSchedulingSignatures": True}
    p.update(params or {})
    r = {"pods": list(pods), "nodes": list(nodes), "queues": list(queues), "podGroups": list(pod_groups)}
    r.update(raw)
    return {"config": config if config is not None else {"actions": "allocate"}, "schedulerParams": p, "rawObjects": r}


def node(name, cpu="10", mem="64Gi", gpu="8", pods="110", labels=None, taints=None, **extra):
    alloc = {"cpu": cpu, "memory": mem, "pods": pods}
    if gpu is not None: alloc["nvidia.com/gpu"] = gpu
    alloc.update(extra.pop("alloc", {}))
    n = {"metadata": {"name": name, "labels": labels or {}}, "spec": {}, "status": {"allocatable": alloc}}
    if taints: n["spec"]["taints"] = taints
    n["spec"].update(extra.pop("spec", {}))
    n["status"].update(extra.pop("status", {}))
    return n


def queue(name, parent=None, gpu_quota=-1, labels=None, **spec):
    s = {"resources": {k: {"quota": -1, "limit": -1, "overQuotaWeight": 1} for k in ("cpu", "memory", "gpu")}}
    s["resources"]["gpu"]["quota"] = gpu_quota
    if parent: s["parentQueue"] = parent
    s.update(spec)
    return {"metadata": {"name": name, "labels": labels or {}, "creationTimestamp": "2024-01-01T00:00:00Z"}, "spec": s}


def pod_group(name, queue_name="q", min_member=1, **spec):
    s = {"queue": queue_name, "minMember": min_member}
    s.update(spec)
    return {"metadata": {"name": name, "namespace": "ns", "creationTimestamp": "2024-01-01T00:00:00Z"}, "spec": s}


def pod(name, group=None, requests=None, phase="Pending", node_name=None, **over):
    md = {"name": name, "namespace": "ns", "uid": "uid-" + name, "creationTimestamp": "2024-01-01T00:00:00Z", "labels": {}, "annotations": {}}
    if group: md["annotations"]["pod-group-name"] = group
    spec = {"schedulerName": "kai-scheduler", "containers": [{"name": "c", "resources": {"requests": requests if requests is not None else {"cpu": "1", "memory": "1Gi", "nvidia.com/gpu": "1"}}}]}
    if node_name: spec["nodeName"] = node_name
    md["labels"].update(over.pop("labels", {})); md["annotations"].update(over.pop("annotations", {}))
    md.update(over.pop("metadata", {})); spec.update(over.pop("spec", {}))
    return {"metadata": md, "spec": spec, "status": dict({"phase": phase}, **over.pop("status", {}))}


def ingest(d, **kw):
    return ing.ingest_json(json.dumps(d), **kw)


def rl(cpu, mem, gpu=None):
    r = {"cpu": cpu, "memory": mem}
    if gpu is not None: r["nvidia.com/gpu"] = gpu
    return r


# ------------------------------------------------------------------------------------------------ reference known answers
def test_pod_resource_request_reference_cases():
    """api/pod_info/pod_info_test.go:42-165 TestGetPodResourceRequest (4 cases): sum of containers, max with each init container,
    + overhead; expected (gpu, milli-cpu, memory bytes) as the test states them."""
    cases = [
        (dict(containers=[rl("1000m", "1G"), rl("2000m", "1G")]), (0, 3000, 2e9)),
        (dict(init=[rl("2000m", "5G"), rl("2000m", "1G")], containers=[rl("1000m", "1G"), rl("2000m", "1G")]), (0, 3000, 5e9)),
        (dict(init=[rl("2000m", "5G"), rl("2000m", "1G")], containers=[rl("1000m", "1G", "1"), rl("2000m", "1G")]), (1, 3000, 5e9)),
        (dict(containers=[rl("1000m", "1G", "1"), rl("2000m", "1G")], overhead=rl("1000m", "1G")), (1, 4000, 3e9)),
    ]
    pods = []
    for i, (c, _) in enumerate(cases):
        spec = {"containers": [{"name": f"c{k}", "resources": {"requests": r}} for k, r in enumerate(c["containers"])]}
        if "init" in c: spec["initContainers"] = [{"name": f"i{k}", "resources": {"requests": r}} for k, r in enumerate(c["init"])]
        if "overhead" in c: spec["overhead"] = c["overhead"]
        pods.append(pod(f"p{i}", spec=spec))
    got = ingest(doc(pods=pods)).snapshot
    for i, (_, (gpu, cpu, mem)) in enumerate(cases):
        assert tuple(got.pod_req[:, i]) == (cpu, mem, gpu, 1.0), i  # pods := 1 (pod_info.go:390)


def test_queue_hierarchy_reference_case():
    """cache/cluster_info/queue_test.go:15-54 TestUpdateQueueHierarchySetChildQueues: the orphan goes, six queues stay."""
    qs = [queue("queue1", "dep1"), queue("dep1"), queue("queue2", "dep2"), queue("queue3", "dep2"), queue("dep2"), queue("dep3"), queue("orphan", "unexisting")]
    got = ingest(doc(queues=qs))
    s = got.snapshot
    assert s.n_queues == 6 and "orphan" not in s.queue_names
    par = {n: (s.queue_names[p] if p >= 0 else None) for n, p in zip(s.queue_names, s.queue_parent)}
    assert par == {"queue1": "dep1", "dep1": None, "queue2": "dep2", "queue3": "dep2", "dep2": None, "dep3": None}
    assert any("orphan" in w for w in got.warnings)
    # an orphan's whole subtree goes with it (queue.go:117-129)
    got = ingest(doc(queues=[queue("a", "missing"), queue("b", "a"), queue("c", "b"), queue("d")]))
    assert got.snapshot.queue_names == ["d"]


def test_flat_hierarchy_reference_case():
    """cache/cluster_info/cluster_info_test.go:1392-1481 TestSnapshotFlatHierarchy: without full-hierarchy fairness the parents are replaced by
    one unlimited "default" queue; partition label filter on the queues."""
    lab = {"pool": "nodepool-a"}
    dep = lambda n: dict(queue(n, labels=lab), spec={"resources": {"gpu": {"quota": 4, "overQuotaWeight": 2, "limit": 10}}})
    qs = [dep("department0"), dep("department1"), queue("queue0", "department0", labels=lab), queue("queue1", "department1", labels=lab),
          queue("elsewhere", "department0", labels={"pool": "nodepool-b"})]
    got = ingest(doc(queues=qs, params={"fullHierarchyFairness": False, "partitionParams": {"NodePoolLabelKey": "pool", "NodePoolLabelValue": "nodepool-a"}}))
    s = got.snapshot
    assert sorted(s.queue_names) == ["default", "queue0", "queue1"]
    d = s.queue_names.index("default")
    assert s.queue_parent[d] == -1 and all(s.queue_parent[i] == d for i in range(3) if i != d)
    assert list(s.queue_deserved[:, d]) == [-1, -1, -1] and list(s.queue_limit[:, d]) == [-1, -1, -1] and list(s.queue_oqw[:, d]) == [1, 1, 1]


def test_node_accounting_reference_case():
    """cache/cluster_info/cluster_info_test.go:244-310 TestSnapshotNodes "BasicUsage": 10 cpu / 110 pods, one Running 2-cpu pod →
    idle 8 cpu / 109 pods, used 2 cpu / 1 pod.  The accounting itself is the oracle's; the ingest supplies the quantities."""
    d = doc(nodes=[node("node-1", cpu="10", mem="0", gpu=None)], pods=[pod("p", requests={"cpu": "2"}, phase="Running", node_name="node-1")])
    got = ingest(d)
    s = got.snapshot
    assert s.pod_status[0] == ST["Running"] and s.pod_node[0] == 0 and s.pod_job[0] == -1
    res = T.Oracle.run(s, got.config, ())
    assert res.nodes["idle"][0, 0] == 8000 and res.nodes["idle"][0, 3] == 109 and res.nodes["used"][0, 0] == 2000 and res.nodes["used"][0, 3] == 1


def test_priority_and_preemptibility_rules():
    """cluster_info.go:495-531 (priority class by name, else the globalDefault class, else 50) and pkg/common/podgroup/preemptible.go:10-26."""
    pcs = [{"metadata": {"name": "train"}, "value": 50}, {"metadata": {"name": "build"}, "value": 100}, {"metadata": {"name": "dflt"}, "value": 75, "globalDefault": True}]
    pgs = [pod_group("a", priorityClassName="train"), pod_group("b", priorityClassName="build"), pod_group("c", priorityClassName="nope"), pod_group("d"),
           pod_group("e", priorityClassName="build", preemptibility="preemptible"), pod_group("f", priorityClassName="train", preemptibility="non-preemptible"),
           pod_group("g", queue_name="missing", priorityClassName="build")]
    s = ingest(doc(queues=[queue("q")], pod_groups=pgs, priorityClasses=pcs)).snapshot
    assert list(s.job_priority) == [50, 100, 75, 75, 100, 50, 0]
    assert list(s.job_preemptible) == [1, 0, 1, 1, 1, 0, 0]
    assert list(s.job_queue) == [0] * 6 + [-1]
    s = ingest(doc(queues=[queue("q")], pod_groups=[pod_group("a")])).snapshot
    assert list(s.job_priority) == [50]  # DefaultPodGroupPriority


def test_pod_status_rules():
    """api/pod_info/pod_info.go:410-445 getTaskStatus + NodeName from the bind request (:176-179); failed bind requests are ignored
    (bindrequest_info.go:84-92) and so are those for unknown nodes (cluster_info.go:337-345)."""
    del_ts = {"deletionTimestamp": "2024-01-02T00:00:00Z"}
    pods = [pod("running", phase="Running", node_name="n0"), pod("releasing", phase="Running", node_name="n0", metadata=del_ts),
            pod("pend-del", metadata=del_ts), pod("bound", node_name="n0"), pod("binding"), pod("bind-failed"), pod("bind-retry"), pod("bind-nonode"),
            pod("gated", spec={"schedulingGates": [{"name": "g"}]}), pod("pending"), pod("succeeded", phase="Succeeded", node_name="n0"),
            pod("failed", phase="Failed"), pod("unknown", phase="Unknown"), pod("weird", phase="")]
    brs = [{"metadata": {"namespace": "ns"}, "spec": {"podName": "binding", "selectedNode": "n0"}},
           {"metadata": {"namespace": "ns"}, "spec": {"podName": "bind-failed", "selectedNode": "n0"}, "status": {"phase": "Failed"}},
           {"metadata": {"namespace": "ns"}, "spec": {"podName": "bind-retry", "selectedNode": "n0", "backoffLimit": 3}, "status": {"phase": "Failed", "failedAttempts": 1}},
           {"metadata": {"namespace": "ns"}, "spec": {"podName": "bind-nonode", "selectedNode": "gone"}}]
    s = ingest(doc(nodes=[node("n0")], pods=pods, bindRequests=brs)).snapshot
    names = [n.split("/")[1] for n in s.pod_names]
    st = {n: (int(s.pod_status[i]), int(s.pod_node[i])) for i, n in enumerate(names)}
    assert st == {"running": (ST["Running"], 0), "releasing": (ST["Releasing"], 0), "pend-del": (ST["Releasing"], -1), "bound": (ST["Bound"], 0),
                  "binding": (ST["Binding"], 0), "bind-failed": (ST["Pending"], -1), "bind-retry": (ST["Binding"], 0), "bind-nonode": (ST["Pending"], -1),
                  "gated": (ST["Gated"], -1), "pending": (ST["Pending"], -1), "succeeded": (ST["Succeeded"], 0), "failed": (ST["Failed"], -1),
                  "unknown": (ST["Unknown"], -1), "weird": (ST["Unknown"], -1)}


def test_node_rules():
    """scheduler_util/scheduler_utils.go:12-40 (conditions), node_info.go:619-640 (gpu.count), :704-732 (MIG), resource_info.go:53-79 (units),
    cluster_info.go:533-549 (restrictSchedulingNodes keeps labelled nodes only)."""
    cond = lambda t, s: {"type": t, "status": s}
    nodes = [node("a", cpu="64", mem="512Gi", gpu="8", labels={"nvidia.com/gpu.count": "8", "node-role.kubernetes.io/gpu-worker": ""}, status={"conditions": [cond("Ready", "True"), cond("MemoryPressure", "False")]}),
             node("b", cpu="1500m", mem="1G", gpu=None, labels={"node-role.kubernetes.io/cpu-worker": "x"}, status={"conditions": [cond("Ready", "False")]}),
             node("c", spec={"unschedulable": True}), node("d", status={"conditions": [cond("Ready", "True"), cond("DiskPressure", "Unknown")]}),
             node("e", labels={"node-role.kubernetes.io/mig-enabled": "true", "nvidia.com/mig.strategy": "mixed", "nvidia.com/gpu.count": "x"}),
             node("f", alloc={"nvidia.com/mig-1g.5gb": "7"}), node("g", gpu=None, alloc={"amd.com/gpu": "4", "example.com/foo": "3", "ephemeral-storage": "10Gi"})]
    s = ingest(doc(nodes=nodes, pods=[pod("p", requests={"example.com/foo": "2", "ephemeral-storage": "1Gi", "cpu": "1"})])).snapshot
    f = dict(zip(s.node_names, (int(x) for x in s.node_flags)))
    assert f["a"] == abi.NODE_GPU_WORKER and f["b"] == abi.NODE_NOT_READY | abi.NODE_CPU_WORKER and f["c"] == abi.NODE_NOT_READY and f["d"] == abi.NODE_NOT_READY
    assert f["e"] == abi.NODE_MIG_ENABLED | abi.NODE_MIG_MIXED and f["f"] == abi.NODE_MIG_ENABLED and f["g"] == 0
    assert list(s.node_gpu_count) == [8, -1, -1, -1, -1, -1, -1]
    assert list(s.node_allocatable[:4, 0]) == [64000, 512 * 2**30, 8, 110] and list(s.node_allocatable[:4, 1]) == [1500, 1e9, 0, 110]
    assert s.node_allocatable[2, 6] == 4  # amd.com/gpu counts as gpu (resource_requirment.go:17-18)
    # scalar columns appear only for resources some pod requests, alphabetical: ephemeral-storage → Value, extended → MilliValue, on both sides
    got = ingest(doc(nodes=nodes, pods=[pod("p", requests={"example.com/foo": "2", "ephemeral-storage": "1Gi", "cpu": "1"})]))
    assert got.resource_names == ["cpu", "memory", "gpu", "pods", "ephemeral-storage", "example.com/foo"] and got.snapshot.n_res == 6
    assert list(got.snapshot.node_allocatable[4:, 6]) == [10 * 2**30, 3000] and list(got.snapshot.pod_req[:, 0]) == [1000, 0, 0, 1, 2**30, 2000]
    kept = ingest(doc(nodes=nodes, params={"restrictSchedulingNodes": True})).snapshot
    assert kept.node_names == ["a", "b"]


def test_subgroup_tree_rules():
    """subgroup_info/factory.go:16-135: entries with children become SubGroupSets, leaves PodSets (minAvailable = max(minMember, 1)), parents are
    lower-cased, the root carries spec.topologyConstraint; job_info.go:231-251: a pod whose sub-group is unknown stays out of the job."""
    topo = {"metadata": {"name": "t"}, "spec": {"levels": [{"nodeLabel": "zone"}, {"nodeLabel": "rack"}]}}
    sgs = [{"name": "workers", "topologyConstraint": {"topology": "t", "requiredTopologyLevel": "rack"}}, {"name": "w-a", "parent": "Workers", "minMember": 2},
           {"name": "w-b", "parent": "workers", "minMember": 0, "topologyConstraint": {"topology": "t", "preferredTopologyLevel": "rack"}}, {"name": "leader", "minMember": 1}]
    pg = pod_group("job", subGroups=sgs, topologyConstraint={"topology": "t", "requiredTopologyLevel": "zone", "preferredTopologyLevel": "nope"})
    pods = [pod("l0", "job", labels={"kai.scheduler/subgroup-name": "leader"}), pod("a0", "job", labels={"kai.scheduler/subgroup-name": "w-a"}),
            pod("a1", "job", labels={"kai.scheduler/subgroup-name": "w-a"}), pod("b0", "job", labels={"kai.scheduler/subgroup-name": "w-b"}),
            pod("stray", "job"), pod("lost", "job", labels={"kai.scheduler/subgroup-name": "ghost"}), pod("solo", "other"), pod("free")]
    nodes = [node("n0", labels={"zone": "z0", "rack": "r0"}), node("n1", labels={"zone": "z0", "rack": "r1"}), node("n2", labels={"zone": "z1"}), node("n3")]
    got = ingest(doc(nodes=nodes, queues=[queue("q")], pods=pods, pod_groups=[pg, pod_group("other", min_member=3)], topologies=[topo]))
    s = got.snapshot
    assert s.job_names == ["job", "other"] and list(s.job_n_podsets) == [3, 1] and list(s.job_n_pods) == [4, 1]
    assert s.podset_names == ["w-a", "w-b", "leader", "default"] and list(s.podset_min_available) == [2, 1, 1, 3]
    assert list(s.podset_name_rank) == [1, 2, 0, 0]  # leader < w-a < w-b inside the job
    # groups: job root (zone required, unknown preferred level lies beyond the last level), "workers" (rack required), root of "other"
    assert list(s.group_job) == [0, 0, 1] and list(s.group_parent) == [-1, 0, -1] and list(s.job_root_group) == [0, 2]
    assert list(s.group_topology) == [0, 0, -1] and list(s.group_required_level) == [0, 1, -1] and list(s.group_preferred_level) == [10**6, -1, -1]
    assert list(s.podset_group) == [1, 1, 0, 2] and list(s.podset_topology) == [-1, 0, -1, -1] and list(s.podset_preferred_level) == [-1, 1, -1, -1]
    names = [n.split("/")[1] for n in s.pod_names]
    assert names[:5] == ["l0", "a0", "a1", "b0", "solo"] and sorted(names[5:]) == ["free", "lost", "stray"]
    assert list(s.pod_job) == [0, 0, 0, 0, 1, -1, -1, -1] and list(s.pod_podset[:5]) == [2, 0, 0, 1, 3]
    assert sum("left out of the job" in w for w in got.warnings) == 2
    # topology tables: a node joins only with every level label (topology/common.go:70-77); domain ids "z0", "z0.r0", "z0.r1"
    assert list(s.topo_level_off) == [0, 2] and s.node_domain.tolist() == [[0, 0, -1, -1], [1, 2, -1, -1]]
    assert list(s.domain_level) == [0, 1, 1] and list(s.domain_parent) == [-1, 0, 0] and list(s.domain_id_rank) == [0, 1, 2]


def test_topology_tree_of_the_reference_plugin_test():
    """plugins/topology/topology_plugin_test.go:29-206 (TestTopologyPlugin_initializeTopologyTree): three nodes under two levels — the same rack label VALUE under two
    blocks gives two rack domains ("test-block-1.test-rack-1" and "test-block-2.test-rack-1": a domain is a prefix of label values, topology_structs.go:94-101); block 1
    has two children, block 2 one, racks none.  The domain table is the host's job (include/kai_core.h): checked for kai_ingest.cpp and for the scene builder of the tests."""
    labels = [("test-block-1", "test-rack-1"), ("test-block-1", "test-rack-2"), ("test-block-2", "test-rack-1")]  # :33-90
    lv = ("test-topology-label/block", "test-topology-label/rack")                                                 # :92-107
    topo = {"metadata": {"name": "test-topology"}, "spec": {"levels": [{"nodeLabel": k} for k in lv]}}
    s = ingest(doc(nodes=[node(f"test-node-{i}", labels=dict(zip(lv, l))) for i, l in enumerate(labels)], queues=[queue("q")], pods=[pod("p")], topologies=[topo])).snapshot
    scene = {"Name": "tree", "Nodes": {f"test-node-{i}": {"CPUMillis": 1000, "GPUs": 1, "MaxTaskNum": 100, "Labels": dict(zip(lv, l))} for i, l in enumerate(labels)},
             "Topologies": [{"ObjectMeta": {"Name": "test-topology"}, "Spec": {"Levels": [{"NodeLabel": k} for k in lv]}}], "Queues": [{"Name": "q", "DeservedGPUs": 1}],
             "Jobs": [{"Name": "j", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 0, "Tasks": [{"State": "Pending"}]}], "JobExpectedResults": {}}
    s2, _, _ = T.case_to_snapshot(scene)
    for snap in (s, s2):
        assert list(snap.topo_level_off) == [0, 2] and len(snap.domain_level) == 5                    # :188-191: root + two levels; 2 blocks + 3 racks
        level, parent = list(snap.domain_level), list(snap.domain_parent)
        blocks = [d for d in range(5) if level[d] == 0]; racks = [d for d in range(5) if level[d] == 1]
        assert len(blocks) == 2 and len(racks) == 3 and all(parent[d] == -1 for d in blocks)
        nd = np.asarray(snap.node_domain).reshape(2, 3); order = [list(snap.node_names).index(f"test-node-{i}") for i in range(3)]
        b = [int(nd[0, n]) for n in order]; r = [int(nd[1, n]) for n in order]
        assert b[0] == b[1] != b[2] and len(set(r)) == 3                                         # the two "test-rack-1" are different domains
        children = lambda d: sum(1 for x in racks if parent[x] == d)
        assert children(b[0]) == 2 and children(b[2]) == 1                                       # :196-203
        assert [parent[x] for x in r] == [b[0], b[1], b[2]] and all(children(x) == 0 for x in racks)


def test_static_predicate_classes():
    """n4: NodeAffinity (nodeSelector + required terms: In, NotIn, Exists, DoesNotExist, Gt, Lt, matchFields) and TaintToleration
    (NoSchedule / NoExecute only; Equal / Exists, empty key, effect match) — k8s.io/kubernetes v1.34.2 semantics."""
    taint = lambda k, v, e: {"key": k, "value": v, "effect": e}
    nodes = [node("plain"), node("a100", labels={"gpu": "a100", "mem": "80"}), node("h100", labels={"gpu": "h100", "mem": "94"}),
             node("tainted", labels={"gpu": "a100", "mem": "40"}, taints=[taint("dedicated", "ml", "NoSchedule")]),
             node("soft", taints=[taint("x", "y", "PreferNoSchedule")]), node("evict", taints=[taint("down", "", "NoExecute")])]
    term = lambda *ex: {"affinity": {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": [{"matchExpressions": list(ex)}]}}}}
    ex = lambda k, op, *v: dict({"key": k, "operator": op}, **({"values": list(v)} if v else {}))
    cases = {
        "any": ({}, {"plain", "a100", "h100", "soft"}),
        "selector": ({"nodeSelector": {"gpu": "a100"}}, {"a100"}),
        "in": (term(ex("gpu", "In", "a100", "h100")), {"a100", "h100"}),
        "notin": (term(ex("gpu", "NotIn", "a100")), {"plain", "h100", "soft"}),
        "exists": (term(ex("gpu", "Exists")), {"a100", "h100"}),
        "absent": (term(ex("gpu", "DoesNotExist")), {"plain", "soft"}),
        "gt": (term(ex("mem", "Gt", "79")), {"a100", "h100"}),
        "lt": (term(ex("mem", "Lt", "90")), {"a100"}),
        "and": (term(ex("gpu", "Exists"), ex("mem", "Gt", "90")), {"h100"}),
        "or": ({"affinity": {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": [
            {"matchExpressions": [ex("gpu", "In", "h100")]}, {"matchFields": [ex("metadata.name", "In", "plain")]}]}}}}, {"h100", "plain"}),
        "emptyterms": ({"affinity": {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": []}}}}, set()),
        "emptyterm": ({"affinity": {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": [{}]}}}}, set()),
        "tol-equal": ({"tolerations": [{"key": "dedicated", "operator": "Equal", "value": "ml", "effect": "NoSchedule"}]}, {"plain", "a100", "h100", "soft", "tainted"}),
        "tol-wrongvalue": ({"tolerations": [{"key": "dedicated", "value": "web"}]}, {"plain", "a100", "h100", "soft"}),
        "tol-exists": ({"tolerations": [{"key": "dedicated", "operator": "Exists"}]}, {"plain", "a100", "h100", "soft", "tainted"}),
        "tol-all": ({"tolerations": [{"operator": "Exists"}]}, {"plain", "a100", "h100", "soft", "tainted", "evict"}),
        "tol-effect": ({"tolerations": [{"operator": "Exists", "effect": "NoExecute"}]}, {"plain", "a100", "h100", "soft", "evict"}),
        "both": ({"nodeSelector": {"gpu": "a100"}, "tolerations": [{"key": "dedicated", "operator": "Exists"}]}, {"a100", "tainted"}),
    }
    pods = [pod(n, spec=spec) for n, (spec, _) in cases.items()]
    s = ingest(doc(nodes=nodes, pods=pods)).snapshot
    for i, (n, (_, want)) in enumerate(cases.items()):
        fit = {s.node_names[k] for k in range(s.n_nodes) if s.class_fit[s.pod_class[i], s.node_class[k]]}
        assert fit == want, (n, fit, want)
    assert s.n_node_classes <= 6 and s.n_pod_classes <= len(cases)
    # nodes that no constraint can tell apart share a class
    s2 = ingest(doc(nodes=nodes, pods=[pod("p")])).snapshot
    assert s2.n_pod_classes == 1 and s2.n_node_classes == 3  # untainted / dedicated:NoSchedule / down:NoExecute


# The reference's own (pod constraint, node) cases for the NodeAffinity Filter of k8s.io/kubernetes v1.34.2 (absent from /root/reference): the table of
# accumulated_scenario_filters/node_affinities/node_affinities_test.go:248-424 (TestNodeAffinitiesFilter_Filter, ten rows) and the three tests before it in which no
# filter is created (:220-246) — tests/golden/kat_node_affinities.json, generated from the Go source by tools/go_kat_node_affinities.py.
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_node_affinities.json")) as _fh:
    NODE_AFFINITIES = json.load(_fh)
_TERMS = lambda *terms: {"affinity": {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": list(terms)}}}}


def _pod_spec(p):
    """a fixture pod (kind + arguments of the test file's helper, node_affinities_test.go:79-218) as a pod spec"""
    if p["kind"] == "selector": return {"nodeSelector": p["selector"]}
    if p["kind"] == "required_in": return _TERMS({"matchExpressions": [{"key": p["key"], "operator": "In", "values": p["values"]}]})
    if p["kind"] == "required_match_fields": return _TERMS({"matchFields": [{"key": "metadata.name", "operator": "In", "values": p["node_names"]}]})
    if p["kind"] == "preferred_only":
        return {"affinity": {"nodeAffinity": {"preferredDuringSchedulingIgnoredDuringExecution": [{"weight": p["weight"], "preference": {"matchExpressions": [{"key": p["key"], "operator": "In", "values": p["values"]}]}}]}}}
    assert p["kind"] == "none"
    return {}


def _node_affinity_answers(all_nodes, pending):
    """What the upstream NodeAffinity plugin answers for each pending pod, from the COMPILED class table of kai_ingest.cpp: required (hasRequiredNodeAffinity,
    node_affinities.go:134-145), match per node (Filter), PreFilter's node names (terms of matchFields metadata.name In [...] only; names the cluster does not
    hold left out, :157-160)."""
    specs = [_pod_spec(p) for p in pending]
    s = ingest(doc(nodes=[node(n, labels=l) for n, l in all_nodes.items()] or [node("placeholder")], pods=[pod("p%d" % i, spec=sp) for i, sp in enumerate(specs)])).snapshot
    names = list(s.node_names) if all_nodes else []; idx = {n: k for k, n in enumerate(names)}
    required, match, pre = [], [], []
    for i, sp in enumerate(specs):
        req = (sp.get("affinity", {}).get("nodeAffinity", {}) or {}).get("requiredDuringSchedulingIgnoredDuringExecution")
        required.append("nodeSelector" in sp or req is not None)
        terms = (req or {}).get("nodeSelectorTerms", [])
        field_only = bool(terms) and all(t.get("matchFields") and not t.get("matchExpressions") and all(f["key"] == "metadata.name" and f["operator"] == "In" for f in t["matchFields"]) for t in terms)
        pre.append(sorted(idx[v] for t in terms for f in t["matchFields"] for v in f["values"] if v in idx) if field_only else None)
        pi = [k for k, n in enumerate(s.pod_names) if n.endswith("/p%d" % i)][0]
        match.append([bool(s.class_fit[s.pod_class[pi], s.node_class[idx[n]]]) for n in names])
    return names, idx, required, match, pre


def _oracle_node_affinities(n_nodes, feasible, required, match, pre, victim_nodes, with_scenario=True):
    """one call of kai_oracle_node_affinities_kat (oracle_solver.hpp AccumulatedNodeAffinities): -1 no filter, else Filter"""
    lib = T.Oracle.lib(); lib.kai_oracle_node_affinities_kat.restype = C.c_int
    i32 = lambda v: (C.c_int32 * max(len(v), 1))(*v); u8 = lambda v: (C.c_uint8 * max(len(v), 1))(*[int(x) for x in v])
    off = [0]; flat = []
    for p in pre: flat += p or []; off.append(len(flat))
    return lib.kai_oracle_node_affinities_kat(n_nodes, i32(feasible), len(feasible), len(required), u8(required), u8([x for row in match for x in row]),
                                              i32([0 if p is None else 1 for p in pre]), i32(off), i32(flat), i32(victim_nodes), len(victim_nodes), int(with_scenario))


@pytest.mark.parametrize("case", NODE_AFFINITIES["filter_cases"], ids=[c["name"][:60] for c in NODE_AFFINITIES["filter_cases"]])
def test_static_predicate_matcher_on_reference_node_affinity_cases(case):
    """n4 and a29 pinned on the reference: the compiled class_fit table (NodeAffinity Filter semantics restated in kai_ingest.cpp) must give the reference's
    expectation in each of its ten AccumulatedNodeAffinities cases — through the filter as the reference wires it (node_affinities.go:83-188), once restated here
    over the table (victims' nodes join the feasible set; a pod with a nodeSelector or required terms needs ONE node that passes the Filter among the PreFilter's
    node names looked up in the WHOLE cluster, else among the feasible nodes), once as the oracle's AccumulatedNodeAffinities (oracle_solver.hpp) fed with the
    table's answers."""
    names, idx, required, match, pre = _node_affinity_answers(case["all_nodes"], case["pending"])
    feas = set(case["feasible"]) | {n for n in case["victims"] if n in idx}
    ok = True
    for i in range(len(case["pending"])):
        if not required[i]: continue
        cand = pre[i] if pre[i] is not None else [idx[n] for n in feas if n in idx]
        if not any(match[i][k] for k in cand): ok = False
    assert ok == case["want"], case["name"]
    got = _oracle_node_affinities(len(names), [idx[n] for n in case["feasible"]], required, match, pre, [idx.get(n, -1) for n in case["victims"]])
    assert got == int(case["want"]), (case["name"], got)


@pytest.mark.parametrize("case", NODE_AFFINITIES["no_filter_cases"], ids=[c["name"] for c in NODE_AFFINITIES["no_filter_cases"]])
def test_node_affinities_filter_is_not_created_without_a_required_affinity(case):
    """node_affinities_test.go:220-246: no scenario, no pending pod with a node affinity, a preferred-only affinity → NewNodeAffinitiesFilter returns nil"""
    names, idx, required, match, pre = _node_affinity_answers({}, case["pending"])
    assert not any(required)
    assert _oracle_node_affinities(0, [], required, match, pre, [], with_scenario=case["scenario"]) == -1


def test_fallback_flags_and_config_maps():
    """SURVEY §8b fallback rule + k8s_internal/predicates/config_maps.go (a missing non-optional config map fits no node)."""
    cm_vol = {"volumes": [{"name": "v", "configMap": {"name": "cm1"}}], "containers": [{"name": "c", "volumeMounts": [{"name": "v"}], "resources": {}}]}
    pods = [pod("frac", annotations={"gpu-fraction": "0.5"}), pod("mem", annotations={"gpu-memory": "2000"}), pod("mig", requests={"nvidia.com/mig-1g.5gb": "1"}),
            pod("multi", annotations={"gpu-fraction": "0.5", "gpu-fraction-num-devices": "2"}), pod("both", annotations={"gpu-fraction": "0.5", "gpu-memory": "2000"}),
            pod("port", spec={"containers": [{"name": "c", "ports": [{"hostPort": 80}]}]}), pod("pvc", spec={"volumes": [{"name": "d", "persistentVolumeClaim": {"claimName": "x"}}]}),
            pod("dra", spec={"resourceClaims": [{"name": "c"}]}), pod("aff", spec={"affinity": {"podAffinity": {}}}), pod("ok"),
            pod("cm-ok", spec=cm_vol), pod("cm-missing", spec=dict(cm_vol, volumes=[{"name": "v", "configMap": {"name": "nope"}}])),
            pod("cm-optional", spec=dict(cm_vol, volumes=[{"name": "v", "configMap": {"name": "nope", "optional": True}}])),
            pod("cm-unmounted", spec={"volumes": [{"name": "v", "configMap": {"name": "nope"}}]}),
            pod("cm-env", spec={"containers": [{"name": "c", "env": [{"name": "E", "valueFrom": {"configMapKeyRef": {"name": "nope", "key": "k"}}}]}]}),
            pod("foreign", spec={"schedulerName": "default-scheduler"}), pod("prio", labels={"kai.scheduler/task-priority": "7"})]
    s = ingest(doc(nodes=[node("n")], pods=pods, configMaps=[{"metadata": {"name": "cm1", "namespace": "ns"}}])).snapshot
    names = [n.split("/")[1] for n in s.pod_names]
    fl = {n: int(s.pod_flags[i]) for i, n in enumerate(names)}
    for n in ("multi", "both", "port", "pvc", "dra", "aff"):
        assert fl[n] & abi.POD_CPU_FALLBACK, n
    for n in ("ok", "cm-ok", "cm-missing", "foreign", "prio", "frac", "mem", "mig"):  # a fraction / MiB of ONE device and MIG instances are described to the device (ABI v4 / v5)
        assert not fl[n] & abi.POD_CPU_FALLBACK, n
    # the MIG profile is a resource row of its own, counted in instances, with its GPU weight and memory beside it (mig.go:13-33)
    rows = [k for k in range(4, s.n_res) if s.res_mig_gpus[k] > 0]
    assert len(rows) == 1 and (int(s.res_mig_gpus[rows[0]]), int(s.res_mig_memory[rows[0]])) == (1, 5) and s.pod_req[rows[0], names.index("mig")] == 1 and s.pod_req[abi.RES_GPU, names.index("mig")] == 0
    i_mem, i_frac = names.index("mem"), names.index("frac")
    assert s.pod_gpu_memory[i_mem] == 2000 and s.pod_req[abi.RES_GPU, i_mem] == 0 and s.pod_gpu_portion[i_mem] == 0  # NewGpuResourceRequirementWithGpus(0, memory): GPUs() == 0
    assert s.pod_gpu_portion[i_frac] == 0.5 and s.pod_gpu_memory[i_frac] == 0
    assert fl["foreign"] & abi.POD_FOREIGN_SCHEDULER and not fl["ok"] & abi.POD_FOREIGN_SCHEDULER
    assert fl["prio"] & abi.POD_HAS_TASK_PRIORITY and s.pod_task_priority[names.index("prio")] == 7
    fits = {n: bool(s.class_fit[s.pod_class[i], s.node_class[0]]) for i, n in enumerate(names)}
    assert fits["cm-ok"] and fits["cm-optional"] and fits["cm-unmounted"] and fits["ok"] and not fits["cm-missing"] and not fits["cm-env"]


def test_active_gpu_state_and_utility_pods():
    """(1) a RUNNING gpu-memory / MIG / DRA pod holds GPU state the device's node accounting does not carry — flagged
    KAI_POD_GPU_UNMODELLED so that kai_session_open can refuse it (api/node_info/node_info.go:457-493 takes a device out of Idle for it);
    (2) kai utility pods (api/pod_info/utility_pods.go:13-33): a reservation pod's own GPU is not booked on its node (node_info.go:465) and neither
    it nor a scale-adjust pod counts as another scheduler's pod (plugins/proportion/proportion.go:276-285)."""
    pods = [pod("mem", annotations={"gpu-memory": "2000"}, phase="Running", node_name="n", labels={"runai-gpu-group": "1"}), pod("mig", requests={"nvidia.com/mig-1g.5gb": "1"}, phase="Running", node_name="n"),
            pod("multi", annotations={"gpu-fraction": "0.5", "gpu-fraction-num-devices": "2"}, phase="Running", node_name="n"),
            pod("frac", annotations={"gpu-fraction": "0.5"}, phase="Running", node_name="n", labels={"runai-gpu-group": "0"}),
            pod("plain", phase="Running", node_name="n"),
            pod("reservation", phase="Running", node_name="n", labels={"app": "kai-resource-reservation"}, spec={"schedulerName": "default-scheduler"}),
            pod("scaler", phase="Running", node_name="n", labels={"app": "scaling-pod"}, spec={"schedulerName": "default-scheduler"}),
            pod("foreign", phase="Running", node_name="n", spec={"schedulerName": "default-scheduler"})]
    s = ingest(doc(nodes=[node("n")], pods=pods)).snapshot
    names = [n.split("/")[1] for n in s.pod_names]
    fl = {n: int(s.pod_flags[i]) for i, n in enumerate(names)}
    assert fl["multi"] & abi.POD_GPU_UNMODELLED and not fl["mig"] & abi.POD_GPU_UNMODELLED  # MIG instances are resource rows (ABI v5 res_mig_*)
    assert not fl["frac"] & abi.POD_GPU_UNMODELLED and not fl["mem"] & abi.POD_GPU_UNMODELLED and not fl["plain"] & abi.POD_GPU_UNMODELLED  # a fraction / MiB of one device is described to the ABI (v4 / v5)
    assert s.pod_gpu_group[names.index("mem")] == 1 and s.pod_gpu_memory[names.index("mem")] == 2000
    assert fl["foreign"] & abi.POD_FOREIGN_SCHEDULER
    assert not fl["reservation"] & abi.POD_FOREIGN_SCHEDULER and not fl["scaler"] & abi.POD_FOREIGN_SCHEDULER
    gpu = {n: float(s.pod_req[abi.RES_GPU, i]) for i, n in enumerate(names)}
    assert gpu["reservation"] == 0.0 and gpu["scaler"] == 1.0 and gpu["plain"] == 1.0


def test_config_and_actions():
    """conf/scheduler_conf.go:31-88, conf_util/scheduler_conf_util.go:36-107, plugin arguments (nodeplacement.go:59-70, proportion.go:67-93,
    minruntime.go:40-70)."""
    got = ingest(doc(config={}))
    assert got.actions == ["allocate", "consolidation", "reclaim", "preempt"] and got.config.plugins == abi.PLUGIN_ALL  # defaults; stalegangeviction is skipped
    assert got.config.k_value == 1.0 and got.config.gpu_strategy == abi.BINPACK and list(got.config.queue_depth) == [-1] * 4 and got.config.min_node_gpu_memory == 100
    tiers = [{"plugins": [{"name": "predicates"}, {"name": "proportion", "arguments": {"kValue": "0.5", "relcaimerSaturationMultiplier": "1.5"}}, {"name": "gpupack"}, {"name": "podaffinity"},
                          {"name": "nodeplacement", "arguments": {"gpu": "spread", "cpu": "binpack"}},
                          {"name": "minruntime", "arguments": {"defaultPreemptMinRuntime": "5m", "defaultReclaimMinRuntime": "1h30m", "reclaimResolveMethod": "queue"}}]}]
    got = ingest(doc(config={"actions": "reclaim, allocate", "tiers": tiers, "queueDepthPerAction": {"reclaim": 10, "preempt": 3}},
                     params={"maxNumberConsolidationPreemptees": 16, "allowConsolidatingReclaim": True, "restrictSchedulingNodes": True}), now_ns=123)
    c = got.config
    assert got.actions == ["reclaim", "allocate"]
    assert c.plugins == abi.PLUGINS["predicates"] | abi.PLUGINS["proportion"] | abi.PLUGINS["nodeplacement"] | abi.PLUGINS["minruntime"] | abi.PLUGINS["gpupack"]
    assert (c.k_value, c.reclaimer_saturation_multiplier, c.gpu_strategy, c.cpu_strategy) == (0.5, 1.5, abi.SPREAD, abi.BINPACK)
    assert (c.default_preempt_min_runtime_ns, c.default_reclaim_min_runtime_ns, c.reclaim_resolve_method) == (300 * 10**9, 5400 * 10**9, 1)
    assert list(c.queue_depth) == [-1, -1, 10, 3] and c.max_consolidation_preemptees == 16 and c.allow_consolidating_reclaim == 1 and c.restrict_node_scheduling == 1
    assert c.now_ns == 123 and any("podaffinity" in w for w in got.warnings)
    with pytest.raises(ing.IngestError, match="failed to find Action bogus"):
        ingest(doc(config={"actions": "allocate, bogus"}))


def test_queue_fields_and_minruntime_inputs():
    q = queue("q", gpu_quota=8, priority=200, preemptMinRuntime="10m", reclaimMinRuntime="1m30s")
    q["spec"]["resources"]["cpu"] = {"quota": 16000, "limit": 32000, "overQuotaWeight": 2}
    pg = pod_group("j"); pg["metadata"]["annotations"] = {"kai.scheduler/last-start-timestamp": "2024-03-01T12:00:00+02:00"}
    got = ingest(doc(queues=[q, queue("plain")], pod_groups=[pg, pod_group("k")]))
    s = got.snapshot
    assert list(s.queue_priority) == [200, 100] and list(s.queue_deserved[:, 0]) == [16000, -1, 8] and list(s.queue_limit[:, 0]) == [32000, -1, -1] and list(s.queue_oqw[:, 0]) == [2, 1, 1]
    assert list(s.queue_preempt_min_runtime_ns) == [600 * 10**9, -1] and list(s.queue_reclaim_min_runtime_ns) == [90 * 10**9, -1]
    assert list(s.job_last_start_ns) == [1709287200 * 10**9, 0]  # 2024-03-01T10:00:00Z
    assert got.config.now_ns == 1709287200 * 10**9  # default "now": the latest timestamp in the snapshot
    assert s.queue_created_ns[0] == 1704067200 * 10**9


def test_scheduling_signatures():
    """job_info.go:547-570 / podset.go:167-196 / scheduling_constraints_signature.go: equal constraint sets ⇔ equal ids; resource requests,
    names and placed pods do not enter."""
    sel = {"nodeSelector": {"gpu": "a100", "zone": "z"}}
    sel_reordered = {"nodeSelector": {"zone": "z", "gpu": "a100"}}
    pods = [pod("a0", "a", spec=sel), pod("a1", "a", spec=sel), pod("b0", "b", spec=sel_reordered, requests={"cpu": "9"}), pod("b1", "b", spec=sel),
            pod("c0", "c", spec=sel), pod("d0", "d"), pod("e0", "e"), pod("e1", "e", phase="Running", node_name="n", spec=sel),
            pod("f0", "f", spec=dict(sel, tolerations=[{"operator": "Exists"}]))]
    s = ingest(doc(nodes=[node("n", labels={"gpu": "a100", "zone": "z"})], queues=[queue("q")], pods=pods, pod_groups=[pod_group(x) for x in "abcdef"])).snapshot
    sig = dict(zip(s.job_names, (int(x) for x in s.job_signature)))
    assert sig["a"] == sig["b"] and sig["d"] == sig["e"] and len({sig["a"], sig["c"], sig["d"], sig["f"]}) == 4


def test_pods_are_converted_from_spans_one_at_a_time():
    """rawObjects.pods is only LOCATED by the document parser; every element is parsed on its own when it is converted (kai_ingest.cpp PodSpans): a syntax error inside one pod
    is reported with its byte offset in the document, an element that is not an object is skipped like encoding/json would leave it zero-valued, and the result equals what
    the same document gives when its pods are few."""
    d = doc(nodes=[node("n")], queues=[queue("q")], pods=[pod("a0", "a"), pod("a1", "a"), pod("b0", "b")], pod_groups=[pod_group("a"), pod_group("b")])
    text = json.dumps(d)
    good = ing.ingest_json(text).snapshot
    assert good.n_pods == 3
    # a broken element: the offset points into that element
    i = text.index('"name": "a1"')
    bad = text[:i] + '"name": @' + text[i + len('"name": "a1"'):]
    with pytest.raises(ing.IngestError, match=r"snapshot.json: .* at byte (\d+)") as e:
        ing.ingest_json(bad)
    at = int(str(e.value).rsplit("at byte ", 1)[1].split()[0].rstrip(")"))
    assert abs(at - i) < 40
    # elements that are not objects are passed over
    j = text.index('"pods": [') + len('"pods": [')
    odd = text[:j] + "null, 7, " + text[j:]
    assert ing.ingest_json(odd).snapshot.n_pods == 3
    assert list(ing.ingest_json(odd).snapshot.pod_names) == list(good.pod_names)


def test_malformed_input_is_rejected():
    for text in ("", "{", "[1,2", '{"a":}', '{"a":1}x', "nul"):
        with pytest.raises(ing.IngestError):
            ing.ingest_json(text)
    with pytest.raises(ing.IngestError, match="bad quantity"):
        ingest(doc(nodes=[node("n", cpu="lots")]))
    with pytest.raises(ing.IngestError, match="cannot open"):
        ing.ingest_file("/nonexistent/snapshot.zip")
    assert ingest({"rawObjects": {}}).snapshot.n_nodes == 0  # an empty cluster is a valid snapshot


def test_json_escapes_and_unicode_names():
    raw = json.dumps(doc(nodes=[node("n")])).replace('"n"', '"n\\u00e9\\ud83d\\ude00\\t\\"q\\""', 1)
    s = ing.ingest_json(raw).snapshot
    assert s.node_names == ['né\U0001F600\t"q"']


# ------------------------------------------------------------------------------------------------ round trips
def _assert_same_arrays(snap, got):
    a, b = snap.arrays, got.arrays
    for k, x in a.items():
        if k in ("job_signature", "class_fit", "pod_class", "node_class", "queue_usage", "domain_id_rank", "node_domain", "domain_level", "domain_parent", "group_name_rank"):
            continue
        y = b[k]
        if k in ("pod_created_ns", "job_created_ns", "queue_created_ns"): y = y - E
        if k == "job_last_start_ns": y = np.where(y != 0, y - E, 0)
        if k == "pod_task_priority": x = np.where(a["pod_flags"] & abi.POD_HAS_TASK_PRIORITY, x, 0)  # only read under the flag
        assert x.shape == y.shape and np.array_equal(x, y), k
    if snap.n_pods and snap.n_nodes:  # the class tables may be factored differently; the pod × node fit relation must be the same
        assert np.array_equal(a["class_fit"][a["pod_class"]][:, a["node_class"]], b["class_fit"][b["pod_class"]][:, b["node_class"]])
    if "group_name_rank" in a:  # only siblings are ever compared (framework/session_plugins.go:273-282): same order under every parent
        sib = lambda arr, g: sum(1 for h in range(len(arr)) if a["group_job"][h] == a["group_job"][g] and a["group_parent"][h] == a["group_parent"][g] and arr[h] < arr[g])
        for g in range(len(a["group_job"])):
            assert sib(a["group_name_rank"], g) == sib(b["group_name_rank"], g)
    if "domain_id_rank" in a:  # domain numbering is free: the tables must be the same up to the renumbering the node rows induce
        m = {-1: -1}
        for x, y in zip(a["node_domain"].ravel().tolist(), b["node_domain"].ravel().tolist()):
            assert m.setdefault(x, y) == y
        assert len(set(m.values())) == len(m) and len(m) - 1 == len(b["domain_level"]) == len(a["domain_level"])
        for d in range(len(a["domain_level"])):
            assert a["domain_level"][d] == b["domain_level"][m[d]] and m[int(a["domain_parent"][d])] == b["domain_parent"][m[d]]
        for lv in np.unique(a["domain_level"]):  # ranks are order-isomorphic inside a level
            ds = [d for d in range(len(a["domain_level"])) if a["domain_level"][d] == lv]
            assert sorted(ds, key=lambda d: a["domain_id_rank"][d]) == sorted(ds, key=lambda d: b["domain_id_rank"][m[d]])
    # job_signature is not compared: a synthetic snapshot carries arbitrary ids, the ingest derives them from the pods' constraints
    # (test_scheduling_signatures); the round-trip tests hand the derived ids to both runs.


def _roundtrip(snap, cfg, actions):
    d = ing.export_snapshot_json(snap, cfg, actions)
    got = ing.ingest_json(json.dumps(d))
    _assert_same_arrays(snap, got.snapshot)
    assert got.actions == list(actions) and not got.warnings
    for f, _ in abi.KaiConfig._fields_:
        if f in ("now_ns", "engine_mode", "reserved", "queue_depth", "full_hierarchy_fairness", "pad0", "min_node_gpu_memory"):
            continue
        assert getattr(got.config, f) == getattr(cfg, f), f
    assert list(got.config.queue_depth) == list(cfg.queue_depth)
    return got


@pytest.mark.parametrize("idx,scale", [(0, 1.0), (1, 0.05), (2, 0.01), (3, 0.01)])
def test_roundtrip_baseline_configs(idx, scale):
    snap, cfg, _ = pkg.synth.config(idx, scale=scale)
    acts = ("allocate",) if idx < 3 else ("allocate", "consolidation", "reclaim")
    got = _roundtrip(snap, cfg, acts)
    snap.arrays["job_signature"] = got.snapshot.arrays["job_signature"]
    ref, res = T.Oracle.run(snap, cfg, acts), T.Oracle.run(got.snapshot, got.config, tuple(got.actions))
    assert ref.ops == res.ops and len(ref.ops) > 0
    assert np.array_equal(ref.pod_status, res.pod_status) and np.array_equal(ref.pod_node, res.pod_node)
    assert np.array_equal(ref.shares_final["fair_share"], res.shares_final["fair_share"])


@pytest.mark.parametrize("seed", range(1000, 1010))
def test_roundtrip_randomized_cycles(seed):
    """Seeds of the broad campaign (full cycles incl. the victim actions: signatures, minruntime inputs, elastic gangs, two pod-sets, replica / rack
    topology, lexicographic node names) through the file format: same arrays, same scheduling result."""
    seen = set()
    for snap, cfg, acts in T.broad_case(seed):
        if id(snap) not in seen:  # one snapshot serves several action orders: shift it to absolute time once
            seen.add(id(snap))
            a = snap.arrays
            if "queue_usage" in a: a["queue_usage"][:] = 0  # usage history is not part of the snapshot schema
            got = ing.ingest_json(json.dumps(ing.export_snapshot_json(snap, cfg, acts)), now_ns=E + int(cfg.now_ns))
            _assert_same_arrays(snap, got.snapshot)
            if "job_last_start_ns" in a: a["job_last_start_ns"] = np.where(a["job_last_start_ns"] != 0, a["job_last_start_ns"] + E, 0)
            a["job_signature"] = got.snapshot.arrays["job_signature"]
            ingested = got.snapshot
        cfg2 = ing.ingest_json(json.dumps(ing.export_snapshot_json(ingested, cfg, acts)), now_ns=E + int(cfg.now_ns)).config
        ref_cfg = abi.KaiConfig.from_buffer_copy(cfg); ref_cfg.now_ns = E + int(cfg.now_ns)
        ref, res = T.Oracle.run(snap, ref_cfg, acts), T.Oracle.run(ingested, cfg2, acts)
        assert ref.ops == res.ops
        assert np.array_equal(ref.pod_status, res.pod_status) and np.array_equal(ref.pod_node, res.pod_node)


def test_snapshot_zip(tmp_path):
    """BASELINE config 1 as an actual snapshot.zip (plugins/snapshot/snapshot.go:33, cmd/snapshot-tool/main.go:118-147): deflated and stored members."""
    import zipfile
    snap, cfg, _ = pkg.synth.config(0)
    d = ing.export_snapshot_json(snap, cfg, ("allocate",))
    z = tmp_path / "snapshot.zip"
    ing.write_snapshot_zip(str(z), d)
    got = ing.ingest_file(str(z))
    _assert_same_arrays(snap, got.snapshot)
    with zipfile.ZipFile(tmp_path / "stored.zip", "w", zipfile.ZIP_STORED) as f:
        f.writestr("readme.txt", "x"); f.writestr("snapshot.json", json.dumps(d))
    _assert_same_arrays(snap, ing.ingest_file(str(tmp_path / "stored.zip")).snapshot)
    (tmp_path / "plain.json").write_text(json.dumps(d))
    _assert_same_arrays(snap, ing.ingest_file(str(tmp_path / "plain.json")).snapshot)
    with zipfile.ZipFile(tmp_path / "empty.zip", "w") as f:
        f.writestr("other.json", "{}")
    with pytest.raises(ing.IngestError, match="snapshot.json"):
        ing.ingest_file(str(tmp_path / "empty.zip"))
    ref, res = T.Oracle.run(snap, cfg, ("allocate",)), T.Oracle.run(got.snapshot, got.config, tuple(got.actions))
    assert ref.ops == res.ops and len(res.ops) == 64


def _rich_document():
    """A small cluster using what only the file format carries: taints / selectors, pods of no job, extended resources, sub-groups."""
    nodes = [node(f"gpu-{i}", labels={"pool": "gpu", "zone": f"z{i // 2}"}, alloc={"example.com/nic": "2"}) for i in range(4)]
    nodes += [node(f"cpu-{i}", gpu=None, cpu="32", labels={"pool": "cpu"}, taints=[{"key": "cpu-only", "value": "", "effect": "NoSchedule"}]) for i in range(2)]
    qs = [queue("dep"), queue("team-a", "dep", gpu_quota=8), queue("team-b", "dep", gpu_quota=8)]
    pods, pgs = [pod("daemon", requests={"cpu": "500m"}, phase="Running", node_name="gpu-0", spec={"schedulerName": "default-scheduler"})], []
    for j in range(10):
        team = "team-a" if j % 2 else "team-b"
        pgs.append(pod_group(f"job-{j}", team, min_member=2 if j % 3 == 0 else 1, priorityClassName="train"))
        pgs[-1]["metadata"]["creationTimestamp"] = f"2024-01-01T00:{j:02d}:00Z"
        for k in range(2 if j % 3 == 0 else 1):
            spec = {}
            if j % 4 == 0: spec = {"nodeSelector": {"zone": "z1"}}
            if j % 5 == 1: spec = {"tolerations": [{"key": "cpu-only", "operator": "Exists"}], "nodeSelector": {"pool": "cpu"}}
            req = {"cpu": "2", "memory": "4Gi"} if j % 5 == 1 else {"cpu": "4", "memory": "8Gi", "nvidia.com/gpu": str(1 + j % 3)}
            if j == 7: req["example.com/nic"] = "2"
            pods.append(pod(f"job-{j}-{k}", f"job-{j}", requests=req, spec=spec, metadata={"creationTimestamp": f"2024-01-01T00:{j:02d}:0{k}Z"}))
    pods.append(pod("run-0", "job-running", phase="Running", node_name="gpu-1", requests={"cpu": "4", "memory": "8Gi", "nvidia.com/gpu": "6"}))
    pgs.append(pod_group("job-running", "team-a", priorityClassName="train"))
    return doc(nodes=nodes, queues=qs, pods=pods, pod_groups=pgs, priorityClasses=[{"metadata": {"name": "train"}, "value": 50}],
               config={"actions": "allocate, reclaim"}, params={"maxNumberConsolidationPreemptees": 16, "allowConsolidatingReclaim": True})


def test_rich_document_schedules_identically_in_oracle_and_engine():
    """Ingest → the engine's control flow (host-compiled) and the oracle agree on a snapshot with predicate classes, pods of no job and an extra
    resource column; placements honour selectors and taints."""
    import test_engine_hostsim as H
    got = ingest(_rich_document())
    s = got.snapshot
    assert got.resource_names[4:] == ["example.com/nic"] and s.n_pod_classes == 3 and s.n_node_classes == 3 and not got.warnings
    ref = T.Oracle.run(s, got.config, tuple(got.actions))
    sim = H.HostSim.run(s, got.config, tuple(got.actions))
    assert sim.ops == ref.ops and np.array_equal(sim.pod_status, ref.pod_status) and np.array_equal(sim.pod_node, ref.pod_node)
    placed = {s.pod_names[p].split("/")[1]: s.node_names[n] for kind, p, n, _ in ref.ops if kind in (0, 1)}
    assert placed and all(v.startswith("cpu-") for k, v in placed.items() if k.startswith(("job-1-", "job-6-")))
    assert all(v in ("gpu-2", "gpu-3") for k, v in placed.items() if k.startswith(("job-0-", "job-4-", "job-8-")))
    assert all(not v.startswith("cpu-") for k, v in placed.items() if not k.startswith(("job-1-", "job-6-")))


@pytest.mark.gpu
def test_gpu_ingested_snapshots_match_oracle(tmp_path):
    """snapshot.zip → ingest → MI355X engine through the C ABI → bit-identical to the oracle (config 1 from the file, and the rich document)."""
    snap, cfg, _ = pkg.synth.config(0)
    z = tmp_path / "snapshot.zip"
    ing.write_snapshot_zip(str(z), ing.export_snapshot_json(snap, cfg, ("allocate",)))
    for got in (ing.ingest_file(str(z)), ingest(_rich_document())):
        s, acts = got.snapshot, tuple(got.actions)
        ref = T.Oracle.run(s, got.config, acts)
        ops = []
        with pkg.KaiCore(got.config) as core:
            ssn = core.open_session(s)
            for a in acts:
                ops += [(int(o["kind"]), int(o["pod"]), int(o["node"]), int(o["job"])) for o in ssn.execute(a)]
            st, nd = ssn.pod_states()
            ssn.close()
        assert ops == ref.ops and len(ops) > 0
        assert np.array_equal(st, ref.pod_status) and np.array_equal(nd, ref.pod_node)


# ------------------------------------------------------------------------------------------------ decisions out (n3)
def test_decisions_document():
    """cache/cache.go:290-330 createBindRequest (name, namespace, owner reference, selected-node + node-pool labels, podName, selectedNode,
    receivedResourceType, receivedGPU{count, portion "%.2f"}) for Allocate; evictions carry the pod group (cache.go:216-252); one batch."""
    d = doc(nodes=[node("n0", labels={"pool": "a"}), node("n1", labels={"pool": "a"})], queues=[queue("q", labels={"pool": "a"})],
            pods=[pod("gpu", "j", requests={"cpu": "1", "nvidia.com/gpu": "2"}), pod('c"pu', "j", requests={"cpu": "1"}), pod("victim", "k", phase="Running", node_name="n1")],
            pod_groups=[dict(pod_group("j"), metadata={"name": "j", "namespace": "ns", "labels": {"pool": "a"}}), dict(pod_group("k"), metadata={"name": "k", "namespace": "team", "labels": {"pool": "a"}})],
            params={"partitionParams": {"NodePoolLabelKey": "pool", "NodePoolLabelValue": "a"}})
    got = ingest(d)
    s = got.snapshot
    idx = {n.split("/")[1]: i for i, n in enumerate(s.pod_names)}
    out = json.loads(got.decisions_json([(0, idx["gpu"], 0, 0), (2, idx["victim"], 1, 1), (0, idx['c"pu'], 1, 0), (1, idx["gpu"], 1, 0)]))
    assert [b["spec"] for b in out["bindRequests"]] == [
        {"podName": "gpu", "selectedNode": "n0", "receivedResourceType": "Regular", "receivedGPU": {"count": 2, "portion": "1.00"}},
        {"podName": 'c"pu', "selectedNode": "n1", "receivedResourceType": "Regular", "receivedGPU": {"portion": "0.00"}}]
    b = out["bindRequests"][0]
    assert b["apiVersion"] == "scheduling.run.ai/v1alpha2" and b["kind"] == "BindRequest"
    assert b["metadata"] == {"name": "gpu", "namespace": "ns", "ownerReferences": [{"apiVersion": "v1", "kind": "Pod", "name": "gpu", "uid": "uid-gpu"}],
                             "labels": {"selected-node": "n0", "pool": "a"}}
    assert out["evictions"] == [{"namespace": "ns", "name": "victim", "uid": "uid-victim", "podGroup": {"namespace": "team", "name": "k"}}]
    assert out["pipelined"] == [{"namespace": "ns", "name": "gpu", "node": "n1"}]
    assert json.loads(got.decisions_json([])) == {"bindRequests": [], "evictions": [], "pipelined": []}
    with pytest.raises(ing.IngestError, match="out of range"):
        got.decisions_json([(0, 99, 0, 0)])


def test_replay_closes_the_loop():
    """snapshot → ingest → schedule (oracle here; the GPU test does the same through the C ABI) → BindRequests → fed back as the next snapshot's
    bind requests: the pods come back Binding on the chosen nodes and nothing is left to place (cmd/snapshot-tool's loop through both formats)."""
    d = _rich_document()
    got = ingest(d)
    res = T.Oracle.run(got.snapshot, got.config, tuple(got.actions))
    out = json.loads(got.decisions_json(res.ops))
    assert len(out["bindRequests"]) == sum(1 for o in res.ops if o[0] == 0) > 0
    d["rawObjects"]["bindRequests"] = out["bindRequests"]
    nxt = ingest(d)
    s = nxt.snapshot
    placed = {b["spec"]["podName"]: b["spec"]["selectedNode"] for b in out["bindRequests"]}
    for i, n in enumerate(s.pod_names):
        if n.split("/")[1] in placed:
            assert s.pod_status[i] == ST["Binding"] and s.node_names[s.pod_node[i]] == placed[n.split("/")[1]]
    res2 = T.Oracle.run(s, nxt.config, ("allocate",))
    assert all(o[0] != 0 or s.pod_names[o[1]].split("/")[1] not in placed for o in res2.ops)


@pytest.mark.gpu
def test_gpu_decisions_close_the_loop():
    """n3 on the device: snapshot → ingest → the configured actions on the MI355X through the C ABI → kai_ingest_decisions_json (one document: a BindRequest per
    Allocate as cache.createBindRequest fills it, cache/cache.go:290-330; evictions with their pod group, :216-252) → fed back as the next snapshot's bind requests →
    the pods come back Binding on the chosen nodes and a second cycle on the device places none of them again; the oracle agrees with both cycles."""
    d = _rich_document()
    got = ingest(d)

    def cycle(g, acts):
        ops = []
        with pkg.KaiCore(g.config) as core:
            ssn = core.open_session(g.snapshot)
            for a in acts:
                ops += [(int(o["kind"]), int(o["pod"]), int(o["node"]), int(o["job"])) for o in ssn.execute(a)]
            ssn.close()
        return ops

    ops = cycle(got, tuple(got.actions))
    assert ops == T.Oracle.run(got.snapshot, got.config, tuple(got.actions)).ops and any(o[0] == 0 for o in ops)
    out = json.loads(got.decisions_json(ops))
    assert len(out["bindRequests"]) == sum(1 for o in ops if o[0] == 0) and len(out["evictions"]) == sum(1 for o in ops if o[0] == 2)
    for b, o in zip(out["bindRequests"], [o for o in ops if o[0] == 0]):
        assert b["spec"]["podName"] == got.snapshot.pod_names[o[1]].split("/")[1] and b["spec"]["selectedNode"] == got.snapshot.node_names[o[2]] == b["metadata"]["labels"]["selected-node"]
    d["rawObjects"]["bindRequests"] = out["bindRequests"]
    nxt = ingest(d); s = nxt.snapshot
    placed = {b["spec"]["podName"]: b["spec"]["selectedNode"] for b in out["bindRequests"]}
    for i, n in enumerate(s.pod_names):
        if n.split("/")[1] in placed:
            assert s.pod_status[i] == ST["Binding"] and s.node_names[s.pod_node[i]] == placed[n.split("/")[1]]
    ops2 = cycle(nxt, ("allocate",))
    assert ops2 == T.Oracle.run(s, nxt.config, ("allocate",)).ops
    assert all(o[0] != 0 or s.pod_names[o[1]].split("/")[1] not in placed for o in ops2)


# ------------------------------------------------------------------------------------------------ shared GPUs (ABI v4)
def test_fraction_fields_and_oracle_placement():
    """gpu-fraction annotation → pod_gpu_portion (pod_info.go:472-477), runai-gpu-group label → pod_gpu_group (numeric names keep their value,
    any other name is numbered from 2^20: plugins/predicates/predicates.go:320-330 takes it for a group being created), nvidia.com/gpu.memory →
    node_gpu_memory floored to 100 (node_info.go:673-687).  The scenario of allocateFractionalGpu_test.go:155-214 through the file format:
    the pending half-GPU pod joins the running half on the same shared GPU."""
    frac = lambda v: {"annotations": {"gpu-fraction": v}}
    pods = [pod("run", "j0", requests={"cpu": "1"}, phase="Running", node_name="node0", labels={"runai-gpu-group": "1"}, **frac("0.5")),
            pod("uuid", "j2", requests={"cpu": "1"}, phase="Running", node_name="node1", labels={"runai-gpu-group": "6c3e-uuid"}, **frac("0.25")),
            pod("pend", "j1", requests={"cpu": "1"}, **frac("0.5")), pod("whole", "j3"), pod("multi", "j4", annotations={"gpu-fraction": "0.5", "gpu-fraction-num-devices": "2"})]
    nodes = [node("node0", gpu="2", labels={"nvidia.com/gpu.memory": "40537"}), node("node1", gpu="2")]
    got = ingest(doc(nodes=nodes, queues=[queue("q")], pods=pods, pod_groups=[pod_group(f"j{i}", priorityClassName="p") for i in range(5)],
                     priorityClasses=[{"metadata": {"name": "p"}, "value": 100}]))
    s = got.snapshot
    idx = {n.split("/")[1]: i for i, n in enumerate(s.pod_names)}
    assert [float(s.pod_gpu_portion[idx[n]]) for n in ("run", "uuid", "pend", "whole", "multi")] == [0.5, 0.25, 0.5, 0.0, 0.0]
    assert [int(s.pod_gpu_group[idx[n]]) for n in ("run", "uuid", "pend", "whole", "multi")] == [1, 1 << 20, -1, -1, -1]
    assert float(s.pod_req[abi.RES_GPU, idx["pend"]]) == 0.5 and float(s.pod_req[abi.RES_GPU, idx["whole"]]) == 1.0
    assert list(s.node_gpu_memory) == [40500, 100]
    assert s.pod_flags[idx["multi"]] & abi.POD_CPU_FALLBACK  # several devices per pod: the host path's
    for n in ("run", "uuid", "pend"):
        assert not s.pod_flags[idx[n]] & abi.POD_CPU_FALLBACK  # a fraction of one device is the device's (shared GPUs, ABI v4)
    # drop the pods the oracle does not model (several devices per pod) and let it place the rest
    nodes[1]["metadata"]["labels"]["nvidia.com/gpu.memory"] = "40537"  # one GPU memory size for the cluster: what the engine twin admits
    keep = doc(nodes=nodes, queues=[queue("q")], pods=[p for p in pods if p["metadata"]["name"] != "multi"], pod_groups=[pod_group(f"j{i}", priorityClassName="p") for i in range(4)],
               priorityClasses=[{"metadata": {"name": "p"}, "value": 100}])
    g2 = ingest(keep); s2 = g2.snapshot
    res = T.Oracle.run(s2, g2.config, ("allocate",))
    i2 = {n.split("/")[1]: i for i, n in enumerate(s2.pod_names)}
    assert res.pod_status[i2["pend"]] == ST["Binding"] and s2.node_names[res.pod_node[i2["pend"]]] == "node0" and res.gpu_groups[i2["pend"]] == 1
    assert res.pod_status[i2["whole"]] == ST["Binding"]
    # the host-compiled engine places the same (the device: tests/test_gpu_parity.py)
    import test_engine_hostsim as H
    sim = H.HostSim.run(s2, g2.config, ("allocate",))
    assert sim.ops == res.ops and sim.gpu_groups[i2["pend"]] == 1


def test_fraction_request_is_fixed_point():
    """ResourceRequirements.GPUs() of a fraction is fixed point, 1/100 (api/resource_info/gpu_resource_requirment.go:230-234: math.Round(portion * 100)):
    a gpu-fraction of 0.125 counts as 0.13 against the queue, 0.374 as 0.37 — while the memory it takes on the device stays int64(portion * memory)."""
    frac = lambda v: {"annotations": {"gpu-fraction": v}}
    got = ingest(doc(nodes=[node("node0", gpu="2")], queues=[queue("q")], pod_groups=[pod_group(f"j{i}") for i in range(3)],
                     pods=[pod("a", "j0", requests={"cpu": "1"}, **frac("0.125")), pod("b", "j1", requests={"cpu": "1"}, **frac("0.374")), pod("c", "j2", requests={"cpu": "1"}, **frac("0.5"))]))
    s = got.snapshot
    idx = {n.split("/")[1]: i for i, n in enumerate(s.pod_names)}
    assert [float(s.pod_gpu_portion[idx[n]]) for n in "abc"] == [0.125, 0.374, 0.5]
    assert [float(s.pod_req[abi.RES_GPU, idx[n]]) for n in "abc"] == [0.13, 0.37, 0.5]
    res = T.Oracle.run(s, got.config, ("allocate",))
    q = int(s.job_queue[s.pod_job[idx["a"]]])
    assert abs(res.shares_final["allocated"][q][2] - 1.0) < 1e-12  # 0.13 + 0.37 + 0.5


def test_mig_node_and_legacy_mig_pod():
    """MIG instances on the node side (ResourceFromResourceList, resource_info.go:53-79: instances by Value) and a legacy MIG pod: an annotation named like a
    MIG profile replaces the request by that profile and marks the task (pod_info.go:500-516) — the device never schedules it (node_info.go:317-320)."""
    nodes = [node("n", labels={"nvidia.com/mig.strategy": "mixed", "node-role.kubernetes.io/mig-enabled": "true"})]
    nodes[0]["status"]["allocatable"]["nvidia.com/mig-2g.20gb"] = "3"
    pods = [pod("new", requests={"nvidia.com/mig-2g.20gb": "2"}), pod("legacy", annotations={"nvidia.com/mig-2g.20gb": "1"})]
    s = ingest(doc(nodes=nodes, pods=pods)).snapshot
    names = [n.split("/")[1] for n in s.pod_names]
    row = [k for k in range(4, s.n_res) if s.res_mig_gpus[k] == 2]
    assert len(row) == 1 and s.res_mig_memory[row[0]] == 20 and s.node_allocatable[row[0], 0] == 3
    assert s.pod_req[row[0], names.index("new")] == 2 and s.pod_req[row[0], names.index("legacy")] == 1
    assert s.pod_flags[names.index("legacy")] & abi.POD_LEGACY_MIG and not s.pod_flags[names.index("new")] & abi.POD_LEGACY_MIG
    assert not s.pod_flags[names.index("new")] & abi.POD_CPU_FALLBACK


NODE_CONDITIONS = T.load_golden("kat_node_conditions")


@pytest.mark.parametrize("case", NODE_CONDITIONS["cases"], ids=[f"{c['line']}:{c['test']}" for c in NODE_CONDITIONS["cases"]])
def test_node_condition_predicate_reference_cases(case):
    """CheckNodeConditionPredicate (scheduler_util/scheduler_utils.go:12-40; predicates.go's node-readiness Filter) on the nine cases of scheduler_utils_test.go
    (tools/go_kat_node_conditions.py): the ingest turns a node's conditions and spec.unschedulable into KAI_NODE_NOT_READY, the flag the device path's static predicate reads."""
    n = node("n", spec={"unschedulable": True} if case["unschedulable"] else {}, status={"conditions": [{"type": t, "status": s} for t, s in case["conditions"]]})
    s = ingest(doc(nodes=[n])).snapshot
    assert bool(int(s.node_flags[0]) & abi.NODE_NOT_READY) == (not case["ready"])


MAX_NODE = T.load_golden("kat_max_node_resources")


@pytest.mark.parametrize("case", MAX_NODE["cases"], ids=[f"{c['line']}:{c['name']}" for c in MAX_NODE["cases"]])
def test_max_node_resources_reference_cases(case):
    """MaxNodeResourcesPredicate.PreFilter (k8s_internal/predicates/maxNodeResources.go:59-96: a pod that asks for more of a resource than any ONE node has allocatable is unschedulable)
    on the six non-DRA cases of Test_podToMaxNodeResourcesFiltering (tools/go_kat_max_node_resources.py).  The path does not restate the pre-predicate — the per-node fit reaches the same
    verdict (actions/common/allocate.go:121-163) — so the cases are run END TO END: ingest (quantities, the pod's request over its containers, the extra resource column for ephemeral
    storage, the fraction annotation), then the allocate action in the oracle and on the host-compiled engine: the pod is placed exactly when the reference's PreFilter lets it through
    (the one schedulable case also fits a single node).  For whole-quantity requests the element-wise-maximum statement itself is checked on the ingested arrays."""
    from test_engine_hostsim import HostSim
    nodes = []
    for name, al in case["nodes"].items():
        extra = {k: v for k, v in al.items() if k not in ("cpu", "memory", "pods", "nvidia.com/gpu")}
        nodes.append(node(name, cpu=al["cpu"], mem=al["memory"], gpu=al.get("nvidia.com/gpu"), pods=al["pods"], alloc=extra))
    ann = {k: v for k, v in case["annotations"].items() if k != "pod-group-name"}
    p = pod("name1", group="pg", requests=case["containers"][0], annotations=ann)
    p["spec"]["containers"] = [{"name": f"c{i + 1}", "resources": {"requests": r}} for i, r in enumerate(case["containers"])]
    got = ingest(doc(nodes=nodes, pods=[p], queues=[queue("q")], pod_groups=[pod_group("pg")]))
    s, cfg = got.snapshot, got.config
    assert s.n_pods == 1 and int(s.pod_status[0]) == ST["Pending"] and int(s.pod_job[0]) == 0
    ref = T.Oracle.run(s, cfg, ("allocate",))
    placed = any(o[0] in (0, 1) and o[1] == 0 for o in ref.ops)
    assert placed == case["schedulable"], (case["name"], ref.ops)
    res = HostSim.run(s, cfg, ("allocate",))
    assert [tuple(o) for o in res.ops] == ref.ops
    if "gpu-fraction" not in case["annotations"]:
        fits_max = all(float(s.pod_req[r, 0]) <= float(s.node_allocatable[r].max()) for r in range(s.n_res))
        assert fits_max == case["schedulable"]


def test_a_pod_set_that_is_all_placed_does_not_enter_the_job_signature():
    """podset.go:150-154: a pod-set whose pods are all active-allocated (or that has none) has the EMPTY scheduling-constraints signature, whatever its topology constraint; the job's
    hash (job_info.go:555-569) is over the sorted pod-set signatures written one after the other, so such a pod-set adds nothing to it.  Jobs a / b / c: the same pending `workers`;
    b also has a `servers` pod-set (own rack constraint) whose only pod runs, c has it with that pod still pending."""
    topo = {"metadata": {"name": "t"}, "spec": {"levels": [{"nodeLabel": "zone"}, {"nodeLabel": "rack"}]}}
    servers = {"name": "servers", "minMember": 1, "topologyConstraint": {"topology": "t", "requiredTopologyLevel": "rack"}}
    pgs = [pod_group("a", subGroups=[{"name": "workers", "minMember": 1}]), pod_group("b", subGroups=[{"name": "workers", "minMember": 1}, servers]),
           pod_group("c", subGroups=[{"name": "workers", "minMember": 1}, servers])]
    lab = lambda n: {"kai.scheduler/subgroup-name": n}
    pods = [pod("a-w", "a", labels=lab("workers")), pod("b-w", "b", labels=lab("workers")), pod("b-s", "b", labels=lab("servers"), phase="Running", node_name="n0"),
            pod("c-w", "c", labels=lab("workers")), pod("c-s", "c", labels=lab("servers"))]
    s = ingest(doc(nodes=[node("n0", labels={"zone": "z", "rack": "r"})], queues=[queue("q")], pods=pods, pod_groups=pgs, topologies=[topo])).snapshot
    sig = dict(zip(s.job_names, (int(x) for x in s.job_signature)))
    assert sig["a"] == sig["b"] and sig["c"] != sig["a"]


JOB_SIGNATURE = T.load_golden("kat_job_signature")


def _pod_group_objects(job, tree):
    """a pod group of kat_job_signature.json (the tree the reference's test builds by hand) as snapshot objects: the PodGroup with its sub-group list, its pods"""
    tc = lambda c: {k: v for k, v in (("topology", (c or {}).get("topology")), ("requiredTopologyLevel", (c or {}).get("required")), ("preferredTopologyLevel", (c or {}).get("preferred"))) if v}
    subgroups, pods = [], []
    flat = len(tree["children"]) == 1 and tree["children"][0]["kind"] == "PodSet" and tree["children"][0]["name"] == "default"

    def walk(node, parent):
        for ch in node["children"]:
            e = {"name": ch["name"]}
            if parent: e["parent"] = parent
            if tc(ch["constraint"]): e["topologyConstraint"] = tc(ch["constraint"])
            if ch["kind"] == "PodSet":
                e["minMember"] = ch["minAvailable"]
                for name, state in ch["pods"]:
                    pods.append(pod(f"{job}-{ch['name']}-{name}", job, labels={} if flat else {"kai.scheduler/subgroup-name": ch["name"]},
                                    **({"phase": "Running", "node_name": "n0"} if state == "Running" else {})))
                if not flat: subgroups.append(e)
            else:
                subgroups.append(e); walk(ch, ch["name"])
    walk(tree, None)
    spec = {}
    if tc(tree["constraint"]): spec["topologyConstraint"] = tc(tree["constraint"])
    if subgroups: spec["subGroups"] = subgroups
    return pod_group(job, min_member=tree["children"][0]["minAvailable"] if flat else 1, **spec), pods


@pytest.mark.parametrize("case", JOB_SIGNATURE["cases"], ids=[f"{c['line']}:{c['name']}" for c in JOB_SIGNATURE["cases"]])
def test_job_signature_reference_cases(case):
    """PodGroupInfo.GetSchedulingConstraintsSignature (job_info.go:547-570; podset.go:150-196) on the eleven pairs of TestPodGroupInfo_GetSchedulingConstraintsSignature
    (tools/go_kat_job_signature.py interprets the closures that build them): both pod groups of a pair go into ONE snapshot document as the objects the reference's cache would
    hold, and the ids the ingest gives their jobs (kai_snapshot_soa.job_signature: equal exactly when the reference's hashes are) are equal exactly when the test expects it."""
    levels = [{"nodeLabel": l} for l in ("zone", "rack", "node")]
    topos = [{"metadata": {"name": n}, "spec": {"levels": levels}} for n in ("topo", "topology")]
    pga, pa = _pod_group_objects("a", case["a"]); pgb, pb = _pod_group_objects("b", case["b"])
    s = ingest(doc(nodes=[node("n0", labels={"zone": "z", "rack": "r", "node": "n0"})], queues=[queue("q")], pods=pa + pb, pod_groups=[pga, pgb], topologies=topos)).snapshot
    assert s.n_pods == len(pa) + len(pb) and all(int(j) >= 0 for j in s.pod_job) and all(int(k) >= 0 for k in s.pod_podset)  # every pod found its pod-set
    sig = dict(zip(s.job_names, (int(x) for x in s.job_signature)))
    assert (sig["a"] == sig["b"]) == case["equal"], (sig, s.podset_names, list(s.pod_podset))
