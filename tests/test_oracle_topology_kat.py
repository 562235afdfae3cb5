"""The oracle's topology.subSetNodesFn against plugins/topology/job_filtering_test.go TestTopologyPlugin_subsetNodesFn (:31-643): the first node set a job with a
topology constraint is offered.  The Go cases build the domain tree by hand; here it comes from the nodes' labels and the Topology object, as in the plugin
(topology_plugin.go:57-110).  CPU quantities are cores: the fixtures' "CPUMillis: 1000" is parsed as the quantity "1000" (nodes_fake/nodes.go:78-80)."""
import ctypes as C

import numpy as np
import pytest

import kai_testlib as T

TOPO = [{"ObjectMeta": {"Name": "test-topology"}, "Spec": {"Levels": [{"NodeLabel": "zone"}, {"NodeLabel": "rack"}]}}]
N = lambda cpu, zone=None, rack=None: {"CPUMillis": cpu, "GPUs": 6, "MaxTaskNum": 100, "Labels": {k: v for k, v in (("zone", zone), ("rack", rack)) if v}}
TWO = {"node-1": N(1000, "zone1", "rack1"), "node-2": N(400, "zone1", "rack2")}
CASES = [  # (line, name, cpu cores per task, tasks (RequiredGPUs or None), constraint, nodes, expected first node set | "error" | "none" | "all")
    (47, "right nodes", 500, [None, None], ("test-topology", "zone", "rack"), TWO, {"node-1"}),
    (149, "required equal preferred", 500, [None, None], ("test-topology", "rack", "rack"), TWO, {"node-1"}),
    (251, "no topology constraint - early return", 500, [None], None, {"node-1": N(1000, "zone1")}, "all"),
    (284, "topology not found", 500, [None], ("nonexistent-topology", "", ""), {"node-1": N(1000, "zone1")}, "none"),
    (320, "insufficient allocatable pods - no domains found", 2000, [None], ("test-topology", "zone", ""), {"node-1": N(1000, "zone1")}, "none"),
    (378, "mixed GPU tasks", 2000, [1, 0], ("test-topology", "zone", "rack"), {"node-1": N(2000, "zone1", "rack1"), "node-2": N(2000, "zone1", "rack2")}, {"node-1", "node-2"}),
    (484, "constraint names a level the topology does not have", 500, [None], ("test-topology", "nonexistent-level", "rack"), {"node-1": N(1000, "zone1", "rack1")}, "error"),
]


@pytest.mark.parametrize("line,name,cpu,tasks,tc,nodes,want", CASES, ids=[f"{c[0]}:{c[1].replace(' ', '_')}" for c in CASES])
def test_subset_nodes_first_set(line, name, cpu, tasks, tc, nodes, want):
    root = {"Name": "", "PodSets": [], "SubGroups": [], "TopologyConstraint": {"Topology": tc[0], "RequiredLevel": tc[1], "PreferredLevel": tc[2]} if tc else None}
    case = {"Name": name, "Nodes": nodes, "Topologies": TOPO, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": cpu, "RootSubGroupSet": root,
                      "Tasks": [{"State": "Pending", **({"RequiredGPUs": g} if g is not None else {})} for g in tasks]}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins |= T.abi.PLUGINS["topology"]
    lib = T.Oracle.lib(); lib.kai_oracle_subset_nodes.restype = C.c_int
    out = np.zeros(16, np.int32); n_sets = C.c_int(0); s = snap.as_struct()
    n = lib.kai_oracle_subset_nodes(C.byref(cfg), C.byref(s), snap.job_names.index("test-job"), out.ctypes.data_as(C.POINTER(C.c_int32)), 16, C.byref(n_sets))
    if want == "error": assert n == -1, n
    elif want == "none": assert n == -2 and n_sets.value == 0, n  # no error, and no node set is offered: a fit error on the job (the Go test reads its message)
    elif want == "all": assert n == len(nodes) and n_sets.value == 1
    else: assert {snap.node_names[out[i]] for i in range(n)} == want, (n, n_sets.value)


def test_required_level_follows_the_pods_that_already_run():
    """job_filtering_test.go:1860-1928 (getJobAllocatableDomains "mixed task statuses with required constraint - choose zone with existing pods"): a gang with one pod
    running in zone2 and two pending, required level zone — both zones have room for two pods, only zone2 is offered"""
    topo = [{"ObjectMeta": {"Name": "test-topology"}, "Spec": {"Levels": [{"NodeLabel": "zone"}]}}]
    nodes = {"node1": {"CPUMillis": 2, "GPUs": 0, "MaxTaskNum": 100, "Labels": {"zone": "zone1"}}, "node2": {"CPUMillis": 3, "GPUs": 0, "MaxTaskNum": 100, "Labels": {"zone": "zone2"}}}
    root = {"Name": "", "PodSets": [{"Name": "default", "MinAvailable": 2, "TopologyConstraint": None}], "SubGroups": [],
            "TopologyConstraint": {"Topology": "test-topology", "RequiredLevel": "zone", "PreferredLevel": ""}}
    case = {"Name": "existing pods", "Nodes": nodes, "Topologies": topo, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 1, "RootSubGroupSet": root,
                      "Tasks": [{"State": "Running", "NodeName": "node2"}, {"State": "Pending"}, {"State": "Pending"}]}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins |= T.abi.PLUGINS["topology"]
    lib = T.Oracle.lib(); lib.kai_oracle_subset_nodes.restype = C.c_int
    out = np.zeros(16, np.int32); n_sets = C.c_int(0); s = snap.as_struct()
    n = lib.kai_oracle_subset_nodes(C.byref(cfg), C.byref(s), 0, out.ctypes.data_as(C.POINTER(C.c_int32)), 16, C.byref(n_sets))
    assert n == 1 and snap.node_names[out[0]] == "node2" and n_sets.value == 1


def _node_sets(racks, cpu, gpus):
    """one node per rack of zone1 with the given idle (cores, GPUs); a one-task job with required zone / preferred rack → the node sets in the order they are tried"""
    nodes = {f"node-{r}": {"CPUMillis": c, "GPUs": g, "MaxTaskNum": 100, "Labels": {"zone": "zone1", "rack": r}} for r, (c, g) in racks.items()}
    root = {"Name": "", "PodSets": [], "SubGroups": [], "TopologyConstraint": {"Topology": "test-topology", "RequiredLevel": "zone", "PreferredLevel": "rack"}}
    case = {"Name": "sort", "Nodes": nodes, "Topologies": TOPO, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": cpu, "RequiredGPUsPerTask": gpus, "RootSubGroupSet": root, "Tasks": [{"State": "Pending"}]}],
            "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins |= T.abi.PLUGINS["topology"]
    lib = T.Oracle.lib(); lib.kai_oracle_subset_nodes_all.restype = C.c_int
    out = np.zeros(64, np.int32); s = snap.as_struct()
    n = lib.kai_oracle_subset_nodes_all(C.byref(cfg), C.byref(s), 0, out.ctypes.data_as(C.POINTER(C.c_int32)), 64)
    assert n > 0
    sets, cur = [], []
    for v in out[:n]:
        if v < 0: sets.append(cur); cur = []
        else: cur.append(snap.node_names[v].replace("node-", ""))
    return sets


def test_sort_tree_orders_racks_by_job_to_free_ratio():
    """node_scoring_test.go:294-345 (TestSortTree): the children of a domain are tried fullest first — descending ratio of the job's request to the domain's idle-or-releasing
    resources in the job's dominant resource (job_filtering.go:460-524), GPUs or CPU — so racks with 2 / 5 / 8 free come in that order; the zone itself comes last"""
    assert _node_sets({"rack3": (100, 5), "rack1": (100, 2), "rack2": (100, 8)}, cpu=0.001, gpus=1)[:3] == [["rack1"], ["rack3"], ["rack2"]]
    assert _node_sets({"rack3": (5, 0), "rack1": (2, 0), "rack2": (8, 0)}, cpu=1, gpus=0)[:3] == [["rack1"], ["rack3"], ["rack2"]]


def _topology_scores(racks, n_tasks, cpu):
    """racks: {rack: [cores of each of its nodes]}; a gang of n_tasks x cpu cores, required zone, preferred rack → {node: topology score}"""
    nodes = {f"{r}-n{i}": {"CPUMillis": c, "GPUs": 0, "MaxTaskNum": 100, "Labels": {"zone": "zone1", "rack": r}} for r, cs in racks.items() for i, c in enumerate(cs)}
    root = {"Name": "", "PodSets": [], "SubGroups": [], "TopologyConstraint": {"Topology": "test-topology", "RequiredLevel": "zone", "PreferredLevel": "rack"}}
    case = {"Name": "scores", "Nodes": nodes, "Topologies": TOPO, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": cpu, "RootSubGroupSet": root, "Tasks": [{"State": "Pending"}] * n_tasks}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins |= T.abi.PLUGINS["topology"]
    lib = T.Oracle.lib(); lib.kai_oracle_topology_scores.restype = C.c_int
    out = np.zeros(snap.n_nodes); s = snap.as_struct()
    assert lib.kai_oracle_topology_scores(C.byref(cfg), C.byref(s), 0, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    return {snap.node_names[i]: out[i] for i in range(snap.n_nodes)}


def test_preferred_level_node_scores():
    """node_scoring_test.go:106-257 (TestCalculateNodeScores): the domains of the preferred level in tree order get floor((i + 1) / D * 10) x scores.Topology (10 000) — one
    rack: 10 for its nodes; three racks that can take 1 / 2 / 3 pods (fullest first in the sorted tree): 3, 6, 10; four racks: 2, 5, 7, 10"""
    K = 10000.0
    assert _topology_scores({"rack1": [1, 1]}, 1, 1) == {"rack1-n0": 10 * K, "rack1-n1": 10 * K}
    assert _topology_scores({"rack3": [1], "rack2": [2], "rack1": [3]}, 1, 1) == {"rack1-n0": 10 * K, "rack2-n0": 6 * K, "rack3-n0": 3 * K}
    assert _topology_scores({"rack4": [1], "rack3": [2], "rack2": [3], "rack1": [4]}, 1, 1) == {"rack1-n0": 10 * K, "rack2-n0": 7 * K, "rack3-n0": 5 * K, "rack4-n0": 2 * K}


def test_releasing_resources_count_towards_a_domain():
    """job_filtering_test.go:1260-1327 (calcTreeAllocatable "Can pipeline on domain with releasing pods"): each rack's node has 500 cores idle and 500 being released; a gang
    of 2 x 500 with the rack REQUIRED fits a rack only if releasing resources count (AllocatablePods 2 per rack) — both racks are offered, fullest-first ties by ID"""
    nodes = {"node-1": {"CPUMillis": 1000, "GPUs": 6, "MaxTaskNum": 100, "Labels": {"zone": "zone1", "rack": "rack1"}},
             "node-2": {"CPUMillis": 1000, "GPUs": 6, "MaxTaskNum": 100, "Labels": {"zone": "zone1", "rack": "rack2"}}}
    root = {"Name": "", "PodSets": [], "SubGroups": [], "TopologyConstraint": {"Topology": "test-topology", "RequiredLevel": "rack", "PreferredLevel": ""}}
    case = {"Name": "releasing", "Nodes": nodes, "Topologies": TOPO, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 500, "RootSubGroupSet": root, "Tasks": [{"State": "Pending"}] * 2},
                     {"Name": "running-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 500,
                      "Tasks": [{"State": "Releasing", "NodeName": "node-1"}, {"State": "Releasing", "NodeName": "node-2"}]}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins |= T.abi.PLUGINS["topology"]
    lib = T.Oracle.lib(); lib.kai_oracle_subset_nodes_all.restype = C.c_int
    out = np.zeros(16, np.int32); s = snap.as_struct()
    n = lib.kai_oracle_subset_nodes_all(C.byref(cfg), C.byref(s), snap.job_names.index("test-job"), out.ctypes.data_as(C.POINTER(C.c_int32)), 16)
    assert [snap.node_names[v] if v >= 0 else None for v in out[:n]] == ["node-1", None, "node-2", None]


def test_a_job_that_requests_nothing_fits_every_domain():
    """job_filtering_test.go:1329-1380 ("Job requests 0 resources - set maximal amount of pods on each node"): a best-effort gang of four with the rack required — every
    node can take all of its tasks, so both racks are offered"""
    nodes = {"node-1": {"CPUMillis": 1000, "GPUs": 6, "MaxTaskNum": 100, "Labels": {"zone": "zone1", "rack": "rack1"}},
             "node-2": {"CPUMillis": 1000, "GPUs": 6, "MaxTaskNum": 100, "Labels": {"zone": "zone1", "rack": "rack2"}}}
    root = {"Name": "", "PodSets": [], "SubGroups": [], "TopologyConstraint": {"Topology": "test-topology", "RequiredLevel": "rack", "PreferredLevel": ""}}
    case = {"Name": "zero", "Nodes": nodes, "Topologies": TOPO, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": "test-job", "Priority": 50, "QueueName": "q", "IsBestEffortJob": True, "RootSubGroupSet": root, "Tasks": [{"State": "Pending"}] * 4}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins |= T.abi.PLUGINS["topology"]
    lib = T.Oracle.lib(); lib.kai_oracle_subset_nodes_all.restype = C.c_int
    out = np.zeros(16, np.int32); s = snap.as_struct()
    n = lib.kai_oracle_subset_nodes_all(C.byref(cfg), C.byref(s), 0, out.ctypes.data_as(C.POINTER(C.c_int32)), 16)
    assert [snap.node_names[v] if v >= 0 else None for v in out[:n]] == ["node-1", None, "node-2", None]


# ------------------------------------------------------------------------------------------------ the same scenes through the allocate action: engine against oracle
def _scene(nodes, jobs, topo=TOPO):
    case = {"Name": "scene", "Nodes": nodes, "Topologies": topo, "Queues": [{"Name": "q", "DeservedGPUs": 1}], "Jobs": jobs, "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins |= T.abi.PLUGINS["topology"]
    return snap, cfg


def _root(required, preferred=""):
    return {"Name": "", "PodSets": [], "SubGroups": [], "TopologyConstraint": {"Topology": "test-topology", "RequiredLevel": required, "PreferredLevel": preferred}}


def _rack_node(cpu, rack, gpus=6):
    return {"CPUMillis": cpu, "GPUs": gpus, "MaxTaskNum": 100, "Labels": {"zone": "zone1", "rack": rack}}


SCENES = {
    "right nodes": lambda: _scene(TWO, [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 500, "RootSubGroupSet": _root("zone", "rack"), "Tasks": [{"State": "Pending"}] * 2}]),
    "mixed GPU tasks": lambda: _scene({"node-1": _rack_node(2000, "rack1"), "node-2": _rack_node(2000, "rack2")},
                                      [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 2000, "RootSubGroupSet": _root("zone", "rack"),
                                        "Tasks": [{"State": "Pending", "RequiredGPUs": 1}, {"State": "Pending", "RequiredGPUs": 0}]}]),
    "releasing counts": lambda: _scene({"node-1": _rack_node(1000, "rack1"), "node-2": _rack_node(1000, "rack2")},
                                       [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 500, "RootSubGroupSet": _root("rack"), "Tasks": [{"State": "Pending"}] * 2},
                                        {"Name": "running-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 500,
                                         "Tasks": [{"State": "Releasing", "NodeName": "node-1"}, {"State": "Releasing", "NodeName": "node-2"}]}]),
    "requests nothing": lambda: _scene({"node-1": _rack_node(1000, "rack1"), "node-2": _rack_node(1000, "rack2")},
                                       [{"Name": "test-job", "Priority": 50, "QueueName": "q", "IsBestEffortJob": True, "RootSubGroupSet": _root("rack"), "Tasks": [{"State": "Pending"}] * 4}]),
    "racks fullest first": lambda: _scene({"node-rack3": _rack_node(100, "rack3", 5), "node-rack1": _rack_node(100, "rack1", 2), "node-rack2": _rack_node(100, "rack2", 8)},
                                          [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredGPUsPerTask": 1, "RootSubGroupSet": _root("zone", "rack"), "Tasks": [{"State": "Pending"}] * 2}]),
    "no room in the zone": lambda: _scene({"node-1": {"CPUMillis": 1000, "GPUs": 6, "MaxTaskNum": 100, "Labels": {"zone": "zone1"}}},
                                          [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 2000, "RootSubGroupSet": _root("zone"), "Tasks": [{"State": "Pending"}]}]),
}


def _same(res, ref):
    assert res.ops == ref.ops and np.array_equal(res.pod_status, ref.pod_status) and np.array_equal(res.pod_node, ref.pod_node)
    for k in ref.nodes: assert np.array_equal(res.nodes[k], ref.nodes[k]), k


@pytest.mark.parametrize("scene", sorted(SCENES))
def test_scenes_allocate_on_the_host_compiled_engine(scene):
    from test_engine_hostsim import HostSim
    snap, cfg = SCENES[scene]()
    _same(HostSim.run(snap, cfg, ("allocate",)), T.Oracle.run(snap, cfg, ("allocate",)))


@pytest.mark.gpu
@pytest.mark.parametrize("scene", sorted(SCENES))
def test_gpu_scenes_allocate(scene):
    import torch
    assert torch.cuda.is_available()
    from test_gpu_parity import run_gpu
    snap, cfg = SCENES[scene]()
    _same(run_gpu(snap, cfg, ("allocate",)), T.Oracle.run(snap, cfg, ("allocate",)))


# job_filtering_test.go TestTopologyPlugin_calcTreeAllocatable (:979-1448): AllocatablePods of every domain after the roll-up.  The Go cases hand a hand-built tree to
# calcTreeAllocatable; here the tree comes from the nodes' labels and the job asks for the zone (the tree's top level), so the same roll-up runs inside subSetNodesFn.
TREE_CASES = [  # (line, name, cores per task, tasks, nodes {name: (cores, zone, rack)}, expected {frozenset of node names: AllocatablePods})
    (1043, "parent takes child values when children can allocate full job", 500, 2, {"node-1": (1000, "zone1", "rack1"), "node-2": (1000, "zone1", "rack2")},
     {("node-1",): 2, ("node-2",): 2, ("node-1", "node-2"): 4}),
    (1096, "children cannot allocate full job individually - parent sums allocations", 800, 2, {"node-1": (1000, "zone1", "rack1"), "node-2": (1000, "zone1", "rack2")},
     {("node-1",): 1, ("node-2",): 1, ("node-1", "node-2"): 2}),
    (1149, "mixed distances - parent takes minimum distance", 500, 2, {"node-1": (500, "zone1", "rack1"), "node-2": (500, "zone1", "rack1"), "node-3": (1000, "zone1", "rack2")},
     {("node-1", "node-2"): 2, ("node-3",): 2, ("node-1", "node-2", "node-3"): 4}),
    (1208, "no leaf domains - no allocatable domains", 2000, 1, {"node-1": (1000, "zone1", None)}, {("node-1",): 0}),
]


@pytest.mark.parametrize("line,name,cpu,n_tasks,nodes,want", TREE_CASES, ids=[f"{c[0]}" for c in TREE_CASES])
def test_calc_tree_allocatable(line, name, cpu, n_tasks, nodes, want):
    two_levels = any(r for _, _, r in nodes.values())
    topo = TOPO if two_levels else [{"ObjectMeta": {"Name": "test-topology"}, "Spec": {"Levels": [{"NodeLabel": "zone"}]}}]
    root = {"Name": "", "PodSets": [], "SubGroups": [], "TopologyConstraint": {"Topology": "test-topology", "RequiredLevel": "zone", "PreferredLevel": ""}}
    case = {"Name": name, "Nodes": {k: N(c, z, r) for k, (c, z, r) in nodes.items()}, "Topologies": topo, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": cpu, "RootSubGroupSet": root, "Tasks": [{"State": "Pending"}] * n_tasks}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins |= T.abi.PLUGINS["topology"]
    lib = T.Oracle.lib(); lib.kai_oracle_tree_allocatable.restype = C.c_int
    nn = snap.n_nodes; pods = np.zeros(32, np.int32); member = np.zeros(32 * nn, np.uint8); s = snap.as_struct()
    nd = lib.kai_oracle_tree_allocatable(C.byref(cfg), C.byref(s), snap.job_names.index("test-job"), pods.ctypes.data_as(C.POINTER(C.c_int32)), member.ctypes.data_as(C.POINTER(C.c_uint8)), 32)
    assert nd > 0, nd
    got = {}
    for d in range(nd):
        names = tuple(sorted(snap.node_names[k] for k in range(nn) if member[d * nn + k]))
        if names and int(pods[d]) != -1: got.setdefault(names, set()).add(int(pods[d]))  # (-1 = allocatablePodsNotSet: the root domain above the zone the roll-up started at)
    for names, n in want.items():
        assert got.get(tuple(sorted(names))) == {n}, (name, names, got)


# plugins/topology/common_test.go TestLowestCommonDomainID (:66-166) and TestIsNodePartOfTopology (:25-64): levels zone, rack; a node that misses a level label is outside the
# topology (the packers apply that rule when they build node_domain: kai_testlib.py, kai_ingest.cpp, shim/kai_cgo_classes.go).  Expected: the level of the common domain
# ("rack" / "zone" / root), the nodes that share it (what the ID z1.r1 stands for) and the valid nodes.
LCD_CASES = [  # (line, name, nodes {name: labels}, preferred level, expected level, expected ID parts, expected valid nodes)
    (81, "all nodes share full topology", {"node-1": {"zone": "z1", "rack": "r1"}, "node-2": {"zone": "z1", "rack": "r1"}}, "", "rack", ("z1", "r1"), ["node-1", "node-2"]),
    (92, "all nodes share full topology - but the preferred level is zone", {"node-1": {"zone": "z1", "rack": "r1"}, "node-2": {"zone": "z1", "rack": "r1"}}, "zone", "zone", ("z1",), ["node-1", "node-2"]),
    (105, "mismatch at deeper level returns common prefix", {"node-1": {"zone": "z1", "rack": "r1"}, "node-2": {"zone": "z1", "rack": "r2"}}, "", "zone", ("z1",), ["node-1", "node-2"]),
    (116, "mismatch at first level returns root", {"node-1": {"zone": "z1", "rack": "r1"}, "node-2": {"zone": "z2", "rack": "r1"}}, "", "root", (), ["node-1", "node-2"]),
    (127, "invalid nodes are filtered out", {"node-1": {"zone": "z1", "rack": "r1"}, "node-2": {"zone": "z1"}}, "", "rack", ("z1", "r1"), ["node-1"]),
    (138, "no valid nodes returns root and empty map", {"node-1": {"zone": "z1"}}, "", "root", (), []),
]


@pytest.mark.parametrize("line,name,nodes,pref,want_level,want_id,want_valid", LCD_CASES, ids=[f"{c[0]}" for c in LCD_CASES])
def test_lowest_common_domain(line, name, nodes, pref, want_level, want_id, want_valid):
    root = {"Name": "", "PodSets": [], "SubGroups": [], "TopologyConstraint": {"Topology": "test-topology", "RequiredLevel": "", "PreferredLevel": pref}}
    case = {"Name": name, "Nodes": {k: {"CPUMillis": 1000, "GPUs": 6, "MaxTaskNum": 100, "Labels": lb} for k, lb in nodes.items()}, "Topologies": TOPO, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": "test-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 100, "RootSubGroupSet": root, "Tasks": [{"State": "Pending"}]}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins |= T.abi.PLUGINS["topology"]
    lib = T.Oracle.lib(); lib.kai_oracle_lowest_common_domain.restype = C.c_int
    nn = snap.n_nodes; level = C.c_int32(-7); member = np.zeros(nn, np.uint8); valid = np.zeros(nn, np.uint8); s = snap.as_struct()
    rc = lib.kai_oracle_lowest_common_domain(C.byref(cfg), C.byref(s), snap.job_names.index("test-job"), C.byref(level), member.ctypes.data_as(C.POINTER(C.c_uint8)), valid.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0, rc
    assert {snap.node_names[k] for k in range(nn) if valid[k]} == set(want_valid)
    assert level.value == {"root": -1, "zone": 0, "rack": 1}[want_level]
    if want_id:  # the domain z1(.r1) = the nodes that carry those label values on every level
        in_domain = {k for k, lb in nodes.items() if len(lb) == 2 and tuple(lb[l] for l in ("zone", "rack"))[:len(want_id)] == want_id}
        assert {snap.node_names[k] for k in range(nn) if member[k]} == in_domain


# ------------------------------------------------------------------------------------------------ TopologyAwareIdleGpus (topology_aware_idle_gpus_test.go), tools/go_kat_topo_idle_gpus.py
import json, os  # noqa: E402
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_topo_idle_gpus.json")) as _fh:
    TOPO_IDLE = json.load(_fh)["cases"]


def _topo_idle_case(case):
    """a fixture case as a scene of the reference's test framework: the Topology object lists the label keys the case's constraints name (top level first — the
    test file's filter reads the labels directly, a session needs the object), the pending job carries one SubGroupSet per newConstrainedSubGroup, every victim is a
    Running pod of a job of its own"""
    levels = []
    for sg in case["subgroups"]:
        if sg["required_level"] not in levels: levels.append(sg["required_level"])
    levels.sort(key=lambda k: (not k.endswith("zone"), k))  # zone above rack where a case has both (TestTopologyAwareIdleGpus_MultipleLevels)
    topo = [{"ObjectMeta": {"Name": "cluster-topology"}, "Spec": {"Levels": [{"NodeLabel": k} for k in levels]}}] if levels else []
    nodes = {n: {"CPUMillis": 1000, "GPUs": v["gpus"], "MaxTaskNum": 100, "Labels": v["labels"]} for n, v in case["nodes"].items()}
    root = {"Name": "", "PodSets": [], "TopologyConstraint": None,
            "SubGroups": [{"Name": sg["name"], "PodSets": [{"Name": sg["name"], "MinAvailable": sg["pods"], "TopologyConstraint": None}], "SubGroups": [],
                           "TopologyConstraint": {"Topology": sg["topology"], "RequiredLevel": sg["required_level"], "PreferredLevel": ""}} for sg in case["subgroups"]]}
    jobs = [{"Name": "pending-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 0, **({"RootSubGroupSet": root} if case["subgroups"] else {}),
             "Tasks": [{"State": "Pending", "RequiredGPUs": t["gpus"], **({"SubGroupName": t["subgroup"]} if case["subgroups"] else {})} for t in case["tasks"]]}]
    for v in case["victims"]:
        jobs.append({"Name": v["name"] + "-job", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 0, "Tasks": [{"State": "Running", "NodeName": v["node"], "RequiredGPUs": v["gpus"]}]})
    return {"Name": case["name"], "Nodes": nodes, "Topologies": topo, "Queues": [{"Name": "q", "DeservedGPUs": 1}], "Jobs": jobs, "JobExpectedResults": {}}


@pytest.mark.parametrize("case", TOPO_IDLE, ids=[f"{c['line']}:{c['name'].split('_', 1)[1]}" for c in TOPO_IDLE])
def test_topology_aware_idle_gpus_filter(case):
    """accumulated_scenario_filters/idle_gpus/topology_aware_idle_gpus.go against its own twelve tests: whether a filter is created, and what it answers to every
    scenario it is asked about, in the test's order (a victim counted once over two calls; recorded victims counted; a domain that moves two places up the
    capacity order; the greedy match over several sub-groups of one level, fragmentation included)."""
    snap, cfg, _ = T.case_to_snapshot(_topo_idle_case(case))
    cfg.plugins |= T.abi.PLUGINS["topology"]
    pod_of = {}
    for v in case["victims"]:
        j = snap.job_names.index(v["name"] + "-job")
        pod_of[v["name"]] = (int(np.nonzero(snap.arrays["pod_job"] == j)[0][0]), j)
    pot_off, pot, rec_off, rec = [0], [], [0], []
    for c in case["calls"]:
        pot += [pod_of[v][0] for v in c["potential"]]; pot_off.append(len(pot))
        rec += [pod_of[v][1] for v in c["recorded"]]; rec_off.append(len(rec))
    i32 = lambda v: (C.c_int32 * max(len(v), 1))(*v)
    lib = T.Oracle.lib(); lib.kai_oracle_topo_idle_gpus_kat.restype = C.c_int
    out = (C.c_int32 * len(case["calls"]))(); s = snap.as_struct()
    r = lib.kai_oracle_topo_idle_gpus_kat(C.byref(cfg), C.byref(s), snap.job_names.index("pending-job"), len(case["calls"]), i32(pot_off), i32(pot), i32(rec_off), i32(rec), out)
    if not case["want_filter"]:
        assert r == -1, r
        return
    assert r == len(case["calls"]), r
    assert [bool(x) for x in out] == [c["want"] for c in case["calls"]], (case["name"], list(out))


LEVEL_ORDER = T.load_golden("kat_level_order")


@pytest.mark.parametrize("case", LEVEL_ORDER["cases"], ids=[f"{c['line']}:{c['name']}" for c in LEVEL_ORDER["cases"]])
def test_reverse_level_order(case):
    """reverseLevelOrder (plugins/topology/topology_utils.go:20-55) on the five trees of TestReverseLevelOrder (topology_utils_test.go:12-140; tools/go_kat_level_order.py): the order in
    which the domains that can hold a job are tried (job_filtering.go:526-542) — the oracle's SubsetNodesFn walks the sorted domain tree with the same function."""
    lib = T.Oracle.lib(); lib.kai_oracle_reverse_level_order.restype = C.c_int
    n = len(case["ids"])
    off = np.zeros(n + 1, np.int32); flat = []
    for d, kids in enumerate(case["children"]):
        off[d + 1] = off[d] + len(kids); flat += kids
    flat = np.asarray(flat + [0], np.int32); out = np.full(max(n, 1), -1, np.int32)
    got = lib.kai_oracle_reverse_level_order(n, off.ctypes.data_as(C.POINTER(C.c_int32)), flat.ctypes.data_as(C.POINTER(C.c_int32)), 0 if n else -1, out.ctypes.data_as(C.POINTER(C.c_int32)), max(n, 1))
    if case["expected"] is None:
        assert got == 0
    else:
        assert got == len(case["expected"]) and [case["ids"][i] for i in out[:got]] == case["expected"]
