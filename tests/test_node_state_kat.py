"""Node accounting at session open against the reference's own expectations (api/node_info/node_info_test.go TestAddRemovePods :393-676): what adding
releasing, running and pipelined fraction pods to a node leaves in Idle / Used / Releasing (NodeInfo.addTaskResources :457-493 with the shared-GPU rules of
gpu_sharing_node_info.go).  Checked on the oracle, on the host-compiled engine and (-m gpu) on the MI355X: the snapshot is opened and nothing is run."""
import numpy as np
import pytest

import kai_testlib as T
from test_gpu_parity import gpu  # noqa: F401 — the fixture that skips without a device

G = 1e9
# (name, pods = (milli-cpu cores, memory bytes, fraction of the device, status, gpu group), expected Idle / Used / Releasing as (milli-cpu, memory, gpus, pods))
CASES = [
    ("releasing pod", [(1.0, 1 * G, 0.5, "Releasing", "1")],
     (7000, 9 * G, 0, 109), (1000, 1 * G, 0, 1), (1000, 1 * G, 1, 1)),                                   # :410-455
    ("pipelined pod - different gpus", [(1.0, 1 * G, 0.5, "Releasing", "1"), (0.5, 1 * G, 0.5, "Pipelined", "2")],
     (7000, 9 * G, 0, 109), (1500, 2 * G, 0, 2), (500, 0, 0, 0)),                                        # :457-531
    ("pipelined pod - same gpus", [(1.0, 1 * G, 0.5, "Releasing", "1"), (1.0, 1 * G, 0.2, "Running", "1"), (0.5, 1 * G, 0.5, "Pipelined", "1")],
     (6000, 8 * G, 0, 108), (2500, 3 * G, 0, 3), (500, 0, 0, 0)),                                        # :532-610
]


def build(pods):
    case = {"Name": "node state", "Nodes": {"n1": {"GPUs": 1, "CPUMillis": 8, "CPUMemory": 10 * G, "MaxTaskNum": 110}},
            "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": f"j{i}", "Priority": 50, "QueueName": "q", "RequiredGPUsPerTask": frac, "RequiredCPUsPerTask": cpu, "RequiredMemoryPerTask": mem,
                      "Tasks": [{"State": st, "NodeName": "n1", "GPUGroups": [grp]}]} for i, (cpu, mem, frac, st, grp) in enumerate(pods)],
            "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case, fractions=True)
    return snap, cfg


def check(res, idle, used, releasing, who):
    for name, want in (("idle", idle), ("used", used), ("releasing", releasing)):
        got = res.nodes[name][0]
        assert np.array_equal(got[:4], np.array(want, np.float64)), f"{who}: {name} {got[:4].tolist()} want {list(want)}"


@pytest.mark.parametrize("name,pods,idle,used,releasing", CASES, ids=[c[0].replace(" ", "_") for c in CASES])
def test_add_pods_node_state(name, pods, idle, used, releasing):
    from test_engine_hostsim import HostSim
    snap, cfg = build(pods)
    check(T.Oracle.run(snap, cfg, ()), idle, used, releasing, "oracle")
    check(HostSim.run(snap, cfg, ()), idle, used, releasing, "host-compiled engine")


@pytest.mark.gpu
@pytest.mark.parametrize("name,pods,idle,used,releasing", CASES, ids=[c[0].replace(" ", "_") for c in CASES])
def test_gpu_add_pods_node_state(gpu, name, pods, idle, used, releasing):
    from test_gpu_parity import run_gpu
    snap, cfg = build(pods)
    check(run_gpu(snap, cfg, ()), idle, used, releasing, "MI355X")


# ------------------------------------------------------------------------------------------------ NodeInfo.IsTaskAllocatable (node_info_test.go:677-794)
M = 1e6
# (name, node (cores, memory, gpus), pods running there (cores, memory, gpus), the task (cores, memory, gpus; a pod's overhead is part of its request,
#  pod_info.go:373-393), allocatable)
FIT = [
    ("not enough cpu and memory", (2, 2 * G, 0), [(1, 1 * G, 0)], (2, 2 * G, 0), False),
    ("not enough cpu - 1 millicpu", (2, 2 * G, 0), [(1, 1 * G, 0)], (1.001, 1 * G, 0), False),
    ("not enough memory - 1 Kb", (2, 2 * G, 0), [(1, 2 * G, 0)], (1, 1024, 0), False),
    ("enough cpu and memory", (2, 2 * G, 0), [(1, 1 * G, 0)], (1, 1 * G, 0), True),
    ("missing gpu", (2, 2 * G, 0), [], (1, 1 * G, 1), False),
    ("missing gpu - requesting a fraction", (2, 2 * G, 0), [], (1, 1 * G, 0.5), False),
    ("already used gpu so missing gpu", (2, 2 * G, 1), [(1, 1 * G, 1)], (1, 1 * G, 1), False),
    ("enough cpu memory and gpu", (2, 2 * G, 2), [(1, 1 * G, 1)], (1, 1 * G, 1), True),
    ("overhead: fits without it, not with it", (2, 2 * G, 0), [(1, 1 * G, 0)], (0.5 + 0.6, 500 * M + 600 * M, 0), False),
    ("overhead: does not fit even without it", (2, 2 * G, 0), [(1, 1 * G, 0)], (1.5 + 0.1, 1500 * M + 100 * M, 0), False),
    ("without overhead, does not fit", (2, 2 * G, 0), [(1, 1 * G, 0)], (1.5, 1500 * M, 0), False),
    ("overhead: fits with it", (2, 2 * G, 0), [(1, 1 * G, 0)], (0.5 + 0.1, 500 * M + 100 * M, 0), True),
    # TestIsTaskAllocatableOnReleasingOrIdle :796-833: a CPU-only task on MIG-enabled nodes (MigStrategy single / mixed)
    ("cpu only job on single mig node", (2, 2 * G, 8, "single"), [(0.2, 1, 0)], (0.2, 1, 0), True),
    ("cpu only job on mixed mig node", (2, 2 * G, 8, "mixed"), [(0.2, 1, 0)], (0.2, 1, 0), True),
]


def build_fit(node, running, task):
    def job(name, r, state):
        return {"Name": name, "Priority": 50, "QueueName": "q", "RequiredGPUsPerTask": r[2], "RequiredCPUsPerTask": r[0], "RequiredMemoryPerTask": r[1],
                "Tasks": [{"State": state, **({"NodeName": "n1"} if state != "Pending" else {})}]}
    case = {"Name": "fit", "Nodes": {"n1": {"GPUs": node[2], "CPUMillis": node[0], "CPUMemory": node[1], "MaxTaskNum": 110, **({"MigStrategy": node[3]} if len(node) > 3 else {})}},
            "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [job(f"run{i}", r, "Running") for i, r in enumerate(running)] + [job("task", task, "Pending")], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case, fractions=True)
    return snap, cfg, snap.pod_names.index("task-0")


def oracle_fits(snap, cfg, pod):
    import ctypes as C
    lib = T.Oracle.lib(); lib.kai_oracle_best_node.restype = C.c_int
    s = snap.as_struct(); node, pipe = C.c_int(-1), C.c_int(0)
    assert lib.kai_oracle_best_node(C.byref(cfg), C.byref(s), pod, None, 0, C.byref(node), C.byref(pipe)) == 0
    return node.value == 0


@pytest.mark.parametrize("name,node,running,task,want", FIT, ids=[c[0].replace(" ", "_") for c in FIT])
def test_is_task_allocatable(name, node, running, task, want):
    """the resource fit of FittingNode on a one-node cluster without releasing pods = NodeInfo.IsTaskAllocatable (node_info.go:168-188), against node_info_test.go:677-794"""
    snap, cfg, pod = build_fit(node, running, task)
    assert oracle_fits(snap, cfg, pod) == want


@pytest.mark.gpu
@pytest.mark.parametrize("name,node,running,task,want", FIT, ids=[c[0].replace(" ", "_") for c in FIT])
def test_gpu_is_task_allocatable(gpu, name, node, running, task, want):
    snap, cfg, pod = build_fit(node, running, task)
    with T.pkg.KaiCore(cfg) as core:
        ssn = core.open_session(snap)
        got = ssn.best_node(pod)
        ssn.close()
    assert (got[0] == 0) == want


# ------------------------------------------------------------------------------------------------ GetSumOfIdleGPUs / GetSumOfReleasingGPUs (node_info_test.go:1052-1283)
# (node GPUs, pods (gpus or fraction, group), releasing, expected sum of GPUs, expected GPU memory at 100 MiB per device)
GPU_SUMS = [
    (2, [], False, 2, 200), (8, [(2, None), (1, None)], False, 5, 500), (8, [(0.5, "1"), (0.1, "2")], False, 7.4, 740), (8, [(0.5, "1"), (0.1, "1")], False, 7.4, 740),
    (8, [(2, None), (0.5, "1"), (1, None), (0.1, "2")], False, 4.4, 440),
    (2, [], True, 0, 0), (8, [(2, None), (1, None)], True, 3, 300), (8, [(0.5, "1"), (0.1, "2")], True, 2, 200), (8, [(0.5, "1"), (0.1, "1")], True, 1, 100),
    (8, [(2, None), (0.5, "1"), (1, None)], True, 4, 400),
]


@pytest.mark.parametrize("gpus,pods,releasing,want,want_mem", GPU_SUMS)
def test_sum_of_idle_and_releasing_gpus(gpus, pods, releasing, want, want_mem):
    """whole devices plus what is free (or being released) on shared ones — what the idle-GPU scenario filters and FeasibleNodes read"""
    import ctypes as C
    case = {"Name": "sums", "Nodes": {"n1": {"GPUs": gpus, "CPUMillis": 8, "CPUMemory": 10 * G}}, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": f"j{i}", "Priority": 50, "QueueName": "q", "RequiredGPUsPerTask": g,
                      "Tasks": [{"State": "Releasing" if releasing else "Running", "NodeName": "n1", **({"GPUGroups": [grp]} if grp else {})}]} for i, (g, grp) in enumerate(pods)],
            "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case, fractions=True)
    lib = T.Oracle.lib(); lib.kai_oracle_node_gpu_sums.restype = C.c_int
    out = np.zeros(4); s = snap.as_struct()
    assert lib.kai_oracle_node_gpu_sums(C.byref(cfg), C.byref(s), out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    got, got_mem = (out[2], out[3]) if releasing else (out[0], out[1])
    assert got == want and got_mem == want_mem, out.tolist()


# ------------------------------------------------------------------------------------------------ isTaskAllocatableOnNonAllocatedResources with a GPU-memory request (node_info_test.go:911-1050)
@pytest.mark.parametrize("mib,want", [(1500, False), (1000, True)])
def test_gpu_memory_request_against_device_size(mib, want):
    """a request for more MiB than one device has is an invalid portion (isValidGpuPortion, node_info.go:668-671); exactly one device's worth fits on an idle GPU"""
    case = {"Name": "gm", "Nodes": {"n1": {"GPUs": 2, "GPUMemory": 1000, "CPUMillis": 8, "CPUMemory": 10 * G, "MaxTaskNum": 10}}, "Queues": [{"Name": "q", "DeservedGPUs": 2}],
            "Jobs": [{"Name": "task", "Priority": 50, "QueueName": "q", "RequiredGpuMemory": mib, "Tasks": [{"State": "Pending"}]}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case, fractions=True)
    cfg.min_node_gpu_memory = 1000
    assert oracle_fits(snap, cfg, snap.pod_names.index("task-0")) == want


# ------------------------------------------------------------------------------------------------ the node-order plugins one at a time (plugins/*/…_test.go)
def _score(case, plugin, task="task-0", node="n1", nominated=None):
    import ctypes as C
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins = T.abi.PLUGINS[plugin]
    if nominated is not None: snap.arrays["pod_nominated_node"][snap.pod_names.index(task)] = snap.node_names.index(nominated)
    lib = T.Oracle.lib(); lib.kai_oracle_node_score.restype = C.c_double
    s = snap.as_struct()
    return lib.kai_oracle_node_score(C.byref(cfg), C.byref(s), snap.pod_names.index(task), snap.node_names.index(node))


def _one_task(gpus, nodes, running=()):
    return {"Name": "score", "Nodes": nodes, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
            "Jobs": [{"Name": "task", "Priority": 50, "QueueName": "q", "RequiredGPUsPerTask": gpus, "Tasks": [{"State": "Pending"}]}] +
                    [{"Name": f"run{i}", "Priority": 50, "QueueName": "q", "RequiredGPUsPerTask": g, "Tasks": [{"State": "Running", "NodeName": "n1"}]} for i, g in enumerate(running)],
            "JobExpectedResults": {}}


def test_node_availability_score():
    """nodeavailability_test.go:51-66: scores.Availability (100) when the task fits the node's idle resources (one GPU idle of two), 0 when it does not"""
    assert _score(_one_task(1, {"n1": {"GPUs": 2}}, running=(1,)), "nodeavailability") == 100.0
    assert _score(_one_task(2, {"n1": {"GPUs": 2}}, running=(1,)), "nodeavailability") == 0.0


def test_resource_type_score():
    """resourcetype_test.go:52-84: scores.ResourceType (10) only for a CPU-only task on a node without GPUs — not for a GPU task there, not on a GPU or MIG node"""
    assert _score(_one_task(0, {"n1": {"GPUs": 0}}), "resourcetype") == 10.0
    assert _score(_one_task(1, {"n1": {"GPUs": 0}}), "resourcetype") == 0.0
    assert _score(_one_task(0, {"n1": {"GPUs": 2}}), "resourcetype") == 0.0
    assert _score(_one_task(1, {"n1": {"GPUs": 2}}), "resourcetype") == 0.0


def test_nominated_node_score():
    """nominatednode_test.go:25-83: scores.NominatedNode (1e6) iff the pod's nominated node is this node"""
    nodes = {"n1": {"GPUs": 2}, "other-node": {"GPUs": 2}}
    assert _score(_one_task(1, nodes), "nominatednode") == 0.0
    assert _score(_one_task(1, nodes), "nominatednode", nominated="other-node") == 0.0
    assert _score(_one_task(1, nodes), "nominatednode", nominated="n1") == 1000000.0


# ------------------------------------------------------------------------------------------------ common.FeasibleNodesForJob (actions/common/feasible_nodes_test.go:20-277)
def test_feasible_nodes_for_job():
    """a job whose every pod needs some kind of GPU looks only at nodes with idle or releasing GPUs — whole devices, room on a shared device, MIG instances; any other
    job looks at all nodes.  The seven nodes of the reference's test: CPU only, idle GPU, releasing GPU, idle fraction, releasing fraction, idle MIG, releasing MIG."""
    import ctypes as C
    MIG = "nvidia.com/mig-1g.10gb"
    nodes = {"cpu-node": {"GPUs": 0}, "idle-gpu-node": {"GPUs": 1}, "releasing-gpu-node": {"GPUs": 1}, "idle-fraction-node": {"GPUs": 1}, "releasing-fraction-node": {"GPUs": 1},
             "idle-mig-node": {"GPUs": 0, "MigStrategy": "mixed", "MigInstances": {MIG: 1}}, "releasing-mig-node": {"GPUs": 0, "MigStrategy": "mixed", "MigInstances": {MIG: 1}}}
    J = lambda name, tasks, **kw: {"Name": name, "Priority": 50, "QueueName": "q", "Tasks": tasks, **kw}
    P = {"State": "Pending"}
    jobs = [J("hold-gpu", [{"State": "Releasing", "NodeName": "releasing-gpu-node"}], RequiredGPUsPerTask=1),
            J("half-a", [{"State": "Running", "NodeName": "idle-fraction-node", "GPUGroups": ["0"]}], RequiredGPUsPerTask=0.5),
            J("half-b", [{"State": "Running", "NodeName": "releasing-fraction-node", "GPUGroups": ["0"]}], RequiredGPUsPerTask=0.5),
            J("half-c", [{"State": "Releasing", "NodeName": "releasing-fraction-node", "GPUGroups": ["0"]}], RequiredGPUsPerTask=0.5),
            J("hold-mig", [{"State": "Releasing", "NodeName": "releasing-mig-node", "RequiredMigInstances": {MIG: 1}}]),
            J("cpu only", [P]), J("whole gpu", [P], RequiredGPUsPerTask=1), J("distributed whole gpu", [P, P], RequiredGPUsPerTask=1),
            J("mixed", [{"State": "Pending", "RequiredGPUs": 1}, {"State": "Pending", "RequiredGPUs": 0}], RequiredGPUsPerTask=1),
            J("fraction", [P], RequiredGPUsPerTask=0.5), J("gpu memory", [P], RequiredGpuMemory=50), J("mig", [{"State": "Pending", "RequiredMigInstances": {MIG: 1}}])]
    case = {"Name": "feasible", "Nodes": nodes, "Queues": [{"Name": "q", "DeservedGPUs": 4}], "Jobs": jobs, "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case, fractions=True)
    lib = T.Oracle.lib(); lib.kai_oracle_feasible_nodes.restype = C.c_int
    s = snap.as_struct()
    gpu_nodes = set(nodes) - {"cpu-node"}
    for job, want in (("cpu only", set(nodes)), ("whole gpu", gpu_nodes), ("distributed whole gpu", gpu_nodes), ("mixed", set(nodes)), ("fraction", gpu_nodes), ("gpu memory", gpu_nodes), ("mig", gpu_nodes)):
        out = np.zeros(snap.n_nodes, np.uint8)
        n = lib.kai_oracle_feasible_nodes(C.byref(cfg), C.byref(s), snap.job_names.index(job), out.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert {snap.node_names[i] for i in range(snap.n_nodes) if out[i]} == want and n == len(want), job


# ------------------------------------------------------------------------------------------------ GetTasksToEvict (api/podgroup_info/eviction_info_test.go:16-106)
EVICT = [  # (name, pod-sets (name, minAvailable), running pods by pod-set, tasks expected, has more)
    ("WithoutSubGroups_EvictOne", [("default", 1)], ["default"] * 3, 1, True),
    ("WithoutSubGroups_EmptyQueue", [("default", 1)], [], 0, False),
    ("WithoutSubGroups_MultipleEvict", [("default", 2)], ["default"] * 2, 2, False),
    ("WithSubGroups_SingleEvict", [("default", 2), ("sg1", 1), ("sg2", 1)], ["sg1", "sg1", "sg2"], 1, True),
    ("WithSubGroups_EvictAll", [("default", 2), ("sg1", 1), ("sg2", 1)], ["sg1", "sg2"], 2, False),
]


@pytest.mark.parametrize("name,podsets,pods,want,more", EVICT, ids=[c[0] for c in EVICT])
def test_tasks_to_evict(name, podsets, pods, want, more):
    """one elastic pod above the gang's minimum at a time, the whole gang once it is down to its minimum — per pod-set, and whether anything is left afterwards"""
    import ctypes as C
    root = {"Name": "", "PodSets": [{"Name": n, "MinAvailable": m, "TopologyConstraint": None} for n, m in podsets], "SubGroups": [], "TopologyConstraint": None}
    case = {"Name": name, "Nodes": {"n1": {"GPUs": 8}}, "Queues": [{"Name": "q", "DeservedGPUs": 8}],
            "Jobs": [{"Name": "pg1", "Priority": 50, "QueueName": "q", "RequiredGPUsPerTask": 1, "RootSubGroupSet": root,
                      "Tasks": [{"State": "Running", "NodeName": "n1", **({"SubGroupName": p} if p != "default" else {})} for p in pods]}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    lib = T.Oracle.lib(); lib.kai_oracle_tasks_to_evict.restype = C.c_int
    out = np.zeros(8, np.int32); hm = C.c_int(0); s = snap.as_struct()
    n = lib.kai_oracle_tasks_to_evict(C.byref(cfg), C.byref(s), 0, out.ctypes.data_as(C.POINTER(C.c_int32)), 8, C.byref(hm))
    assert (n, bool(hm.value)) == (want, more)


# ------------------------------------------------------------------------------------------------ GetTasksToAllocate / getNumTasksToAllocate (api/podgroup_info/allocation_info_test.go)
TO_ALLOCATE = T.load_golden("kat_tasks_to_allocate")["cases"]  # tools/go_kat_tasks_to_allocate.py: Test_GetTasksToAllocate :62-217 (8), Test_getNumTasksToAllocate :396-461 (5)


@pytest.mark.parametrize("case", TO_ALLOCATE, ids=[f"{c['line']}:{c['fn']}:{c['name']}".replace(" ", "_").replace(",", "") for c in TO_ALLOCATE])
def test_tasks_to_allocate_reference_cases(case):
    """GetTasksToAllocate (allocation_info.go:27-54): pod-sets in PodSetOrderFn order, of each the gang's missing tasks up to minAvailable — in one chunk — or, once every pod-set has its
    gang, one elastic task of the first pod-set that has one; getNumTasksToAllocate (:145-177) is the chunk's size.  The test's order functions (pod-sets by name, tasks by UID) order its
    cases like the default plugins do (its names rise with creation order)."""
    import ctypes as C
    if case["fn"] == "GetTasksToAllocate":
        sets = sorted(case["minAvailable"]); tasks = case["tasks"]
    else:
        sets = ["default"]; tasks = [{"name": f"task{i}", "subGroup": "default", "status": st} for i, st in enumerate(case["statuses"])]
    root = {"Name": "", "PodSets": [{"Name": n, "MinAvailable": case["minAvailable"][n] if case["fn"] == "GetTasksToAllocate" else case["minAvailable"], "TopologyConstraint": None} for n in sets], "SubGroups": [], "TopologyConstraint": None}
    tc = {"Name": case["name"], "Nodes": {"n1": {"GPUs": 8}}, "Queues": [{"Name": "q", "DeservedGPUs": 8}],
          "Jobs": [{"Name": "pg", "Priority": 50, "QueueName": "q", "RequiredGPUsPerTask": 1, "RootSubGroupSet": root,
                    "Tasks": [{"State": t["status"], "SubGroupName": t["subGroup"], **({"NodeName": "n1"} if t["status"] != "Pending" else {})} for t in tasks]}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(tc)
    lib = T.Oracle.lib(); lib.kai_oracle_tasks_to_allocate.restype = C.c_int
    s = snap.as_struct(); out = np.full(8, -1, np.int32)
    n = lib.kai_oracle_tasks_to_allocate(C.byref(cfg), C.byref(s), 0, 1 if case.get("realAllocation", True) else 0, out.ctypes.data_as(C.POINTER(C.c_int32)), 8)
    if case["fn"] == "GetTasksToAllocate":
        assert n == case["wantNumTasks"] and [tasks[int(p)]["name"] for p in out[:n]] == case["wantTasks"]   # (pod index = position in the job's task list)
    else:
        assert n == case["want"]
