"""The oracle's JobsOrderByQueues against the reference's own unit tests of that type (actions/utils/job_order_by_queue_test.go, SURVEY.md section 8c
"ordering"): the tables there build a session by hand (priority + elastic job order, proportion's queue order where the test opens the plugin) and pop.
Transcribed by hand — the Go literals are object graphs, not TestTopologyBasic tables; every case cites its line.  Creation timestamps are zero there
unless a test sets them, so the queue order falls through to the UID (session_plugins.go:283-299)."""
import ctypes as C

import numpy as np
import pytest

import kai_testlib as T

abi = T.abi
VICTIM, NON_PENDING, UNREADY = 1, 2, 4


def build(queues, jobs, proportion=False, queue_created=None, node=None):
    """queues: [(name, parent)], jobs: [(name, priority, queue, state, gpus)] → snapshot over no nodes (the tests have none).  `node`: GPUs of one node that
    holds the active pods — AcceptedResource is set when a pod is added to its node (node_info.go), the Go literals set it by hand."""
    case = {"Name": "job order", "DisableDefaultDepartment": True, "Nodes": {"node0": {"GPUs": node}} if node else {},
            "Queues": [{"Name": n, "ParentQueue": p, "DeservedGPUs": 1, "DeservedCPUs": 1, "DeservedMemory": 1, "GPUOverQuotaWeight": 1} for n, p in queues],
            "Jobs": [{"Name": n, "Priority": pr, "QueueName": q, "RequiredGPUsPerTask": g, "Tasks": [{"State": st, **({"NodeName": "node0"} if node and st != "Pending" else {})}]} for n, pr, q, st, g in jobs],
            "JobExpectedResults": {}}
    snap, cfg, meta = T.case_to_snapshot(case)
    a = snap.arrays
    a["queue_created_ns"][:] = 0
    a["job_created_ns"][:] = 0
    for name, ns in (queue_created or {}).items(): a["queue_created_ns"][snap.queue_names.index(name)] = ns
    cfg.plugins = abi.PLUGINS["priority"] | abi.PLUGINS["elastic"] | (abi.PLUGINS["proportion"] if proportion else 0)
    return snap, cfg


def pops(snap, cfg, flags, script, init=None):
    lib = T.Oracle.lib()
    lib.kai_oracle_jobs_order.restype = C.c_int
    out = (C.c_int32 * 64)(); ln = C.c_int(0)
    mask = None
    if init is not None:
        m = np.zeros(snap.n_jobs, np.uint8)
        for n in init: m[snap.job_names.index(n)] = 1
        mask = m.ctypes.data_as(C.POINTER(C.c_uint8))
    sc = np.array([(-1 if s is None else snap.job_names.index(s)) for s in script], np.int32)
    s = snap.as_struct()
    n = lib.kai_oracle_jobs_order(C.byref(cfg), C.byref(s), flags, 0, mask, sc.ctypes.data_as(C.POINTER(C.c_int32)), len(sc), out, 64, C.byref(ln))
    assert n >= 0, n
    return [snap.job_names[out[i]] if out[i] >= 0 else None for i in range(n)], ln.value


POP = None


def test_numerical_priority_within_same_queue():  # :42-144
    snap, cfg = build([("q1", "pq1"), ("pq1", "")], [("p150", 150, "q1", "Pending", 0), ("p255", 255, "q1", "Pending", 0), ("p160", 160, "q1", "Pending", 0), ("p200", 200, "q1", "Pending", 0)])
    got, n = pops(snap, cfg, NON_PENDING | UNREADY, [POP] * 5)
    assert n == 4 and got == ["p255", "p200", "p160", "p150", None]


@pytest.mark.parametrize("node_counts", (False, True))
def test_victim_queue_pop_next_job(node_counts):  # :146-334 — two queues over their quota by the same amount, q2 a second older: the victims alternate, lowest priority first
    jobs = [(f"q{q}j{j}", 101 - j, f"q{q}", "Allocated", 1) for q in (1, 2) for j in (1, 2, 3)]
    snap, cfg = build([("q1", "pq1"), ("q2", "pq1"), ("pq1", "")], jobs, proportion=True, queue_created={"q1": 10**9, "pq1": 10**9, "q2": 0}, node=6)
    if not node_counts: snap.arrays["node_flags"][0] |= abi.NODE_NOT_READY  # the reference's session has no nodes: the cluster total is zero (proportion.go:263-264 skips such a node)
    got, n = pops(snap, cfg, VICTIM | UNREADY, [POP] * 6)
    assert got == ["q1j3", "q2j3", "q1j2", "q2j2", "q1j1", "q2j1"]


PUSH_CASES = [  # TestJobsOrderByQueues_PushJob :336-646: (jobs already in, the job pushed, expected pops)
    ([], ("p150", 150), ["p150"]),
    ([("p140", 150)], ("p150", 150), ["p140", "p150"]),  # same priority: the UID decides ("1" < "2" there; the names order the same way here)
    ([("p150", 150)], ("p160", 160), ["p160", "p150"]),
]


@pytest.mark.parametrize("inside,pushed,want", PUSH_CASES)
def test_push_job(inside, pushed, want):
    jobs = [(n, p, "q1", "Pending", 0) for n, p in inside + [pushed]]
    snap, cfg = build([("q1", "pq1"), ("pq1", "")], jobs)
    got, n = pops(snap, cfg, NON_PENDING | UNREADY, [pushed[0]] + [POP] * len(want), init=[n for n, _ in inside])
    assert n == len(inside) and got == want


def test_requeue_job():  # :650-745 — pop, push the same job back, pop
    snap, cfg = build([("q1", "pq1"), ("pq1", "")], [("p150", 150, "q1", "Pending", 0)])
    got, _ = pops(snap, cfg, NON_PENDING | UNREADY, [POP, "p150", POP, POP])
    assert got == ["p150", "p150", None]


def test_orphan_queue_jobs_are_skipped():  # :747-796 — a queue whose parent does not exist takes no part
    snap, cfg = build([("orphan-queue", "missing-parent")], [("orphan-job", 100, "orphan-queue", "Pending", 0)])
    got, n = pops(snap, cfg, NON_PENDING, [POP])
    assert n == 0 and got == [None]


NLEVEL = [  # TestNLevelQueueHierarchy :798-978: (queues, jobs, pushed instead of initialized, expected order)
    ("three level hierarchy", [("root", ""), ("dept1", "root"), ("dept2", "root"), ("team1", "dept1"), ("team2", "dept1"), ("team3", "dept2")],
     [("job1-team1-p100", 100, "team1"), ("job2-team2-p200", 200, "team2"), ("job3-team3-p150", 150, "team3"), ("job4-team1-p250", 250, "team1")], False,
     ["job4-team1-p250", "job1-team1-p100", "job2-team2-p200", "job3-team3-p150"]),
    ("four level hierarchy", [("org", ""), ("div1", "org"), ("dept1", "div1"), ("team1", "dept1")], [("deep-job", 100, "team1")], False, ["deep-job"]),
    ("single level hierarchy", [("default", "")], [("job1-default-p100", 100, "default"), ("job2-default-p200", 200, "default")], False, ["job2-default-p200", "job1-default-p100"]),
    ("two level hierarchy", [("root", ""), ("leaf1", "root"), ("leaf2", "root")], [("job1-leaf1-p100", 100, "leaf1"), ("job2-leaf2-p200", 200, "leaf2")], False,
     ["job1-leaf1-p100", "job2-leaf2-p200"]),
    ("mixed depth hierarchy", [("root", ""), ("leaf1", "root"), ("dept", "root"), ("team", "dept")], [("job1-shallow-p150", 150, "leaf1"), ("job2-deep-p200", 200, "team")], False,
     ["job2-deep-p200", "job1-shallow-p150"]),
    ("multiple root queues", [("root1", ""), ("leaf1", "root1"), ("root2", ""), ("leaf2", "root2")], [("job1-root1-p100", 100, "leaf1"), ("job2-root2-p200", 200, "leaf2")], False,
     ["job1-root1-p100", "job2-root2-p200"]),
    ("multiple single level root queues", [("queue-a", ""), ("queue-b", ""), ("queue-c", "")],
     [("job-a-p100", 100, "queue-a"), ("job-b-p300", 300, "queue-b"), ("job-c-p200", 200, "queue-c")], False, ["job-a-p100", "job-b-p300", "job-c-p200"]),
    ("push job builds n-level tree", [("root", ""), ("dept", "root"), ("team", "dept")], [("job1-p100", 100, "team"), ("job2-p200", 200, "team")], True, ["job2-p200", "job1-p100"]),
    ("push job to single level queue", [("default", "")], [("pushed-job", 100, "default")], True, ["pushed-job"]),
    ("tree cleanup after all jobs popped", [("root", ""), ("dept1", "root"), ("dept2", "root"), ("team1", "dept1"), ("team2", "dept2")],
     [("job1-team1", 200, "team1"), ("job2-team2", 100, "team2")], False, ["job1-team1", "job2-team2"]),
]


@pytest.mark.parametrize("name,queues,jobs,push,want", NLEVEL, ids=[c[0].replace(" ", "_") for c in NLEVEL])
def test_n_level_queue_hierarchy(name, queues, jobs, push, want):
    snap, cfg = build(queues, [(n, p, q, "Pending", 0) for n, p, q in jobs])
    script = ([n for n, _, _ in jobs] if push else []) + [POP] * (len(want) + 1)
    got, n = pops(snap, cfg, NON_PENDING | UNREADY, script, init=[] if push else None)
    assert (len(want) == n or push) and got == want + [None]


def test_victim_queue_two_queues_with_running_jobs():  # :1047-1121 — two pops, then nil
    snap, cfg = build([("default", ""), ("team-0", "default"), ("team-1", "default")], [("job0", 100, "team-0", "Running", 0), ("job1", 100, "team-1", "Running", 0)])
    got, n = pops(snap, cfg, VICTIM, [POP] * 3)
    assert n == 2 and sorted(got[:2]) == ["job0", "job1"] and got[2] is None


READY = [(c["name"], [(ps["name"], ps["minAvailable"], ps["statuses"]) for ps in c["podSets"]], c["ready"], c["line"])  # api/podgroup_info/job_info_test.go:397-779 (TestPodGroupInfo_IsReadyForScheduling):
         for c in T.load_golden("kat_job_ready")["cases"]]                                                                   # pod-sets (name, minAvailable, pod states) → ready; tools/go_kat_job_ready.py


@pytest.mark.parametrize("name,podsets,ready,line", READY, ids=[f"{c[3]}:{c[0]}".replace(" ", "_").replace(",", "") for c in READY])
def test_is_ready_for_scheduling(name, podsets, ready, line):
    """a job enters a FilterUnready queue iff every pod-set has minAvailable alive tasks that are not scheduling-gated (job_info.go:399-406, podset.go:114-120)"""
    root = {"Name": "", "PodSets": [{"Name": n, "MinAvailable": m, "TopologyConstraint": None} for n, m, _ in podsets], "SubGroups": [], "TopologyConstraint": None}
    tasks = [{"State": st, **({"SubGroupName": n} if n != "default" else {}), **({"NodeName": "n1"} if st == "Running" else {})} for n, _, sts in podsets for st in sts]
    case = {"Name": name, "Nodes": {"n1": {"GPUs": 8}}, "Queues": [{"Name": "q", "DeservedGPUs": 8}],
            "Jobs": [{"Name": "job", "Priority": 50, "QueueName": "q", "RequiredGPUsPerTask": 1, "RootSubGroupSet": root, "Tasks": tasks}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins = abi.PLUGINS["priority"] | abi.PLUGINS["elastic"]
    got, n = pops(snap, cfg, UNREADY, [POP])
    assert (n == 1 and got == ["job"]) if ready else (n == 0 and got == [None])
