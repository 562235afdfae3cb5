"""The oracle's Statement against framework/statement_checkpoint_test.go (TestStatement_Checkpoint :30-215): a checkpoint, one or two operations on a running and on
a pending task of a two-GPU node, Rollback — jobs and nodes must be what they were; an Allocate of a task the statement has just evicted is an error."""
import ctypes as C

import numpy as np
import pytest

import kai_testlib as T

CP, EVICT, ALLOCATE, PIPELINE, ROLLBACK = 0, 1, 2, 3, 4
RUN, PEND = "running_job0-0", "pending_job0-0"
CASES = [  # (name, operations (op, task, updateTaskIfExistsOnNode), expected results of the calls)
    ("rollback evict", [(EVICT, RUN, 0)], [1]),
    ("rollback allocate", [(ALLOCATE, PEND, 0)], [1]),
    ("rollback pipeline updateIfNeeded true", [(PIPELINE, PEND, 1)], [1]),
    ("rollback pipeline updateIfNeeded false", [(PIPELINE, PEND, 0)], [1]),
    ("rollback allocate evict", [(ALLOCATE, PEND, 0), (EVICT, PEND, 0)], [1, 1]),
    ("rollback pipeline evict", [(PIPELINE, PEND, 1), (EVICT, PEND, 0)], [1, 1]),
    ("rollback evict pipeline", [(EVICT, RUN, 0), (PIPELINE, RUN, 1)], [1, 1]),
    ("rollback pipeline evict update false", [(PIPELINE, PEND, 0), (EVICT, PEND, 0)], [1, 1]),
    ("rollback evict pipeline update false", [(EVICT, RUN, 0), (PIPELINE, RUN, 0)], [1, 1]),
    ("rollback illegal evict allocate", [(EVICT, RUN, 0), (ALLOCATE, RUN, 0)], [1, 0]),
]


def session():
    case = {"Name": "statement", "Nodes": {"node0": {"GPUs": 2}}, "Queues": [{"Name": "queue0", "DeservedGPUs": 2}],
            "Jobs": [{"Name": "running_job0", "RequiredGPUsPerTask": 1, "QueueName": "queue0", "Priority": 50, "Tasks": [{"State": "Running", "NodeName": "node0"}]},
                     {"Name": "pending_job0", "RequiredGPUsPerTask": 1, "QueueName": "queue0", "Priority": 50, "Tasks": [{"State": "Pending"}]}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case)
    cfg.plugins = 0  # the test's session has no plugins: no event handlers, no predicates
    return snap, cfg


def run_script(snap, cfg, rows):
    lib = T.Oracle.lib(); lib.kai_oracle_statement_script.restype = C.c_int
    s = snap.as_struct(); sc = np.array(rows, np.int32).reshape(-1, 4) if rows else np.zeros((0, 4), np.int32)
    res = np.zeros(max(len(sc), 1), np.int32); st = np.zeros(snap.n_pods, np.int32); nd = np.zeros(snap.n_pods, np.int32); nodes = (T.abi.KaiNodeState * max(snap.n_nodes, 1))()
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    assert lib.kai_oracle_statement_script(C.byref(cfg), C.byref(s), ip(sc), len(sc), ip(res), ip(st), ip(nd), nodes) == 0
    return res[: len(sc)].tolist(), st.tolist(), nd.tolist(), T.nodes_to_np(nodes, snap.n_nodes, snap.n_res)


@pytest.mark.parametrize("name,ops,want", CASES, ids=[c[0].replace(" ", "_") for c in CASES])
def test_statement_checkpoint_rollback(name, ops, want):
    snap, cfg = session()
    _, st0, nd0, nodes0 = run_script(snap, cfg, [])
    rows = [(CP, 0, 0, 0)] + [(op, snap.pod_names.index(task), 0, flag) for op, task, flag in ops] + [(ROLLBACK, 0, 0, 0)]
    res, st, nd, nodes = run_script(snap, cfg, rows)
    assert res[1:-1] == want
    assert st == st0 and nd == nd0
    for k in nodes0: assert np.array_equal(nodes[k], nodes0[k]), k


def test_statement_operations_change_the_state_before_the_rollback():
    """(so that the equality above is not vacuous) an eviction moves the pod's GPU to Releasing, an allocation takes an idle one"""
    snap, cfg = session()
    _, st, nd, nodes = run_script(snap, cfg, [(EVICT, snap.pod_names.index(RUN), 0, 0), (ALLOCATE, snap.pod_names.index(PEND), 0, 0)])
    S = T.abi.POD_STATUS
    assert st[snap.pod_names.index(RUN)] == S["Releasing"] and st[snap.pod_names.index(PEND)] == S["Allocated"] and nd[snap.pod_names.index(PEND)] == 0
    assert nodes["releasing"][0, T.abi.RES_GPU] == 1 and nodes["idle"][0, T.abi.RES_GPU] == 0 and nodes["used"][0, T.abi.RES_GPU] == 2


def test_evicting_a_pending_task_is_an_error():
    """statement_test.go:156-210 (TestStatement_Evict "Pending job eviction") and :307-360: a task without a node cannot be evicted ("node doesn't exist in session")"""
    snap, cfg = session()
    _, st0, nd0, nodes0 = run_script(snap, cfg, [])
    res, st, nd, nodes = run_script(snap, cfg, [(EVICT, snap.pod_names.index(PEND), 0, 0)])
    assert res == [0] and st == st0 and nd == nd0 and all(np.array_equal(nodes[k], nodes0[k]) for k in nodes0)


def test_pipeline_unpipeline_on_a_full_node():
    """statement_test.go:462-560 (TestStatement_Pipeline_Unpipeline "basic pipeline/unpipeline"): a pending task pipelined onto a node whose two GPUs are in use, then
    undone: the job holds no GPU again and the node's used GPUs are the two of the running job"""
    case = {"Name": "statement", "Nodes": {"node0": {"GPUs": 2}}, "Queues": [{"Name": "queue0", "DeservedGPUs": 2}],
            "Jobs": [{"Name": "pending_job0", "RequiredGPUsPerTask": 1, "QueueName": "queue0", "Priority": 50, "Tasks": [{"State": "Pending"}]},
                     {"Name": "running_job0", "RequiredGPUsPerTask": 1, "QueueName": "queue0", "Priority": 50, "Tasks": [{"State": "Running", "NodeName": "node0"}] * 2}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(case); cfg.plugins = 0
    p = snap.pod_names.index(PEND)
    res, st, nd, nodes = run_script(snap, cfg, [(PIPELINE, p, 0, 1)])
    assert res == [1] and st[p] == T.abi.POD_STATUS["Pipelined"] and nd[p] == 0 and nodes["used"][0, T.abi.RES_GPU] == 3
    res, st, nd, nodes = run_script(snap, cfg, [(CP, 0, 0, 0), (PIPELINE, p, 0, 1), (ROLLBACK, 0, 0, 0)])
    assert st[p] == T.abi.POD_STATUS["Pending"] and nd[p] == -1 and nodes["used"][0, T.abi.RES_GPU] == 2
