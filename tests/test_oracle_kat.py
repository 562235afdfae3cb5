"""Known-answer tests of the reference's own unit tests, hand-transcribed into tests/golden/kat_*.json (source file and line in
each entry): the oracle's pure functions must return exactly (==, float64) what the Go tests expect."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import kai_testlib as T


def _load(name):
    with open(os.path.join(T.GOLDEN, name)) as f:
        return json.load(f)


DIV = _load("kat_resource_division.json")
NP = _load("kat_nodeplacement.json")


def _lib():
    lib = T.Oracle.lib()
    lib.kai_oracle_spread_score.restype = C.c_double
    lib.kai_oracle_spread_score.argtypes = [C.c_double, C.c_double]
    lib.kai_oracle_divide_one.restype = C.c_int
    return lib


@pytest.mark.parametrize("case", DIV["cases"], ids=[f"L{c['line']}" for c in DIV["cases"]])
def test_resource_division_kat(case):
    lib = _lib()
    qs = case["queues"]; Q = len(qs)
    arr = lambda k: np.array([q[k] for q in qs], np.float64)
    deserved, limit, oqw, request, fair = arr("deserved"), arr("max_allowed"), arr("oqw"), arr("request"), arr("fair")
    prio = np.array([q["priority"] for q in qs], np.int32); created = np.array([q["created"] for q in qs], np.int64)
    out = np.zeros(Q, np.float64); rem = C.c_double(0)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = lib.kai_oracle_divide_one(0 if case["fn"] == "setResourceShare" else 1, Q, C.c_double(case["total"]), C.c_double(case["k_value"]),
                                   p(deserved, C.c_double), p(limit, C.c_double), p(oqw, C.c_double), p(request, C.c_double), None, p(fair, C.c_double),
                                   p(prio, C.c_int), p(created, C.c_int64), p(out, C.c_double), C.byref(rem))
    assert rc == 0
    assert rem.value == case["remaining"], (case["name"], rem.value)
    for got, want in zip(out.tolist(), case["fair"]):  # (null: the reference's It does not look at that queue)
        assert want is None or got == float(want), (case["name"], out.tolist())


@pytest.mark.parametrize("case", NP["pack"], ids=[f"L{c['line']}" for c in NP["pack"]])
def test_nodepack_kat(case):
    lib = _lib()
    # getMinMaxPerNode (plugins/nodeplacement/pack.go:66-86): nodes without the resource are skipped, min starts at MaxFloat64, max at 0
    lo, hi = np.finfo(np.float64).max, 0.0
    for n in case["nodes"]:
        if n["alloc"] == 0:
            continue
        lo, hi = min(lo, float(n["idle"])), max(hi, float(n["idle"]))
    for n in case["nodes"]:
        got = lib.kai_oracle_pack_score(lo, hi, float(n["idle"]), float(n["alloc"]))
        assert got == n["score"], (case["name"], n, got)


def test_nodespread_kat():
    lib = _lib()
    for c in NP["spread"]:
        assert lib.kai_oracle_spread_score(float(c["non_allocated"]), float(c["count"])) == c["score"], c


QO = _load("kat_queue_order.json")


@pytest.mark.parametrize("case", QO["cases"], ids=[f"L{c['line']}" for c in QO["cases"]])
def test_queue_order_kat(case):
    lib = _lib()
    lib.kai_oracle_queue_order.restype = C.c_int
    f = ("deserved", "fair", "max_allowed", "oqw", "allocated", "allocated_np", "request")
    shares = np.array([[[side[res][k] for k in f] for res in ("cpu", "memory", "gpu")] for side in (case["l"], case["r"])], np.float64)
    prio = np.array([case["l"]["priority"], case["r"]["priority"]], np.int32); created = np.array([1, 2], np.int64); total = np.zeros(3, np.float64)
    got = lib.kai_oracle_queue_order(shares.ctypes.data_as(C.POINTER(C.c_double)), prio.ctypes.data_as(C.POINTER(C.c_int)),
                                     created.ctypes.data_as(C.POINTER(C.c_int64)), total.ctypes.data_as(C.POINTER(C.c_double)))
    assert got == case["expected"], case["name"]


def test_minruntime_resolver_known_answers():
    """plugins/minruntime/resolver_test.go:38-212 (queue tree of createTestQueues :345-430, defaults 2 s / 1 s): preempt and reclaim min-runtime
    resolution — own value, inheritance, the queue method and the lowest-common-ancestor method incl. different top-level queues."""
    import ctypes as C
    lib = T.Oracle.lib(); f = lib.kai_oracle_min_runtime; f.restype = C.c_int64
    S = 1_000_000_000
    # dev, prod, research, dev-team1, dev-team2, prod-team1, prod-team2, research-project
    parent = np.array([-1, -1, -1, 0, 0, 1, 1, 2], np.int32)
    pre = np.array([5, 20, 4, -1, 3, -1, 15, 7], np.int64); rec = np.array([10, 30, 6, 8, -1, 25, 35, 9], np.int64)
    pre = np.where(pre >= 0, pre * S, -1); rec = np.where(rec >= 0, rec * S, -1)
    q = lambda pq, vq, kind: f(8, parent.ctypes.data_as(C.POINTER(C.c_int32)), pre.ctypes.data_as(C.POINTER(C.c_int64)), rec.ctypes.data_as(C.POINTER(C.c_int64)),
                               C.c_int64(2 * S), C.c_int64(1 * S), pq, vq, kind) // S
    assert [q(0, 6, 0), q(0, 5, 0), q(0, 3, 0), q(0, 1, 0), q(0, 2, 0)] == [15, 20, 5, 20, 4]     # getPreemptMinRuntime :38-77
    assert [q(3, 6, 1), q(3, 4, 1)] == [35, 10]                                                    # queue method :97-120
    assert [q(3, 6, 2), q(4, 3, 2), q(5, 6, 2), q(6, 6, 2), q(3, 7, 2)] == [30, 8, 35, 35, 6]      # LCA method :148-203


MR = _load("kat_minruntime.json")


@pytest.mark.parametrize("spec", MR["specs"], ids=[f"{c['fn']}:{c['line']}" for c in MR["specs"]])
def test_minruntime_filters_and_validators(spec):
    """plugins/minruntime/minruntime.go against the eleven specs of minruntime_test.go (tools/go_kat_minruntime.py): a non-elastic victim inside its min-runtime is
    filtered out of preemption / reclaim (resolved per queue: the victim's chain for preempt, the LCA or queue method for reclaim), a victim without a start time is
    not; an elastic victim passes the filter and the scenario validator keeps minAvailable of its pods"""
    import ctypes as C
    names = sorted(MR["queues"]); idx = {n: i for i, n in enumerate(names)}
    S = 1_000_000_000
    ns = lambda v: -1 if v is None else v * S
    parent = np.array([idx.get(MR["queues"][n]["parent"], -1) for n in names], np.int32)
    pre = np.array([ns(MR["queues"][n]["preempt_s"]) for n in names], np.int64); rec = np.array([ns(MR["queues"][n]["reclaim_s"]) for n in names], np.int64)
    lib = T.Oracle.lib(); f = lib.kai_oracle_minruntime_kat; f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int]
    mode = ("preemptFilterFn", "reclaimFilterFn", "preemptScenarioValidatorFn", "reclaimScenarioValidatorFn").index(spec["fn"])
    got = f(mode, len(names), parent.ctypes.data_as(C.POINTER(C.c_int32)), pre.ctypes.data_as(C.POINTER(C.c_int64)), rec.ctypes.data_as(C.POINTER(C.c_int64)),
            MR["default_preempt_s"] * S, MR["default_reclaim_s"] * S, 1 if spec["resolve_method"] == "queue" else 0, idx[spec["pending_queue"]], idx[spec["victim_queue"]],
            ns(spec["victim_started_ago_s"]), spec["victim_min_available"], spec["victim_pods"], spec["scenario_victim_tasks"])
    assert got == int(spec["want"]), (spec["name"], got)


# ------------------------------------------------------------------------------------------------ plugins/proportion/reclaimable
RECLAIMABLE = T.load_golden("kat_reclaimable")


@pytest.mark.parametrize("case", RECLAIMABLE["cases"], ids=[f"{c['mode']}:{c['line']}" for c in RECLAIMABLE["cases"]])
def test_reclaimable_known_answers(case):
    """plugins/proportion/reclaimable/reclaimable_test.go: CanReclaimResources (:34-531: the reclaimer's queue stays within its fair share, a non-preemptible
    reclaimer within its quota) and Reclaimable (:533-1163: strategies per reclaimee chunk, saturation ratios between siblings on every level, non-preemptible
    quota up the reclaimer's chain) on hand-set queue attributes — through kai_oracle_reclaimable, the same core the oracle's scenario validator calls."""
    import ctypes as C
    lib = T.Oracle.lib(); lib.kai_oracle_reclaimable.restype = C.c_int
    names = list(case["queues"])
    Q = len(names)
    parent = np.array([names.index(case["queues"][n][0]) if case["queues"][n][0] else -1 for n in names], np.int32)
    shares = np.array([[case["queues"][n][1][r] for r in ("cpu", "memory", "gpu")] for n in names], np.float64)  # Q x 3 x 5
    rq, req, preemptible = case["reclaimer"]
    req = np.array(req, np.float64)
    res_q = np.array([names.index(q) for q, _ in case["reclaimees"]], np.int32)
    res = np.array([r for _, r in case["reclaimees"]], np.float64).reshape(-1, 3)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)); ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    got = lib.kai_oracle_reclaimable(1 if case["mode"] == "can_reclaim" else 0, Q, ip(parent), dp(shares), names.index(rq), dp(req), int(preemptible),
                                     len(res_q), ip(res_q), dp(res), C.c_double(RECLAIMABLE["saturation_multiplier"]))
    assert got in (0, 1), got
    assert bool(got) == case["want"], f"{case['name']} (reclaimable_test.go:{case['line']})"


def _resource_share(rs, total=(0.0, 0.0, 0.0)):
    lib = T.Oracle.lib(); lib.kai_oracle_resource_share.restype = C.c_int
    a = np.array(rs, np.float64); t = np.array(total, np.float64); out = np.zeros(7)
    assert lib.kai_oracle_resource_share(a.ctypes.data_as(C.POINTER(C.c_double)), t.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    return out


def test_resource_share_known_answers():
    """plugins/proportion/resource_share: queue_resource_share_test.go:81-105 (requestable share 3 / 10 / 17, dominant share 2.5 of createQueueResourceShare :120-151)
    and resource_share_test.go:46-80 (requestable = min(MaxAllowed, Request), allocatable = min(MaxAllowed, max(Deserved, FairShare)), both with an unlimited MaxAllowed)."""
    q = [[1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11, 12, 13, 14], [15, 16, 17, 18, 19, 20, 21]]
    out = _resource_share(q)
    assert list(out[:3]) == [3.0, 10.0, 17.0] and out[6] == 2.5
    r = [21, 22, 10, 2, 5, 4, 17]  # createResourceShare
    out = _resource_share([r, r, r]); assert out[0] == 10.0 and out[3] == 10.0
    r[2] = -1.0
    out = _resource_share([r, r, r]); assert out[0] == 17.0 and out[3] == 22.0


QA = _load("kat_queue_attributes.json")


def test_queue_attributes_known_answers():
    """plugins/proportion/resource_share/queue_attributes_test.go (tools/go_kat_queue_attributes.py): GetRequestableShare per resource (12 cases), GetDominantResourceShare over
    a total capacity — what prioritizeSmallerResourceShare compares; a resource that is allocated but neither deserved nor a fair share counts 1000-fold, an unlimited
    deserved amount is measured against the cluster's total — (6, exact), GetAllocatableShare (5)"""
    row = lambda deserved=0, fair=0, max_allowed=0, allocated=0, request=0: [deserved, fair, max_allowed, 0, allocated, 0, request]
    for c in QA["requestable_share"]:
        rs = [row(), row(), row()]; rs[c["resource"]] = row(max_allowed=c["max_allowed"], request=c["request"])
        assert _resource_share(rs)[c["resource"]] == c["want"], c
    for c in QA["dominant_share"]:
        rs = [row(deserved=c["deserved"][r], fair=c["fair_share"][r], max_allowed=-1.0, allocated=c["allocated"][r]) for r in range(3)]
        assert _resource_share(rs, c["total"])[6] == c["want"], c
    for c in QA["allocatable_share"]:
        rs = [row(deserved=c["deserved"][r], fair=c["fair_share"][r], max_allowed=c["max_allowed"][r]) for r in range(3)]
        assert list(_resource_share(rs)[3:6]) == c["want"], c


# ------------------------------------------------------------------------------------------------ plugins/elastic, subgrouporder, taskorder
def _order_fn(which, l, r):
    lib = T.Oracle.lib(); lib.kai_oracle_order_fn.restype = C.c_int
    a = np.array(l, np.int32).reshape(-1); b = np.array(r, np.int32).reshape(-1)
    n = 2 if which == 0 else 1
    return lib.kai_oracle_order_fn(which, a.ctypes.data_as(C.POINTER(C.c_int32)), len(a) // n if which == 0 else 1, b.ctypes.data_as(C.POINTER(C.c_int32)), len(b) // n if which == 0 else 1)


ACTIVE_ALLOCATED = ("Allocated", "Binding", "Bound", "Running", "Pipelined")  # pod_status.IsActiveAllocatedStatus: a releasing pod is not
ELASTIC = [(c["line"], c["lMinAvailable"], c["rMinAvailable"], c["lPods"], c["rPods"], c["want"]) for c in T.load_golden("kat_elastic")["cases"]]  # plugins/elastic/elastic_test.go:17-533 (tools/go_kat_elastic.py): minAvailable of l / r, pod states of l / r, JobOrderFn


@pytest.mark.parametrize("line,lmin,rmin,lp,rp,want", ELASTIC, ids=[f"elastic:{c[0]}" for c in ELASTIC])
def test_elastic_job_order_known_answers(line, lmin, rmin, lp, rp, want):
    """elastic.JobOrderFn (plugins/elastic/elastic.go:25-65): below minAvailable first, then exactly at it, then above — by the active ALLOCATED pods of every pod-set"""
    n = lambda pods: sum(1 for s in pods if s in ACTIVE_ALLOCATED)
    assert _order_fn(0, [lmin, n(lp)], [rmin, n(rp)]) == want, f"elastic_test.go:{line}"


SUBGROUP_ORDER = [(c["lMinAvailable"], c["lAllocated"], c["rMinAvailable"], c["rAllocated"], c["want"]) for c in T.load_golden("kat_subgroup_order")["cases"]]  # subgroup_order_test.go:33-102 (tools/go_kat_subgroup_order.py)


@pytest.mark.parametrize("lmin,lalloc,rmin,ralloc,want", SUBGROUP_ORDER)
def test_subgroup_order_known_answers(lmin, lalloc, rmin, ralloc, want):
    """subgrouporder.PodSetOrderFn (subgroup_order.go:31-62): a pod-set below its minAvailable first, above it the smaller allocated / minAvailable ratio"""
    assert _order_fn(1, [lmin, lalloc], [rmin, ralloc]) == want


def test_task_order_known_answers():
    """taskorder.TaskOrderFn (task_order_test.go:17-53): the higher kai.scheduler/task-priority label first, a pod with the label before one without"""
    L = lambda p: [1, p]; NONE = [0, 0]
    assert _order_fn(2, L(1), L(2)) == 1 and _order_fn(2, L(2), L(2)) == 0 and _order_fn(2, L(2), L(1)) == -1
    assert _order_fn(2, NONE, L(1)) == 1 and _order_fn(2, L(1), NONE) == -1 and _order_fn(2, NONE, NONE) == 0


# ------------------------------------------------------------------------------------------------ scheduler_util/priority_queue.go
def _pq(max_size, ops):
    lib = T.Oracle.lib(); lib.kai_oracle_priority_queue.restype = C.c_int
    sc = np.array(ops, np.int32).reshape(-1); out = (C.c_int32 * 32)()
    n = lib.kai_oracle_priority_queue(max_size, sc.ctypes.data_as(C.POINTER(C.c_int32)), len(ops), out, 32)
    assert n >= 0, n
    return [out[i] for i in range(n)]


def test_priority_queue_known_answers():
    """scheduler_util/priority_queue_test.go: Push / Pop with and without a capacity (:13-108: a full queue drops an item — the one the heap holds at index
    maxQueueSize after the push), Peek (:110-186), Fix after the top item's key changed (:188-245), Empty / Len (:247-305)"""
    PUSH, POP, PEEK, FIX, LEN = 0, 1, 2, 3, 4
    assert _pq(0, [(PUSH, 2), (PUSH, 3), (PUSH, 1), (LEN, 0), (POP, 0)]) == [3, 1]
    assert _pq(0, [(PUSH, 1), (PUSH, 3), (PUSH, 2), (LEN, 0), (POP, 0)]) == [3, 1]
    assert _pq(2, [(PUSH, 2), (PUSH, 3), (PUSH, 4), (PUSH, 1), (LEN, 0), (POP, 0)]) == [2, 1]
    assert _pq(0, [(PUSH, 2), (PUSH, 3), (PEEK, 0), (LEN, 0)]) == [2, 2] and _pq(0, [(PEEK, 0)]) == [-1]
    assert _pq(0, [(PUSH, 2), (PUSH, 3), (FIX, 4), (PEEK, 0)]) == [3]
    assert _pq(0, [(LEN, 0)]) == [0] and _pq(0, [(PUSH, 5), (LEN, 0)]) == [1] and _pq(0, [(PUSH, 5), (PUSH, 1), (PUSH, 9), (LEN, 0)]) == [3]


STRATEGIES = T.load_golden("kat_reclaim_strategies")


@pytest.mark.parametrize("case", STRATEGIES["cases"], ids=[f"{c['strategy']}:{c['line']}" for c in STRATEGIES["cases"]])
def test_reclaim_strategies_known_answers(case):
    """plugins/proportion/reclaimable/strategies/strategies_test.go: MaintainFairShareStrategy (the reclaimee's remaining share is not within what it may hold,
    one resource and several, MaxAllowed below the fair share) and GuaranteeDeservedQuotaStrategy (the reclaimer stays within its quota, the reclaimee is over its own)."""
    lib = T.Oracle.lib(); lib.kai_oracle_reclaim_strategy.restype = C.c_int
    arr = lambda q: np.array([q[r] for r in ("cpu", "memory", "gpu")], np.float64)
    a, b = arr(case["reclaimer"]), arr(case["reclaimee"]); req = np.array(case["required"], np.float64); rem = np.array(case["remaining"], np.float64)
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    got = lib.kai_oracle_reclaim_strategy(0 if case["strategy"] == "maintain_fair_share" else 1, dp(a), dp(b), dp(req), dp(rem))
    assert got in (0, 1) and bool(got) == case["want"], f"{case['name']} (strategies_test.go:{case['line']})"


CAPACITY = T.load_golden("kat_capacity_policy")


@pytest.mark.parametrize("case", CAPACITY["cases"], ids=[f"{c['fn']}:{c['line']}" for c in CAPACITY["cases"]])
def test_capacity_policy_known_answers(case):
    """plugins/proportion/capacity_policy: isOverLimit (max_allowed_check_test.go:38-208 — MaxAllowed < Allocated + requested in a resource that is requested and
    limited) and isAllocatedNonPreemptibleOverQuota (quota_check_test.go:32-130 — the same against Deserved with the non-preemptible allocation)"""
    lib = T.Oracle.lib(); lib.kai_oracle_capacity_check.restype = C.c_int
    dp = lambda x: np.array(x, np.float64).ctypes.data_as(C.POINTER(C.c_double))
    lim, al, rq = (np.array(case[k], np.float64) for k in ("limit", "allocated", "requested"))
    got = lib.kai_oracle_capacity_check(0 if case["fn"] == "isOverLimit" else 1, lim.ctypes.data_as(C.POINTER(C.c_double)), al.ctypes.data_as(C.POINTER(C.c_double)), rq.ctypes.data_as(C.POINTER(C.c_double)))
    assert got in (0, 1) and bool(got) == case["want"], f"{case['name']} ({case['file']}:{case['line']})"


CAP_CHAIN = _load("kat_capacity_chain.json")["cases"]


@pytest.mark.parametrize("case", CAP_CHAIN, ids=[f"{c['fn']}:{c['line']}" for c in CAP_CHAIN])
def test_capacity_policy_over_a_queue_chain(case):
    """plugins/proportion/capacity_policy/capacity_policy.go against the fourteen cases of capacity_policy_test.go: the limit check and the non-preemptible quota check
    walked from the job's leaf queue up to the top (a violation at ANY level, the top one in the cases, makes the job unschedulable; a preemptible job skips the quota
    check) — tools/go_kat_capacity_chain.py"""
    names = sorted(case["queues"]); idx = {n: i for i, n in enumerate(names)}
    parent = [idx.get(case["queues"][n]["parent"], -1) for n in names]
    col = lambda f: np.array([[case["queues"][n]["shares"].get(r, {}).get(f, 0.0) for r in ("CPU", "Memory", "GPU")] for n in names], np.float64)
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ma, de, al, anp = col("MaxAllowed"), col("Deserved"), col("Allocated"), col("AllocatedNotPreemptible")
    rq = np.array(case["requested"], np.float64)
    mode = {"IsJobOverQueueCapacity": 0, "IsNonPreemptibleJobOverQuota": 1, "IsTaskAllocationOnNodeOverCapacity": 2}[case["fn"]]
    lib = T.Oracle.lib(); lib.kai_oracle_capacity_chain.restype = C.c_int
    got = lib.kai_oracle_capacity_chain(mode, len(names), (C.c_int32 * len(names))(*parent), d(ma), d(de), d(al), d(anp), idx[case["job_queue"]], int(case["preemptible"]), d(rq))
    assert got == int(case["want_schedulable"]), (case["name"], got)


GREEDY = [(c["requirements"], c["holders"], c["capacity"], c["want"]) for c in T.load_golden("kat_greedy_match")["cases"]]  # idle_gpus_test.go:106-199 (tools/go_kat_greedy_match.py): requirements, holders in order, capacity by holder, want


@pytest.mark.parametrize("req,holders,cap,want", GREEDY)
def test_greedy_match_requirements_known_answers(req, holders, cap, want):
    """accumulated_scenario_filters/idle_gpus/common.go:34-64 (the idle-GPU scenario filters' feasibility test) against idle_gpus_test.go:106-199"""
    lib = T.Oracle.lib(); lib.kai_oracle_greedy_match.restype = C.c_int
    names = sorted(cap)
    r = np.array(req, np.float64); h = np.array([names.index(x) for x in holders], np.int32); c = np.array([cap[n] for n in names] or [0.0], np.float64)
    got = lib.kai_oracle_greedy_match(r.ctypes.data_as(C.POINTER(C.c_double)), len(r), h.ctypes.data_as(C.POINTER(C.c_int32)), len(h), c.ctypes.data_as(C.POINTER(C.c_double)))
    assert bool(got) == want


MINIMAL_JOB = [(1 if c["fn"] == "UpdateRepresentative" else 0, c["representative"], c["job"], c["want"], c["line"]) for c in T.load_golden("kat_minimal_job")["cases"]]  # actions/common/minimal_job_comparison_test.go (tools/go_kat_minimal_job.py): mode, the representative's pods, the job's pods [pending, milli-CPU, memory, GPUs], want


@pytest.mark.parametrize("mode,rep,job,want,line", MINIMAL_JOB, ids=[f"minimal_job:{c[4]}" for c in MINIMAL_JOB])
def test_minimal_job_comparison_known_answers(mode, rep, job, want, line):
    """MinimalJobRepresentatives (actions/common/minimal_job_comparison.go:15-112): a job is skipped when a job of its signature with no larger sorted requests already failed"""
    lib = T.Oracle.lib(); lib.kai_oracle_minimal_job.restype = C.c_int
    a, b = np.array(rep, np.float64).reshape(-1, 4), np.array(job, np.float64).reshape(-1, 4)
    got = lib.kai_oracle_minimal_job(mode, a.ctypes.data_as(C.POINTER(C.c_double)), len(a), b.ctypes.data_as(C.POINTER(C.c_double)), len(b))
    assert bool(got) == want


FAIR_TREE = T.load_golden("kat_fair_share_tree")


@pytest.mark.parametrize("case", FAIR_TREE["cases"], ids=[f"fair_share_tree:{c['line']}" for c in FAIR_TREE["cases"]])
def test_fair_share_tree_known_answers(case):
    """proportion_test.go:43-524 — setFairShare over a queue hierarchy (proportion.go:403-423): the top queues divide the cluster, every queue's children divide the
    parent's fair share; deserved quota first, then over-quota by priority and weight.  GPU fair share of every queue, exact."""
    lib = T.Oracle.lib(); lib.kai_oracle_set_fair_share_tree.restype = C.c_int
    names = list(case["queues"]); Q = len(names)
    parent = np.array([names.index(case["queues"][n]["parent"]) if case["queues"][n]["parent"] else -1 for n in names], np.int32)
    z = np.zeros((3, Q)); des, lim, oqw, req = z.copy(), z.copy(), z.copy(), z.copy()  # CPU / Memory are zero-valued ResourceShare{} in the fixtures
    for i, n in enumerate(names): des[2, i], lim[2, i], oqw[2, i], req[2, i] = case["queues"][n]["gpu"]
    prio = np.array([case["queues"][n]["priority"] for n in names], np.int32); created = np.zeros(Q, np.int64)
    total = np.array(case["total"], np.float64); out = np.zeros((3, Q))
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    rc = lib.kai_oracle_set_fair_share_tree(Q, parent.ctypes.data_as(C.POINTER(C.c_int32)), dp(total), C.c_double(0.0), dp(des), dp(lim), dp(oqw), dp(req),
                                            prio.ctypes.data_as(C.POINTER(C.c_int)), created.ctypes.data_as(C.POINTER(C.c_int64)), dp(out))
    assert rc == 0
    got = {n: out[2, i] for i, n in enumerate(names)}
    assert got == case["want"], f"{case['name']} (proportion_test.go:{case['line']})"


def test_resource_quantities_comparisons_known_answers():
    """plugins/proportion/resource_share/resource_quantities_test.go:60-147: Less (strict in every resource), LessEqual, LessInAtLeastOneResource, and compareQuantities with
    the unlimited quantity (-1) above every number"""
    lib = T.Oracle.lib(); lib.kai_oracle_quantities_cmp.restype = C.c_int
    def cmp(which, a, b):
        x, y = np.array(a, np.float64), np.array(b, np.float64)
        return lib.kai_oracle_quantities_cmp(which, x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)))
    cpu, mem, gpu = 111.0, 22.0, 0.5; rq = [cpu, mem, gpu]
    assert cmp(0, rq, [cpu + 1, mem + 1, gpu + 0.1]) == 1 and cmp(0, rq, [cpu + 1, mem, gpu + 0.1]) == 0
    assert cmp(1, rq, [cpu + 1, mem + 1, gpu]) == 1 and cmp(1, rq, [cpu + 1, mem + 1, gpu - 0.1]) == 0
    assert cmp(2, rq, [cpu + 1, mem, gpu - 0.1]) == 1 and cmp(2, rq, [cpu, mem - 1, gpu - 0.1]) == 0
    U = -1.0
    for a, b, want in ((1.5, 2.5, -1), (2.5, 1.5, 1), (2.5, 2.5, 0), (U, 2.5, 1), (2.5, U, -1), (U, U, 0)):
        assert cmp(3, [a, 0, 0], [b, 0, 0]) - 1 == want


# ------------------------------------------------------------------------------------------------ AccumulatedIdleGpus (idle_gpus_test.go), tools/go_kat_idle_gpus.py
IG = _load("kat_idle_gpus.json")


def _idle_gpus_run(mode, idle, sorted_nodes, required=(), pend_state=(), rec_cache=(), pot_cache=(), pending=(), potential=(), recorded=(), first=False, node_mem=None):
    """names → small integers, one call of kai_oracle_idle_gpus_kat, the state afterwards back as names"""
    lib = T.Oracle.lib(); lib.kai_oracle_idle_gpus_kat.restype = C.c_int
    nodes = sorted(set(idle) | set(sorted_nodes) | {t["node"] for t in list(potential) + list(recorded) if t.get("node")})
    pods = sorted(set(pend_state) | set(rec_cache) | set(pot_cache) | {t["uid"] for t in list(pending) + list(potential) + list(recorded)})
    ni, pi = {n: i for i, n in enumerate(nodes)}, {p: i for i, p in enumerate(pods)}
    N = max(len(nodes), 1)
    idle_a = np.full(N, np.nan); [idle_a.__setitem__(ni[n], float(v)) for n, v in idle.items()]
    i32 = lambda xs: np.array(list(xs) or [0], np.int32); f64 = lambda xs: np.array(list(xs) or [0.0], np.float64)
    def accepted(t):  # a victim's AcceptedResource: its whole-device request, or its share of one device for a gpu-memory request (node_info.go setAcceptedResources)
        return float(t["gpus"]) if "gpu_memory_mib" not in t else t["gpu_memory_mib"] / float(node_mem[t["node"]])
    victims = list(potential) + list(recorded)
    a_sorted, a_req = i32(ni[n] for n in sorted_nodes), f64(required)
    a_ps, a_rc, a_pc = i32(pi[p] for p in pend_state), i32(pi[p] for p in rec_cache), i32(pi[p] for p in pot_cache)
    a_pid, a_pg = i32(pi[t["uid"]] for t in pending), f64(t["gpus"] for t in pending)
    a_vid, a_vn, a_vg = i32(pi[t["uid"]] for t in victims), i32(ni[t["node"]] if t.get("node") else -1 for t in victims), f64(accepted(t) for t in victims)
    idle_o = np.zeros(N); sorted_o = np.zeros(N + 8, np.int32); ps_o = np.zeros(len(pods) + 8, np.int32); rc_o = np.zeros(len(pods) + 8, np.int32); pc_o = np.zeros(len(pods) + 8, np.int32)
    n_s, n_ps, n_rc, n_pc, err = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    r = lib.kai_oracle_idle_gpus_kat(mode, len(nodes), p(idle_a, C.c_double), p(a_sorted, C.c_int32), len(sorted_nodes), p(a_req, C.c_double), len(required),
                                     p(a_ps, C.c_int32), len(pend_state), p(a_rc, C.c_int32), len(rec_cache), p(a_pc, C.c_int32), len(pot_cache),
                                     p(a_pid, C.c_int32), p(a_pg, C.c_double), len(pending), p(a_vid, C.c_int32), p(a_vn, C.c_int32), p(a_vg, C.c_double), len(potential), len(recorded), int(first),
                                     p(idle_o, C.c_double), p(sorted_o, C.c_int32), C.byref(n_s), p(ps_o, C.c_int32), C.byref(n_ps), p(rc_o, C.c_int32), C.byref(n_rc), p(pc_o, C.c_int32), C.byref(n_pc), C.byref(err))
    state = {"idle": {nodes[i]: float(idle_o[i]) for i in range(len(nodes)) if not np.isnan(idle_o[i])}, "sorted": [nodes[i] for i in sorted_o[:n_s.value]],
             "pending_in_state": sorted(pods[i] for i in ps_o[:n_ps.value]), "recorded_in_cache": sorted(pods[i] for i in rc_o[:n_rc.value]), "potential_in_cache": sorted(pods[i] for i in pc_o[:n_pc.value])}
    return r, bool(err.value), state, nodes


@pytest.mark.parametrize("case", IG["ordered_insert"], ids=[f"L{c['line']}" for c in IG["ordered_insert"]])
def test_idle_gpus_ordered_insert(case):
    """orderedInsert under cmp.Compare[string] = the filter's comparator (idle descending) with idle = minus the string's rank"""
    names = sorted(set(case["array"]) | {case["value"]})
    idle = {n: -float(i) for i, n in enumerate(names)}
    _, _, st, _ = _idle_gpus_run(0, idle, case["array"], potential=[{"uid": "v", "node": case["value"], "gpus": 0}], first=case["replace"])
    assert st["sorted"] == case["want"]


@pytest.mark.parametrize("case", IG["update_with_victim"], ids=[f"L{c['line']}" for c in IG["update_with_victim"]])
def test_idle_gpus_update_with_victim(case):
    r, _, st, nodes = _idle_gpus_run(1, case["idle"], case["sorted"], potential=[case["victim"]])
    assert nodes[r] == case["want_min_relevant"] and st["sorted"] == case["want_sorted"]


@pytest.mark.parametrize("case", IG["update_state"], ids=[f"L{c['line']}" for c in IG["update_state"]])
def test_idle_gpus_update_state_with_scenario(case):
    f, sc, w = case["fields"], case["scenario"], case["want"]
    r, _, st, _ = _idle_gpus_run(2, f["idle"], f["sorted"], f["required"], f["pending_in_state"], f["recorded_in_cache"], f["potential_in_cache"],
                                 sc["pending"], sc["potential_victims"], sc["recorded_victims"], first=case["first"])
    assert (r == 0) == w["err"], case["name"]
    for k in ("idle", "sorted", "pending_in_state", "recorded_in_cache", "potential_in_cache"):  # (the reference's test compares the fields its `want` names)
        if k in w:
            assert st[k] == w[k], (case["name"], k, st[k])


@pytest.mark.parametrize("case", IG["filter"], ids=[f"L{c['line']}" for c in IG["filter"]])
def test_idle_gpus_filter(case):
    f, sc, w = case["fields"], case["scenario"], case["want"]
    r, err, _, _ = _idle_gpus_run(3, f["idle"], f["sorted"], f["required"], f["pending_in_state"], f["recorded_in_cache"], f["potential_in_cache"],
                                  sc["pending"], sc["potential_victims"], sc["recorded_victims"], node_mem=sc.get("node_gpu_memory_mib"))
    assert err == w["err"] and bool(r) == w["valid"], case["name"]


# ------------------------------------------------------------------------------------------------ AccumulatedNodeAffinities in a session (oracle_solver.hpp)
def _with_predicate_classes(snap, seed):
    """random static predicate classes on a crowded snapshot: a few node classes, pod classes that fit some of them (class 0 fits every node)"""
    rng = np.random.default_rng(seed ^ 0xAFF1)
    a = snap.arrays; N, P = snap.n_nodes, snap.n_pods
    n_nc, n_pc = int(rng.integers(2, 5)), int(rng.integers(2, 5))
    a["node_class"] = rng.integers(0, n_nc, N).astype(np.int32)
    fit = (rng.random((n_pc, n_nc)) < 0.5).astype(np.uint8); fit[0, :] = 1
    a["class_fit"] = fit
    pc = np.zeros(P, np.int32)
    pending = a["pod_status"] == T.abi.POD_STATUS["Pending"]
    pc[pending] = rng.integers(0, n_pc, int(pending.sum()))  # running pods keep class 0: where they run already is not in question
    a["pod_class"] = pc
    return snap.finalize()


@pytest.mark.parametrize("seed", range(24))
def test_node_affinities_filter_prunes_without_changing_a_result(seed):
    """DESIGN.md §1 "not built": the AccumulatedNodeAffinities filter (node_affinities.go) is a necessary condition of a simulation's success, so a victim search
    with it and one without it commit the same operations.  The oracle runs the victim actions on crowded snapshots with static predicate classes both ways — the
    filter on the class table, oracle_solver.hpp — and must return the same operations, states and shares; over the seeds the filter must have dropped scenarios
    (it is exercised) and the filtered run must have simulated no more scenarios than the other."""
    synth = T.pkg.synth
    snap = synth.make_crowded_snapshot(6 + seed % 11, 9100 + seed, fill=0.85, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], hog_frac=0.5, elastic_frac=0.3 if seed % 2 else 0.0,
                                       nonpreempt_frac=0.1, n_pending_jobs=6 + seed % 9)
    _with_predicate_classes(snap, seed)
    cfg = T.abi.default_config(max_consolidation_preemptees=8); cfg.allow_consolidating_reclaim = seed % 2
    acts = (("reclaim", "preempt"), ("allocate", "consolidation", "reclaim", "preempt"), ("preempt", "reclaim"))[seed % 3]
    lib = T.Oracle.lib(); lib.kai_oracle_node_affinities_filter.restype = C.c_int64
    stats = lambda: (lambda v: (lib.kai_oracle_last_victim_stats(v), list(v))[1])((C.c_int64 * 3)())
    try:
        lib.kai_oracle_node_affinities_filter(0)
        plain = T.Oracle.run(snap, cfg, acts); st_plain = stats()
        lib.kai_oracle_node_affinities_filter(1)
        filtered = T.Oracle.run(snap, cfg, acts); st_filtered = stats()
    finally:
        dropped = lib.kai_oracle_node_affinities_filter(0)
    assert plain.ops == filtered.ops and (plain.pod_status == filtered.pod_status).all() and (plain.pod_node == filtered.pod_node).all()
    assert all(np.array_equal(plain.shares_final[k], filtered.shares_final[k]) for k in plain.shares_final)
    assert st_filtered[0] <= st_plain[0] <= st_filtered[0] + dropped, (st_plain, st_filtered, dropped)  # scenarios simulated: every one it saved is one it dropped
    test_node_affinities_filter_prunes_without_changing_a_result.dropped = getattr(test_node_affinities_filter_prunes_without_changing_a_result, "dropped", 0) + dropped


def test_node_affinities_filter_was_exercised():
    """(runs after the seeds above in file order; alone it has nothing to check)"""
    d = getattr(test_node_affinities_filter_prunes_without_changing_a_result, "dropped", None)
    if d is None:
        pytest.skip("the parametrized test did not run in this process")
    assert d > 0


# ------------------------------------------------------------------------------------------------ PodAccumulatedScenarioBuilder (pod_scenario_builder_test.go), tools/go_kat_scenario_builder.py
SB = _load("kat_scenario_builder.json")["specs"]


def _scenario_builder_case(spec):
    """initializeSession + createJobWithTasks of the test file (:287-395) as a scene: one node exactly full of the running jobs' one-GPU pods, a queue per job under
    "default", the pending reclaimer in team-a"""
    J, K = spec["jobs"], spec["tasks_per_job"]
    root = lambda m: {"Name": "", "TopologyConstraint": None, "SubGroups": [], "PodSets": [{"Name": "default", "MinAvailable": m, "TopologyConstraint": None}]}
    m = K if spec["min_available"] == "all" else spec["min_available"]
    jobs = [{"Name": f"job{j}", "Priority": 50, "QueueName": f"team-{j}", "RequiredCPUsPerTask": 0, "RootSubGroupSet": root(m),
             "Tasks": [{"State": "Running", "NodeName": "node-1", "RequiredGPUs": 1} for _ in range(K)]} for j in range(J)]
    jobs.append({"Name": "reclaimer", "Priority": 50, "QueueName": "team-a", "RequiredCPUsPerTask": 0, "RootSubGroupSet": root(1),
                 "Tasks": [{"State": "Pending", **({"RequiredGPUs": spec["reclaimer_gpus_per_task"]} if spec["reclaimer_gpus_per_task"] else {})} for _ in range(spec["reclaimer_tasks"])]})
    queues = [{"Name": f"team-{j}", "DeservedGPUs": 1} for j in range(J)] + [{"Name": "team-a", "DeservedGPUs": 1}]
    return {"Name": spec["name"], "Nodes": {"node-1": {"CPUMillis": 1000, "GPUs": J * K, "MaxTaskNum": 100}}, "Queues": queues, "Jobs": jobs, "JobExpectedResults": {}}


@pytest.mark.parametrize("spec", SB, ids=[f"{s['line']}" for s in SB])
def test_scenario_builder(spec):
    """actions/common/solvers/pod_scenario_builder.go against the nine specs of its Ginkgo suite: the scenarios GetValidScenario / GetNextScenario produce — how many,
    the potential victims of each (an elastic job gives up one pod at a time down to its minAvailable, then the rest; recorded victims are stepped over and the rest of
    their job is queued again), the recorded victim jobs of each, the job representatives of the last scenario's victims — with the AccumulatedIdleGpus filter
    deciding whether the scenario without victims counts."""
    snap, cfg, _ = T.case_to_snapshot(_scenario_builder_case(spec))
    a = snap.arrays
    pods_of = lambda j: [int(p) for p in np.nonzero(a["pod_job"] == snap.job_names.index(j))[0]]
    rec_job, rec_off, rec_pods = [], [0], []
    r = spec["recorded"]
    if r and "whole_jobs" in r:
        for j in range(r["whole_jobs"]):  # "indexes" of a Go map range: any two of the three alike jobs
            rec_job.append(snap.job_names.index(f"job{(0, 2)[j]}")); rec_off.append(len(rec_pods))
    elif r:
        rec_job.append(snap.job_names.index("job0")); rec_pods.append(pods_of("job0")[r["pod_of_first_job"]]); rec_off.append(len(rec_pods))
    i32 = lambda v: (C.c_int32 * max(len(v), 1))(*v)
    lib = T.Oracle.lib(); lib.kai_oracle_scenario_builder_kat.restype = C.c_int
    out = (C.c_int32 * 256)(); s = snap.as_struct()
    n = lib.kai_oracle_scenario_builder_kat(C.byref(cfg), C.byref(s), snap.job_names.index("reclaimer"), len(rec_job), i32(rec_job), i32(rec_off), i32(rec_pods), out, 256)
    assert n > 0, n
    S = out[0]; rows = [(out[1 + 2 * i], out[2 + 2 * i]) for i in range(S)]; K = out[1 + 2 * S]; sizes = list(out[2 + 2 * S:2 + 2 * S + K])
    w = spec["want"]
    if "first_scenario" in w: assert (S > 0) == w["first_scenario"], rows
    if w.get("next_of_empty_queue_is_nil"): assert S <= 1, rows
    if "scenarios" in w: assert S == w["scenarios"], rows
    if "potential_per_scenario" in w: assert [p for p, _ in rows] == w["potential_per_scenario"], rows
    if w.get("recorded_jobs_in_every_scenario"): assert all(rj == len(rec_job) for _, rj in rows), rows
    if "last_potential" in w: assert S > 0 and rows[-1][0] == w["last_potential"] == K and sizes == [w["last_representative_size"]] * K, (rows, sizes)


# ------------------------------------------------------------------------------------------------ the scenario objects (scenario/base_scenario_test.go, by_node_scenario_test.go), tools/go_kat_scenarios.py
SCN = _load("kat_scenarios.json")["cases"]


@pytest.mark.parametrize("case", SCN, ids=[f"{c['file'].split('_')[0]}:{c['line']}" for c in SCN])
def test_scenario_objects(case):
    """solvers/scenario/{base_scenario,by_node_scenario}.go against their four test tables: the potential victims and the task groups per job after the constructor and one
    AddPotentialVictimsTasks, the job representative a victim belongs to (one per call that added it: an elastic job's tasks sit in representatives of their own),
    the job of the latest potential victim, the victims a node's jobs bring along (all task groups of a job that has ANY potential victim there)."""
    by_job = {}
    for p in case["pods"]: by_job.setdefault(p["job"], []).append(p)
    jobs = [{"Name": j, "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 0, "RootSubGroupSet": {"Name": "", "TopologyConstraint": None, "SubGroups": [],
             "PodSets": [{"Name": "default", "MinAvailable": 1, "TopologyConstraint": None}]},
             "Tasks": [({"State": "Running", "NodeName": p["node"]} if p["node"] else {"State": "Pending"}) for p in ps]} for j, ps in sorted(by_job.items())]
    jobs.append({"Name": "123", "Priority": 50, "QueueName": "q", "RequiredCPUsPerTask": 0, "Tasks": [{"State": "Pending"}]})  # pendingTasksAsJob (the tests' one has no task; nothing asks about it)
    scene = {"Name": case["name"], "Nodes": {n: {"CPUMillis": 1000, "GPUs": 8, "MaxTaskNum": 100} for n in ("node1", "node2")}, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
             "Jobs": jobs, "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(scene)
    pod = {}
    for j, ps in by_job.items():
        for i, p in enumerate(ps): pod[(j, p["name"])] = snap.pod_names.index(f"{j}-{i}")
    name_of = {v: list(k) for k, v in pod.items()}
    P = lambda ids: [pod[tuple(x)] for x in ids]
    rec_job, rec_off, rec_pods = [], [0], []
    for rj in case["recorded_jobs"]:
        rec_job.append(snap.job_names.index(rj["name"])); rec_pods += P(rj["tasks"]); rec_off.append(len(rec_pods))
    mode = {"AddPotentialVictimsTasks": 0, "GetVictimJobRepresentativeById": 1, "LatestPotentialVictim": 2, "VictimsTasksFromNodes": 3}[case["func"].split("_")[-1]]
    arg = P([case["victim"]]) if mode == 1 else [snap.node_names.index(n) for n in case["node_names"]] if mode == 3 else []
    i32 = lambda v: (C.c_int32 * max(len(v), 1))(*v)
    lib = T.Oracle.lib(); lib.kai_oracle_scenario_kat.restype = C.c_int
    out = (C.c_int32 * 128)(); s = snap.as_struct()
    n = lib.kai_oracle_scenario_kat(C.byref(cfg), C.byref(s), snap.job_names.index("123"), i32(P(case["ctor_potential"])), len(case["ctor_potential"]), len(rec_job), i32(rec_job), i32(rec_off),
                                    i32(rec_pods), i32(P(case["added"])), len(case["added"]), mode, i32(arg), len(arg), out, 128)
    assert n > 0, n
    r = list(out[:n]); w = case["want"]
    if mode == 0:
        np_ = r[0]; assert [name_of[x] for x in r[1:1 + np_]] == w["potential"]
        g = r[1 + np_]; groups = {snap.job_names[r[2 + np_ + 2 * i]]: r[3 + np_ + 2 * i] for i in range(g)}
        assert groups == w["groups_per_job"], groups
    elif mode == 1:
        if w is None: assert r == [-1], r
        else: assert r[0] == len(w["tasks"]) and sorted(name_of[x] for x in r[1:]) == sorted(w["tasks"]), r
    elif mode == 2:
        assert (None if r[0] < 0 else snap.job_names[r[0]]) == (w and w["job"]), r
        if w: assert sorted([j, p["name"]] for j, ps in by_job.items() if j == w["job"] for p in ps) == sorted(w["tasks"])  # the session's whole job (getJobForTask), as the table spells it out
    else:
        assert sorted(name_of[x] for x in r[1:1 + r[0]]) == sorted(w), r  # (map order in the reference: compared as sets; the tables' cases have one job each)


PODSET = T.load_golden("kat_podset")


@pytest.mark.parametrize("case", PODSET["cases"], ids=[f"{c['line']}:{c['question']}:{c['name']}" for c in PODSET["cases"]])
def test_podset_gang_counters(case):
    """PodSet.AssignTask and the questions the gang logic asks of a pod-set (podset.go:56-149) on the sixteen cases of podset_test.go (tools/go_kat_podset.py): ready for scheduling,
    gang satisfied, elastic, and the active-allocated / active-used / alive / gated / pending counts by pod status (pod_status.go:25-71)."""
    import ctypes as C
    lib = T.Oracle.lib(); lib.kai_oracle_podset_kat.restype = C.c_int
    uid = np.asarray([int(u) for u, _ in case["pods"]] + [0], np.int32); st = np.asarray([T.abi.POD_STATUS[s] for _, s in case["pods"]] + [0], np.int32)
    out = np.zeros(8, np.int32)
    assert lib.kai_oracle_podset_kat(case["minAvailable"], uid.ctypes.data_as(C.POINTER(C.c_int32)), st.ctypes.data_as(C.POINTER(C.c_int32)), len(case["pods"]), out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    which = ("IsReadyForScheduling", "IsGangSatisfied", "IsElastic", "GetNumActiveAllocatedTasks", "GetNumActiveUsedTasks", "GetNumAliveTasks", "GetNumGatedTasks", "GetNumPendingTasks").index(case["question"])
    want = case["expected"]
    assert (bool(out[which]) == want) if isinstance(want, bool) else (int(out[which]) == want)


POD_STATUS_KAT = T.load_golden("kat_pod_status")


@pytest.mark.parametrize("case", POD_STATUS_KAT["cases"], ids=[f"{c['line']}:{c['status']}" for c in POD_STATUS_KAT["cases"]])
def test_alive_statuses(case):
    """IsAliveStatus (api/pod_status/pod_status.go:59-71) on the eleven statuses of TestIsAliveStatus (tools/go_kat_pod_status.py), through the pod-set counter that uses it
    (PodSet.GetNumAliveTasks, podset.go:56-77); the host mirror's status bit-set (abi.POD_STATUS) is the ABI's."""
    import ctypes as C
    lib = T.Oracle.lib(); lib.kai_oracle_podset_kat.restype = C.c_int
    uid = np.asarray([1, 0], np.int32); st = np.asarray([T.abi.POD_STATUS[case["status"]], 0], np.int32); out = np.zeros(8, np.int32)
    assert lib.kai_oracle_podset_kat(1, uid.ctypes.data_as(C.POINTER(C.c_int32)), st.ctypes.data_as(C.POINTER(C.c_int32)), 1, out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    assert bool(out[5]) == case["expected"]
