#!/usr/bin/env python3
"""The reference's OWN benchmark topologies as workloads of this path.

Shapes restated from the generators of pkg/scheduler/actions/benchmark_test.go:199-473 (createBenchmarkTopology, ...WithRunningJobs, ...WithMixedJobs,
...WithManyQueues, ...WithGangJobs) and integration_tests/reclaim/reclaim_benchmark_test.go:62-160 (buildReclaimTopology), as TestTopologyBasic dictionaries that
tests/kai_testlib.case_to_snapshot turns into snapshots the same way it does for the golden tables.  Per benchmark: the oracle (checker), the host-compiled engine (one
CPU thread) and — on a box with a GPU — the device through the C ABI: `kai_session_open` + the benchmark's actions per iteration, which is what one `b.N` iteration of the Go
benchmark times minus its fixture construction (`BuildSession`: fake clientset + informers, ~100 ms floor).  The device's operations must equal the oracle's.

    python tools/ref_benchmarks.py [--max-nodes 1000] [--no-oracle-above 500] [--out profiles/r04_reference_benchmarks.json]

The published numbers beside each line are the reference's (Intel Core Ultra 7 165H, docs/developer/designs/vectorizing-resources/README.md:359-391): other hardware, a
Go program including fixture construction — context, not a measured ratio."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

TRAIN = 50  # constants.PriorityTrainNumber


def _nodes(n, fmt="node-%d"):
    return {fmt % i: {"GPUs": 8} for i in range(n)}


def _four_queues(total):
    q = total / 4
    return ([{"Name": f"queue-{i}", "ParentQueue": "dept-a" if i < 2 else "dept-b", "DeservedGPUs": q, "GPUOverQuotaWeight": 1} for i in range(4)],
            [{"Name": "dept-a", "DeservedGPUs": total / 2}, {"Name": "dept-b", "DeservedGPUs": total / 2}])


def _job(i, queue, tasks):
    return {"Name": f"job-{i}", "RequiredGPUsPerTask": 1, "Priority": TRAIN, "QueueName": queue, "Tasks": tasks}


def topology(n_nodes, n_jobs, kind="pending", n_queues=4, tasks_per_job=1):
    """createBenchmarkTopology (:199-245) / WithRunningJobs (:248-297) / WithMixedJobs (:300-356) / WithManyQueues (:359-421) / WithGangJobs (:424-473)"""
    jobs = []
    for i in range(n_jobs):
        if kind == "pending":
            tasks = [{"State": "Pending"}]
        elif kind == "running":
            tasks = [{"State": "Running", "NodeName": "node-%d" % (i % n_nodes)}]
        elif kind == "mixed":
            tasks = [{"State": "Running", "NodeName": "node-%d" % (i % n_nodes)}] if i % 2 == 0 else [{"State": "Pending"}]
        elif kind == "gang":
            tasks = [{"State": "Pending"} for _ in range(tasks_per_job)]
        jobs.append(_job(i, "queue-%d" % (i % n_queues), tasks))
    total = float(n_nodes * 8)
    if n_queues == 4:
        queues, depts = _four_queues(total)
    else:
        n_depts = max(1, (n_queues + 3) // 4)
        queues = [{"Name": f"queue-{i}", "ParentQueue": f"dept-{i % n_depts}", "DeservedGPUs": total / n_queues, "GPUOverQuotaWeight": 1} for i in range(n_queues)]
        depts = [{"Name": f"dept-{i}", "DeservedGPUs": total / n_depts} for i in range(n_depts)]
    return {"Name": f"benchmark-topology-{kind}", "Nodes": _nodes(n_nodes), "Jobs": jobs, "Queues": queues, "Departments": depts,
            "Mocks": {"CacheRequirements": {"NumberOfCacheBinds": n_jobs * 2, "NumberOfCacheEvictions": n_jobs, "NumberOfPipelineActions": n_jobs * 2}}}


def reclaim_large(n_nodes):
    """buildReclaimTopology with benchmarkReclaimLargeJobs' parameters (reclaim_benchmark_test.go:62-160): 8 running one-GPU jobs per node in a queue that deserves
    nothing, one pending job of n/10 tasks x 8 GPUs in a queue that deserves the cluster."""
    jobs = [{"Name": f"running-job-{i}", "RequiredGPUsPerTask": 1, "Priority": TRAIN, "QueueName": "queue-0", "Tasks": [{"NodeName": "node%d" % (i % n_nodes), "State": "Running"}]}
            for i in range(n_nodes * 8)]
    jobs.append({"Name": "very-large-job", "RequiredGPUsPerTask": 8, "Priority": TRAIN, "QueueName": "queue-1", "Tasks": [{"State": "Pending"} for _ in range(n_nodes // 10)]})
    return {"Name": "very large job reclaim benchmark", "Nodes": _nodes(n_nodes, "node%d"), "Jobs": jobs,
            "Queues": [{"Name": "queue-0", "DeservedGPUs": 0.0, "GPUOverQuotaWeight": 0}, {"Name": "queue-1", "DeservedGPUs": float(n_nodes * 8), "GPUOverQuotaWeight": 0}],
            "Mocks": {"CacheRequirements": {"NumberOfCacheBinds": n_nodes * 4, "NumberOfCacheEvictions": n_nodes * 4, "NumberOfPipelineActions": n_nodes * 4}}}


# (name in the reference, builder, actions, the reference's published figure)
ALL4 = ("allocate", "consolidation", "reclaim", "preempt")
BENCHES = [
    ("BenchmarkAllocateAction_SmallCluster", lambda: topology(10, 50), ("allocate",), "106.6 ms/op"),
    ("BenchmarkAllocateAction_MediumCluster", lambda: topology(50, 200), ("allocate",), "127.1 ms/op"),
    ("BenchmarkAllocateAction_LargeCluster", lambda: topology(100, 500), ("allocate",), "183.3 ms/op"),
    ("BenchmarkReclaimAction_SmallCluster", lambda: topology(10, 50, "running"), ("reclaim",), "102.7 ms/op"),
    ("BenchmarkReclaimAction_MediumCluster", lambda: topology(50, 200, "running"), ("reclaim",), "105.0 ms/op"),
    ("BenchmarkPreemptAction_SmallCluster", lambda: topology(10, 50, "mixed"), ("preempt",), "104.7 ms/op"),
    ("BenchmarkPreemptAction_MediumCluster", lambda: topology(50, 200, "mixed"), ("preempt",), "110.5 ms/op"),
    ("BenchmarkConsolidationAction_SmallCluster", lambda: topology(10, 50, "mixed"), ("consolidation",), "111.4 ms/op"),
    ("BenchmarkConsolidationAction_MediumCluster", lambda: topology(50, 200, "mixed"), ("consolidation",), "187.5 ms/op"),
    ("BenchmarkFullSchedulingCycle_LargeCluster", lambda: topology(100, 500, "mixed"), ALL4, None),
    ("BenchmarkManyQueues_MediumCluster", lambda: topology(50, 200, "pending", n_queues=20), ("allocate",), None),
    ("BenchmarkGangScheduling_MediumCluster", lambda: topology(50, 100, "gang", tasks_per_job=4), ("allocate",), None),
] + [("BenchmarkReclaimLargeJobs_%dNode" % n, (lambda n=n: reclaim_large(n)), ("reclaim",), pub)
     for n, pub in ((10, "104.4 ms/op"), (50, "130.2 ms/op"), (100, "241.2 ms/op"), (200, "816.0 ms/op"), (500, "8.97 s/op"), (1000, "did not complete within 40 min"))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-nodes", type=int, default=1000)
    ap.add_argument("--no-oracle-above", type=int, default=1000, help="skip the oracle (faithful O(N)-per-decision restatement) above that many nodes; the host-compiled engine then is the checker")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import kai_testlib as T
    from test_engine_hostsim import HostSim
    try:
        import torch
        gpu = torch.cuda.is_available()
    except Exception:
        gpu = False
    rows = []
    for name, build, actions, published in BENCHES:
        case = build()
        if len(case["Nodes"]) > args.max_nodes:
            continue
        snap, cfg, _meta = T.case_to_snapshot(case, actions)
        row = {"benchmark": name, "nodes": snap.n_nodes, "pods": snap.n_pods, "jobs": snap.n_jobs, "actions": list(actions), "reference_published": published}
        ref = None
        if snap.n_nodes <= args.no_oracle_above:
            ref = T.Oracle.run(snap, cfg, actions)
            row["oracle_ms"] = ref.elapsed_ms
        twin = HostSim.run(snap, cfg, actions)
        row["host_compiled_engine_ms"] = twin.elapsed_ms
        want = ref.ops if ref is not None else [tuple(o) for o in twin.ops]
        if ref is not None:
            assert [tuple(o) for o in twin.ops] == ref.ops, name
        row["operations"] = len(want); row["evictions"] = sum(1 for o in want if o[0] == 2)
        if gpu:
            times = []
            with T.pkg.KaiCore(cfg) as core:
                for it in range(args.iters + 1):
                    t0 = time.perf_counter()
                    ssn = core.open_session(snap)
                    got = []
                    for a in actions:
                        got += [(int(o["kind"]), int(o["pod"]), int(o["node"]), int(o["job"])) for o in ssn.execute(a)]
                    dt = (time.perf_counter() - t0) * 1e3
                    ssn.close()
                    if it:
                        times.append(dt)
                    assert got == want, f"{name}: the device's operations differ"
                    if dt > 3000 and it >= 1:  # seconds per iteration: two are enough
                        break
            times.sort()
            row["mi355x_open_plus_actions_ms"] = times[len(times) // 2]
            row["equal_to_" + ("oracle" if ref is not None else "host_compiled_engine")] = True
        rows.append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        json.dump({"source": "pkg/scheduler/actions/benchmark_test.go:30-473, integration_tests/reclaim/reclaim_benchmark_test.go (shapes restated by tools/ref_benchmarks.py)",
                   "note": "reference_published: the reference's own figures on an Intel Core Ultra 7 165H incl. ~100 ms of fixture construction per op (BASELINE.md section 1) — other hardware, context only",
                   "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
