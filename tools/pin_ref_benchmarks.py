#!/usr/bin/env python3
"""Pins for the reference's own benchmark shapes that bench.py runs beside its headline (`other_shapes.reference_benchmarks`).

For each shape of tools/ref_benchmarks.py named in SHAPES: the ORACLE's committed operations (SHA-256 as tests/kai_testlib.ops_sha256 hashes them), their count, the
evictions among them, and the host-compiled engine's time on one core of this container — written to profiles/reference_benchmark_pins.json.  bench.py rebuilds the same
snapshot on the GPU box, runs `kai_session_open` + the benchmark's actions through the C ABI and compares its operations' hash with the pin (no oracle on that path).

    python tools/pin_ref_benchmarks.py            # ~1 min: the 500-node reclaim takes the oracle 17 s
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))

# the shapes bench.py runs, by their names in the reference (actions/benchmark_test.go, integration_tests/reclaim/reclaim_benchmark_test.go)
SHAPES = ("BenchmarkPreemptAction_MediumCluster", "BenchmarkConsolidationAction_MediumCluster", "BenchmarkFullSchedulingCycle_LargeCluster",
          "BenchmarkReclaimLargeJobs_200Node", "BenchmarkReclaimLargeJobs_500Node")


def main():
    import kai_testlib as T
    import ref_benchmarks as RB
    from test_engine_hostsim import HostSim
    out = {}
    for name, build, actions, published in RB.BENCHES:
        if name not in SHAPES:
            continue
        snap, cfg, _ = T.case_to_snapshot(build(), actions)
        ref = T.Oracle.run(snap, cfg, actions)
        t0 = time.perf_counter(); tw = HostSim.run(snap, cfg, actions); twin_ms = (time.perf_counter() - t0) * 1e3
        assert [tuple(o) for o in tw.ops] == ref.ops, name
        out[name] = {"actions": list(actions), "nodes": snap.n_nodes, "pods": snap.n_pods, "jobs": snap.n_jobs, "ops": len(ref.ops), "evictions": sum(1 for o in ref.ops if o[0] == 2),
                     "ops_sha256": T.ops_sha256(ref.ops), "oracle_ms": ref.elapsed_ms, "host_compiled_engine_ms": twin_ms, "reference_published": published}
        print(name, out[name], flush=True)
    with open(os.path.join(ROOT, "profiles", "reference_benchmark_pins.json"), "w") as f:
        json.dump({"source": "tools/pin_ref_benchmarks.py: the oracle's operations on the shapes of tools/ref_benchmarks.py (pkg/scheduler/actions/benchmark_test.go:199-473, "
                             "integration_tests/reclaim/reclaim_benchmark_test.go:62-160); reference_published = the reference's own figure on an Intel Core Ultra 7 165H incl. "
                             "~100 ms of fixture construction per op (BASELINE.md section 1): other hardware, context only",
                   "pins": out}, f, indent=1)


if __name__ == "__main__":
    main()
