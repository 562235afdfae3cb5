#!/usr/bin/env python3
"""Replay a reference-schema snapshot on the MI355X core — the counterpart of the reference's cmd/snapshot-tool (main.go:38-116): load
snapshot.zip / snapshot.json, open a session, run the configured actions in order, print the per-action duration
(metrics.UpdateActionDuration there) and write the decisions (BindRequests / evictions, one batch) as JSON.

    python tools/kai_replay.py snapshot.zip [--out decisions.json] [--check]     # --check: compare with the CPU oracle (test infrastructure)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kai_testlib as T  # noqa: E402  (loads the package from its in-tree path)

pkg = T.pkg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("snapshot"); ap.add_argument("--out"); ap.add_argument("--check", action="store_true"); ap.add_argument("--scheduler-name")
    a = ap.parse_args()
    t0 = time.perf_counter()
    got = pkg.ingest.ingest_file(a.snapshot, scheduler_name=a.scheduler_name)
    s = got.snapshot
    print(f"ingest: {s.n_nodes} nodes, {s.n_pods} pods, {s.n_jobs} pod groups, {s.n_queues} queues, {s.n_pod_classes}x{s.n_node_classes} predicate classes, "
          f"resources {got.resource_names} in {(time.perf_counter() - t0) * 1e3:.1f} ms")
    for w in got.warnings:
        print("  note:", w)
    ops = []
    with pkg.KaiCore(got.config) as core:
        ssn = core.open_session(s)
        for act in got.actions:
            t = time.perf_counter()
            o = ssn.execute(act)
            st = ssn.stats()
            print(f"action {act}: {len(o)} operations, {st.decisions} decisions, {(time.perf_counter() - t) * 1e3:.2f} ms (kernel {st.kernel_ms:.2f} ms)")
            ops += [(int(x["kind"]), int(x["pod"]), int(x["node"]), int(x["job"])) for x in o]
        ssn.close()
    doc = got.decisions_json(ops)
    d = json.loads(doc)
    print(f"decisions: {len(d['bindRequests'])} bind requests, {len(d['evictions'])} evictions, {len(d['pipelined'])} pipelined")
    if a.out:
        open(a.out, "w").write(doc)
    if a.check:
        ref = T.Oracle.run(s, got.config, tuple(got.actions))
        print("oracle:", "identical" if ref.ops == ops else f"DIFFERENT ({len(ref.ops)} vs {len(ops)} operations)")
        return 0 if ref.ops == ops else 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
