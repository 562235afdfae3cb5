#!/usr/bin/env python3
"""Known answers of the reference's hierarchical fair-share division → tests/golden/kat_fair_share_tree.json.

Source: plugins/proportion/proportion_test.go — "Set fair share for multi hierarchy queues" (:43-263, a table of queue attribute literals) and the DescribeTable
"Set fair share for 2 hierarchy queues - simplified" (:265-524: getBaseQueues + per-entry overrides of deserved quota / over-quota weight / priority).  Both
run proportionPlugin.setFairShare (proportion.go:403-423: SetResourcesShare on the top queues, then on every queue's children with the parent's fair share as
the total) and compare the GPU fair share of every queue.  Parsed with the literal parser of tools/go_fixtures.py; only the reference is read.

A case: queues {name: {parent, priority, gpu: [Deserved, MaxAllowed, OverQuotaWeight, Request]}}, total [cpu, memory, gpu], want {name: gpu fair share}.
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import go_fixtures as G  # noqa: E402

SRC = "/root/reference/pkg/scheduler/plugins/proportion/proportion_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_fair_share_tree.json")


def q3(d):
    d = d or {}
    return [float(d.get("rs.CpuResource", 0)), float(d.get("rs.MemoryResource", 0)), float(d.get("rs.GpuResource", 0))]


def queue(q):
    g = ((q.get("QueueResourceShare") or {}).get("GPU")) or {}
    return {"parent": q.get("ParentQueue", "") or "", "priority": int(q.get("Priority", 0) or 0),
            "gpu": [float(g.get("Deserved", 0)), float(g.get("MaxAllowed", 0)), float(g.get("OverQuotaWeight", 0)), float(g.get("Request", 0))]}


def main():
    src = open(SRC).read()
    cases = []
    c1 = src.index('Context("Set fair share for multi hierarchy queues"')
    start = src.index("}{", c1) + 1
    node = G.Parser(src, start).parse_composite({"map": ("string", None)})
    for k, v in node["_map"]:
        name = G.ev(k); val = G._j(G.ev(v))
        cases.append({"name": name, "line": src.count("\n", 0, src.index('"' + name + '"', start)) + 1,
                      "queues": {qn: queue(q) for qn, q in val["queues"].items()}, "total": q3(val["totalResources"]),
                      "want": {k2: float(v2) for k2, v2 in val["expectedFairShare"].items()}})
    c2 = src.index('Context("Set fair share for 2 hierarchy queues - simplified"')
    end = src.index('Context("Get Node Resources"')
    base = {"d1": ("", ), "q1": ("d1", ), "d2": ("", ), "q2": ("d2", )}  # getBaseQueues :266-345: Deserved 2, OverQuotaWeight 1, Request 100, MaxAllowed unlimited
    for m in re.finditer(r'Entry\("([^"]+)",\s*testData\{', src[c2:end]):
        val = G._j(G.ev(G.Parser(src, c2 + m.end() - 1).parse_composite("testData")))
        queues = {n: {"parent": p[0], "priority": 0, "gpu": [2.0, -1.0, 1.0, 100.0]} for n, p in base.items()}
        for n, d in (val.get("deservedOverride") or {}).items(): queues[n]["gpu"][0] = float(d)
        for n, w in (val.get("weightOverride") or {}).items(): queues[n]["gpu"][2] = float(w)
        for n, pr in (val.get("priorityOverride") or {}).items(): queues[n]["priority"] = int(pr)
        cases.append({"name": m.group(1), "line": src.count("\n", 0, c2 + m.start()) + 1, "queues": queues, "total": q3(val["totalResources"]),
                      "want": {k2: float(v2) for k2, v2 in val["expectedFairShare"].items()}})
    json.dump({"source": "plugins/proportion/proportion_test.go", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases →", OUT)
    for c in cases: print(c["line"], c["name"][:60], c["total"], c["want"])


if __name__ == "__main__":
    main()
