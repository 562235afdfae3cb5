#!/usr/bin/env python3
"""Known answers of the reference's reclaim strategies → tests/golden/kat_reclaim_strategies.json.

Source: pkg/scheduler/plugins/proportion/reclaimable/strategies/strategies_test.go — three tables of Go composite literals (MaintainFairShareStrategy :23-188,
the same with several resources :190-550, GuaranteeDeservedQuotaStrategy :552-806), parsed with the literal parser of tools/go_fixtures.py.  Only the reference
is read; this script and the JSON are committed.  Shares as in kat_reclaimable.json: {res: [Deserved, FairShare, MaxAllowed, Allocated, AllocatedNotPreemptible]}.
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import go_fixtures as G  # noqa: E402

SRC = "/root/reference/pkg/scheduler/plugins/proportion/reclaimable/strategies/strategies_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_reclaim_strategies.json")
G.CONSTS["resource_info.MinMemory"] = 10.0 * 1024 * 1024  # api/resource_info/base_resources.go:18


def share(d):
    d = d or {}
    return [float(d.get(k, 0)) for k in ("Deserved", "FairShare", "MaxAllowed", "Allocated", "AllocatedNotPreemptible")]


def queue(q):
    qrs = (q or {}).get("QueueResourceShare") or {}
    return {"cpu": share(qrs.get("CPU")), "memory": share(qrs.get("Memory")), "gpu": share(qrs.get("GPU"))}


def main():
    src = open(SRC).read()
    cases = []
    contexts = [m for m in re.finditer(r'Context\("([^"]+)"', src)]
    for ci, m in enumerate(contexts):
        end = contexts[ci + 1].start() if ci + 1 < len(contexts) else len(src)
        body_start = src.index("}{", m.end()) + 1
        p = G.Parser(src, body_start)
        node = p.parse_composite({"map": ("string", None)})
        strategy = "guarantee_deserved_quota" if "Guarantee" in m.group(1) else "maintain_fair_share"
        for k, v in node["_map"]:
            name = G.ev(k); line = src.count("\n", 0, src.index('"' + name + '"', body_start, end)) + 1
            val = G._j(G.ev(v))
            reclaimer, reclaimee = queue(val.get("reclaimerQueue")), queue(val.get("reclaimeeQueue"))
            rem = val.get("remainingResourceShare")
            if rem is None:  # the specs pass reclaimeeQueue.GetAllocatedShare() (:182, :801)
                remaining = [reclaimee["cpu"][3], reclaimee["memory"][3], reclaimee["gpu"][3]]
            else:
                remaining = [float(rem.get("rs.CpuResource", 0)), float(rem.get("rs.MemoryResource", 0)), float(rem.get("rs.GpuResource", 0))]
            rr = val.get("reclaimerResources")
            required = [float(x) for x in rr["args"]] if rr else [0.0, 0.0, 0.0]
            cases.append({"name": name, "line": line, "context": m.group(1), "strategy": strategy, "reclaimer": reclaimer, "reclaimee": reclaimee,
                          "required": required, "remaining": remaining, "want": bool(val["expected"])})
    json.dump({"source": "plugins/proportion/reclaimable/strategies/strategies_test.go", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases →", OUT)
    for c in cases: print(c["line"], c["strategy"], c["name"][:70], c["remaining"], c["want"])


if __name__ == "__main__":
    main()
