#!/usr/bin/env python3
"""Known answers of the reference's queue-capacity checks → tests/golden/kat_capacity_policy.json.

Source: plugins/proportion/capacity_policy/max_allowed_check_test.go :38-208 (isOverLimit) and quota_check_test.go :32-130
(isAllocatedNonPreemptibleOverQuota): tables of Go composite literals over ONE queue, parsed with the literal parser of tools/go_fixtures.py.
Quantities as [cpu, memory, gpu].  Only the reference is read; this script and the JSON are committed."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import go_fixtures as G  # noqa: E402

DIR = "/root/reference/pkg/scheduler/plugins/proportion/capacity_policy/"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_capacity_policy.json")


def q3(d):
    d = d or {}
    return [float(d.get("rs.CpuResource", 0)), float(d.get("rs.MemoryResource", 0)), float(d.get("rs.GpuResource", 0))]


def table(path, marker):
    src = open(path).read()
    start = src.index("}{", src.index(marker)) + 1
    node = G.Parser(src, start).parse_composite({"map": ("string", None)})
    for k, v in node["_map"]:
        name = G.ev(k)
        yield name, src.count("\n", 0, src.index('"' + name + '"', start)) + 1, G._j(G.ev(v))


def main():
    cases = []
    for name, line, v in table(DIR + "max_allowed_check_test.go", 'Context("IsOverMaxAllowed tests"'):
        cases.append({"fn": "isOverLimit", "file": "max_allowed_check_test.go", "line": line, "name": name, "limit": q3(v.get("maxAllowed")), "allocated": q3(v.get("allocated")),
                      "requested": q3(v.get("requestedQuota")), "want": bool(v["isOverMaxAllowed"])})
    for name, line, v in table(DIR + "quota_check_test.go", 'Context("isAllocatedNonPreemptibleOverQuota tests"'):
        cases.append({"fn": "isAllocatedNonPreemptibleOverQuota", "file": "quota_check_test.go", "line": line, "name": name, "limit": q3(v.get("deserved")),
                      "allocated": q3(v.get("allocatedNonPreemptible")), "requested": q3(v.get("requestedQuota")), "want": bool(v["expectedResult"])})
    json.dump({"source": "plugins/proportion/capacity_policy/{max_allowed_check,quota_check}_test.go", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases →", OUT)
    for c in cases: print(c["file"], c["line"], c["name"][:50], c["limit"], c["allocated"], c["requested"], c["want"])


if __name__ == "__main__":
    main()
