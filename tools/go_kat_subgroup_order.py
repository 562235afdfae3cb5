#!/usr/bin/env python3
"""Known answers of the sub-group order plugin → tests/golden/kat_subgroup_order.json.

Source: pkg/scheduler/plugins/subgrouporder/subgroup_order_test.go TestSubGroupOrderFn :33-102 — six pairs of pod-sets (minAvailable, number of allocated tasks) and PodSetOrderFn's
verdict (subgroup_order.go:31-62; lPrioritized = -1, rPrioritized = 1, equalPrioritization = 0).  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/plugins/subgrouporder/subgroup_order_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_subgroup_order.json")
WANT = {"lPrioritized": -1, "rPrioritized": 1, "equalPrioritization": 0}


def main():
    src = open(SRC).read()
    at = src.index("func TestSubGroupOrderFn")
    start = src.index("}{", at) + 1; end = match(src, start)
    cases = []
    for m in re.finditer(r'\{\s*name:\s*"([^"]*)",\s*lMinAvailable:\s*(\d+),\s*lAllocated:\s*(\d+),\s*rMinAvailable:\s*(\d+),\s*rAllocated:\s*(\d+),\s*want:\s*(\w+),\s*\}', src[start:end]):
        cases.append({"name": m.group(1), "line": line_of(src, start + m.start()), "lMinAvailable": int(m.group(2)), "lAllocated": int(m.group(3)), "rMinAvailable": int(m.group(4)), "rAllocated": int(m.group(5)), "want": WANT[m.group(6)]})
    json.dump({"source": "plugins/subgrouporder/subgroup_order_test.go TestSubGroupOrderFn", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases: print(c)


if __name__ == "__main__":
    main()
