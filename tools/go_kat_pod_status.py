#!/usr/bin/env python3
"""Known answers of the pod-status classes → tests/golden/kat_pod_status.json.

Source: pkg/scheduler/api/pod_status/pod_status_test.go TestIsAliveStatus :11-81 — eleven statuses and whether IsAliveStatus holds (pod_status.go:59-71: the set the gang logic counts
as alive, PodSet.GetNumAliveTasks).  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/api/pod_status/pod_status_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_pod_status.json")


def main():
    src = open(SRC).read()
    at = src.index("func TestIsAliveStatus")
    start = src.index("}{", at) + 1; end = match(src, start)
    cases = []
    for m in re.finditer(r'\{\s*name:\s*"([^"]*)",\s*status:\s*(\w+),\s*expected:\s*(true|false),\s*\}', src[start:end]):
        cases.append({"fn": "IsAliveStatus", "line": line_of(src, start + m.start()), "name": m.group(1), "status": m.group(2), "expected": m.group(3) == "true"})
    json.dump({"source": "api/pod_status/pod_status_test.go TestIsAliveStatus", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases:
        print(c["line"], c["status"], c["expected"])


if __name__ == "__main__":
    main()
