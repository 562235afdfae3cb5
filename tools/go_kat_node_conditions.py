#!/usr/bin/env python3
"""Known answers of the node-condition predicate → tests/golden/kat_node_conditions.json.

Source: pkg/scheduler/scheduler_util/scheduler_utils_test.go :23-150 — nine single-case tests of ValidateIsNodeReady (= CheckNodeConditionPredicate, scheduler_utils.go:12-46): a node
built by createTestNode from a list of conditions (type, status), optionally marked unschedulable, and whether the test expects it to be ready (`if !ValidateIsNodeReady` = expects ready).
The k8s constants are written out as the strings they stand for (k8s.io/api/core/v1: NodeReady = "Ready", ConditionTrue = "True", …).  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/scheduler_util/scheduler_utils_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_node_conditions.json")
V1 = {"v1.NodeReady": "Ready", "v1.NodeMemoryPressure": "MemoryPressure", "v1.NodeDiskPressure": "DiskPressure", "v1.NodePIDPressure": "PIDPressure", "v1.NodeNetworkUnavailable": "NetworkUnavailable",
      "v1.ConditionTrue": "True", "v1.ConditionFalse": "False", "v1.ConditionUnknown": "Unknown"}


def const(tok):
    tok = tok.strip()
    return tok[1:-1] if tok.startswith('"') else V1[tok]


def main():
    src = open(SRC).read()
    cases = []
    for m in re.finditer(r"func (Test\w+)\(t \*testing\.T\) \{", src):
        lo = m.end() - 1; hi = match(src, lo); body = src[lo:hi]
        if "createTestNode(" not in body:
            continue
        conds = [[const(a), const(b)] for a, b in re.findall(r"Type:\s*([\w.\"]+),\s*Status:\s*([\w.\"]+),", body)]
        want_ready = re.search(r"if (!?)ValidateIsNodeReady\(node\)", body).group(1) == "!"
        cases.append({"test": m.group(1), "line": line_of(src, m.start()), "conditions": conds, "unschedulable": "node.Spec.Unschedulable = true" in body, "ready": want_ready})
    json.dump({"source": "scheduler_util/scheduler_utils_test.go", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases:
        print(c["line"], c["test"], c["conditions"], c["unschedulable"], c["ready"])


if __name__ == "__main__":
    main()
