#!/usr/bin/env python3
"""Known answers of greedyMatchRequirements → tests/golden/kat_greedy_match.json.

Source: pkg/scheduler/actions/common/solvers/accumulated_scenario_filters/idle_gpus/idle_gpus_test.go Test_greedyMatchRequirements :106-199 — eight cases: the pending tasks' GPU
requirements (sorted descending by the caller), the holders (nodes) in the order they are tried, each holder's capacity, and whether every requirement finds a holder
(common.go:34-64: first fit with virtual allocation).  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402
from go_kat_level_order import top_fields  # noqa: E402

SRC = "/root/reference/pkg/scheduler/actions/common/solvers/accumulated_scenario_filters/idle_gpus/idle_gpus_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_greedy_match.json")


def main():
    src = open(SRC).read()
    at = src.index("func Test_greedyMatchRequirements")
    start = src.index("}{", src.index("want bool", at)) + 1; end = match(src, start)
    cases, i = [], start + 1
    while i < end:
        if src[i] == "{":
            j = match(src, i); f = top_fields(src, i, j)
            name = re.search(r'"([^"]*)"', src[f["name"][0]:f["name"][1]]).group(1)
            a0 = src.index("{", f["args"][0]); a = top_fields(src, a0, match(src, a0))
            body = lambda k: src[a[k][0]:a[k][1]].split("{", 1)[1]
            cases.append({"name": name, "line": line_of(src, i), "requirements": [float(x) for x in re.findall(r"[\d.]+", body("requirements"))], "holders": re.findall(r'"([^"]*)"', body("holders")),
                          "capacity": {k: float(v) for k, v in re.findall(r'"([^"]*)":\s*([\d.]+)', body("capacity"))}, "want": src[f["want"][0]:f["want"][1]].strip() == "true"})
            i = j
        elif src.startswith("//", i):
            i = src.index("\n", i)
        i += 1
    json.dump({"source": "accumulated_scenario_filters/idle_gpus/idle_gpus_test.go Test_greedyMatchRequirements", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases: print(c["line"], c["name"], c["requirements"], c["holders"], c["capacity"], c["want"])


if __name__ == "__main__":
    main()
