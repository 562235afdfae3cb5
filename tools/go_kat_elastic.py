#!/usr/bin/env python3
"""Known answers of the elastic plugin's job order → tests/golden/kat_elastic.json.

Source: pkg/scheduler/plugins/elastic/elastic_test.go TestJobOrderFn :17-533 — pairs of pod groups (the default pod-set's minAvailable; the statuses of the pods the test adds) and
JobOrderFn's verdict (elastic.go:25-65: a job below its minAvailable first, then one exactly at it, then one above).  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402
from go_kat_level_order import top_fields  # noqa: E402

SRC = "/root/reference/pkg/scheduler/plugins/elastic/elastic_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_elastic.json")


def main():
    src = open(SRC).read()
    start = src.index("}{", src.index("want int")) + 1; end = match(src, start)
    cases, i = [], start + 1
    while i < end:
        if src[i] == "{":
            j = match(src, i); f = top_fields(src, i, j)
            name = re.search(r'"([^"]*)"', src[f["name"][0]:f["name"][1]]).group(1)
            a0 = src.index("{", f["args"][0]); a = top_fields(src, a0, match(src, a0))
            side = {}
            for k in ("l", "r"):
                group = src[a[k][0]:a[k][1]]
                side[k + "MinAvailable"] = int(re.search(r"NewPodSet\(podgroup_info\.DefaultSubGroup,\s*(\d+),", group).group(1))
                side[k + "Pods"] = re.findall(r"Status:\s*pod_status\.(\w+)", src[a[k + "Pods"][0]:a[k + "Pods"][1]])
            cases.append(dict({"name": name, "line": line_of(src, i)}, **side, want=int(src[f["want"][0]:f["want"][1]].strip())))
            i = j
        elif src.startswith("//", i):
            i = src.index("\n", i)
        i += 1
    json.dump({"source": "plugins/elastic/elastic_test.go TestJobOrderFn", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases: print(c["line"], c["name"], c["lMinAvailable"], c["lPods"], c["rMinAvailable"], c["rPods"], c["want"])


if __name__ == "__main__":
    main()
