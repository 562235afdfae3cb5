#!/usr/bin/env python3
"""Known answers of GetTasksToAllocate → tests/golden/kat_tasks_to_allocate.json.

Source: pkg/scheduler/api/podgroup_info/allocation_info_test.go Test_GetTasksToAllocate :62-217 (eight jobs: pod-sets with their minAvailable and their tasks' statuses → the names of
the tasks the next allocation attempt takes, in order) and Test_getNumTasksToAllocate :396-461 (five pod-sets → the size of the next chunk: the gang's missing tasks up to
minAvailable, then one elastic task at a time; allocation_info.go:27-54, 145-177).  The test's order functions sort pod-sets by name and tasks by UID; its tasks are named in that
order.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402
from go_kat_level_order import top_fields  # noqa: E402

SRC = "/root/reference/pkg/scheduler/api/podgroup_info/allocation_info_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_tasks_to_allocate.json")


def table(src, fn):
    at = src.index("func " + fn + "(")
    start = src.index("{\n\t\t{", at); end = match(src, start)
    i = start + 1
    while i < end:
        if src[i] == "{":
            j = match(src, i); yield i, j; i = j
        elif src.startswith("//", i):
            i = src.index("\n", i)
        i += 1


def main():
    src = open(SRC).read()
    cases = []
    for i, j in table(src, "Test_GetTasksToAllocate"):
        f = top_fields(src, i, j)
        name = re.search(r'"([^"]*)"', src[f["name"][0]:f["name"][1]]).group(1)
        tasks = [{"name": n, "subGroup": g, "status": s} for n, g, s in re.findall(r'simpleTask\("([^"]*)",\s*"([^"]*)",\s*pod_status\.(\w+)\)', src[f["subGroupTasks"][0]:f["subGroupTasks"][1]])]
        mina = {k: int(v) for k, v in re.findall(r'"([^"]*)":\s*(\d+)', src[f["minAvailMap"][0]:f["minAvailMap"][1]])}
        want = re.findall(r'"([^"]*)"', src[f["wantTasks"][0]:f["wantTasks"][1]])
        cases.append({"fn": "GetTasksToAllocate", "line": line_of(src, i), "name": name, "minAvailable": mina, "tasks": tasks, "wantTasks": want, "wantNumTasks": int(src[f["wantNumTasks"][0]:f["wantNumTasks"][1]].strip())})
    for i, j in table(src, "Test_getNumTasksToAllocate"):
        f = top_fields(src, i, j)
        name = re.search(r'"([^"]*)"', src[f["name"][0]:f["name"][1]]).group(1)
        cases.append({"fn": "getNumTasksToAllocate", "line": line_of(src, i), "name": name, "minAvailable": int(src[f["minAvailable"][0]:f["minAvailable"][1]].strip()),
                      "statuses": re.findall(r"pod_status\.(\w+)", src[f["taskStatuses"][0]:f["taskStatuses"][1]].split("{", 1)[1]), "realAllocation": src[f["isRealAllocation"][0]:f["isRealAllocation"][1]].strip() == "true",
                      "want": int(re.match(r"\s*(\d+)", src[f["want"][0]:f["want"][1]]).group(1))})
    json.dump({"source": "api/podgroup_info/allocation_info_test.go", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases:
        print(c["line"], c["fn"], c["name"], c.get("minAvailable"), [(t["name"], t["subGroup"], t["status"]) for t in c.get("tasks", [])], c.get("wantTasks"), c.get("statuses"), c.get("want"))


if __name__ == "__main__":
    main()
