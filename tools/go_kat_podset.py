#!/usr/bin/env python3
"""Known answers of the pod-set's gang counters → tests/golden/kat_podset.json.

Source: pkg/scheduler/api/podgroup_info/subgroup_info/podset_test.go — TestIsReadyForScheduling :140, TestIsGangSatisfied :189, TestIsElastic :307,
TestGetNumActiveAllocatedTasks :353 (tables) and TestGetNumAliveTasks :229, TestGetNumActiveUsedTasks :248, TestGetNumGatedTasks :268, TestGetNumPendingTasks :287 (one case each).
Every case: NewPodSet("test", minAvailable, nil), AssignTask for the listed pods (UID, status) in order, one question (podset.go:56-125).  A case holds minAvailable, the pods as
[uid, status name] and the expected answer.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402
from go_kat_level_order import top_fields  # noqa: E402

SRC = "/root/reference/pkg/scheduler/api/podgroup_info/subgroup_info/podset_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_podset.json")
QUESTION = {"TestIsReadyForScheduling": "IsReadyForScheduling", "TestIsGangSatisfied": "IsGangSatisfied", "TestIsElastic": "IsElastic", "TestGetNumActiveAllocatedTasks": "GetNumActiveAllocatedTasks",
            "TestGetNumAliveTasks": "GetNumAliveTasks", "TestGetNumActiveUsedTasks": "GetNumActiveUsedTasks", "TestGetNumGatedTasks": "GetNumGatedTasks", "TestGetNumPendingTasks": "GetNumPendingTasks"}


def pods_of(txt):
    return [[u, s] for u, s in re.findall(r'UID:\s*"([^"]*)"\s*,\s*Status:\s*pod_status\.(\w+)', txt)]


def value(txt):
    txt = txt.strip()
    return {"true": True, "false": False}.get(txt, None) if txt in ("true", "false") else int(re.match(r"-?\d+", txt).group(0))


def main():
    src = open(SRC).read()
    cases = []
    for fn, question in QUESTION.items():
        at = src.index("func " + fn + "(")
        body_lo = src.index("{", at); body_hi = match(src, body_lo)
        body = src[body_lo:body_hi]
        ctor = re.search(r'NewPodSet\("test",\s*([\w.]+),\s*nil\)', body).group(1)
        if "tests := []struct" in body:
            start = src.index("}{", at) + 1; end = match(src, start)
            i = start + 1
            while i < end:
                if src[i] == "{":
                    j = match(src, i)
                    f = top_fields(src, i, j)
                    name = re.search(r'"([^"]*)"', src[f["name"][0]:f["name"][1]]).group(1)
                    mina = int(src[f["minAvailable"][0]:f["minAvailable"][1]].strip()) if ctor == "tt.minAvailable" else int(ctor)
                    pods = pods_of(src[f["pods"][0]:f["pods"][1]]) if "pods" in f else []
                    cases.append({"question": question, "test": fn, "line": line_of(src, i), "name": name, "minAvailable": mina, "pods": pods, "expected": value(src[f["expected"][0]:f["expected"][1]])})
                    i = j
                elif src.startswith("//", i):
                    i = src.index("\n", i)
                i += 1
        else:
            lst = body.index("[]*pod_info.PodInfo{"); b = body.index("{", lst); e = match(body, b)
            cases.append({"question": question, "test": fn, "line": line_of(src, at), "name": fn, "minAvailable": int(ctor), "pods": pods_of(body[b:e]),
                          "expected": int(re.search(r"expected := (\d+)", body).group(1))})
    json.dump({"source": "api/podgroup_info/subgroup_info/podset_test.go", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases:
        print(c["line"], c["question"], c["name"], c["minAvailable"], c["pods"], c["expected"])


if __name__ == "__main__":
    main()
