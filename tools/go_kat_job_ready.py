#!/usr/bin/env python3
"""Known answers of PodGroupInfo.IsReadyForScheduling → tests/golden/kat_job_ready.json.

Source: pkg/scheduler/api/podgroup_info/job_info_test.go TestPodGroupInfo_IsReadyForScheduling :397-779 — pod groups written as Go literals: either NewPodGroupInfo(uid, tasks…) with
the default pod-set's minAvailable set by the test, or PodSets: {"name": NewPodSet(name, minAvailable, nil).WithPodInfos({…})}; every task is a v1.Pod with a phase and, for a gated
one, scheduling gates (pod_info.go:410-445 turns a pending pod with gates into status Gated).  A case holds its pod-sets (name, minAvailable, the tasks' statuses in source order)
and whether the job is ready (job_info.go:399-406, podset.go:114-120).  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402
from go_kat_level_order import top_fields  # noqa: E402

SRC = "/root/reference/pkg/scheduler/api/podgroup_info/job_info_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_job_ready.json")


def statuses(txt):
    out = []
    for m in re.finditer(r"pod_info\.NewTaskInfo\(", txt):
        p = m.end() - 1; call = txt[p:match(txt, p) + 1]
        phase = re.search(r"Phase:\s*v1\.Pod(\w+)", call).group(1)
        out.append("Gated" if (phase == "Pending" and "SchedulingGates" in call) else phase)
    return out


def main():
    src = open(SRC).read()
    at = src.index("func TestPodGroupInfo_IsReadyForScheduling")
    start = src.index("}{", at) + 1; end = match(src, start)
    cases, i = [], start + 1
    while i < end:
        if src[i] == "{":
            j = match(src, i); f = top_fields(src, i, j)
            name = re.search(r'"([^"]*)"', src[f["name"][0]:f["name"][1]]).group(1)
            job = src[f["job"][0]:f["job"][1]]
            sets = []
            for m in re.finditer(r'subgroup_info\.NewPodSet\("([^"]*)",\s*(\d+),\s*nil\)\.\s*WithPodInfos\(', job):
                p = m.end() - 1; sets.append({"name": m.group(1), "minAvailable": int(m.group(2)), "statuses": statuses(job[p:match(job, p) + 1])})
            if not sets:
                mina = int(re.search(r"int32\((\d+)\)", src[f["minAvailable"][0]:f["minAvailable"][1]]).group(1))
                sets = [{"name": "default", "minAvailable": mina, "statuses": statuses(job)}]
            cases.append({"name": name, "line": line_of(src, i), "podSets": sets, "ready": src[f["expected"][0]:f["expected"][1]].strip() == "true"})
            i = j
        elif src.startswith("//", i):
            i = src.index("\n", i)
        i += 1
    json.dump({"source": "api/podgroup_info/job_info_test.go TestPodGroupInfo_IsReadyForScheduling", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases:
        print(c["line"], c["name"], [(s["name"], s["minAvailable"], s["statuses"]) for s in c["podSets"]], c["ready"])


if __name__ == "__main__":
    main()
