#!/usr/bin/env python3
"""Known answers of the victim search's scenario objects → tests/golden/kat_scenarios.json.

Source: pkg/scheduler/actions/common/solvers/scenario/base_scenario_test.go (TestPodSimpleScenario_AddPotentialVictimsTasks :22-242, _GetVictimJobRepresentativeById
:244-506, _LatestPotentialVictim :508-674) and by_node_scenario_test.go (TestPodByNodeScenario_VictimsTasksFromNodes :22-522): four literal tables, twelve cases.
A case holds the session's pod groups, the potential victims handed to the constructor (added one task at a time, base_scenario.go:49-51), the recorded victim jobs,
the tasks of one AddPotentialVictimsTasks call, the question's argument and the expected answer.  Pod literals go through the Go literal parser of tools/go_fixtures.py;
a task is identified by (job, name) — the tests' pods carry no UID.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import go_fixtures as G  # noqa: E402
from go_kat_resource_division import match, line_of  # noqa: E402

DIR = "/root/reference/pkg/scheduler/actions/common/solvers/scenario/"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_scenarios.json")
G.CONSTS["commonconstants.PodGroupAnnotationForPod"] = "pod-group-name"


def skip_string(src, i):
    i += 1
    while src[i] != '"':
        i += 2 if src[i] == "\\" else 1
    return i


def top_fields(src, lo, hi):
    """key → (value start, value end) of the composite literal whose braces are at lo / hi"""
    out, i, start = {}, lo + 1, lo + 1
    while i <= hi:
        c = src[i]
        if i == hi or c == ",":
            item = src[start:i]
            m = re.match(r"\s*(\w+):", item)
            if m:
                out[m.group(1)] = (start + m.end(), i)
            start = i + 1
        elif c in "({[":
            i = match(src, i)
        elif c == '"':
            i = skip_string(src, i)
        elif src.startswith("//", i):
            i = src.index("\n", i)
        i += 1
    return out


def brace_of(src, span):
    lo, hi = span
    b = src.index("{", lo)
    assert b < hi
    return b, match(src, b)


def tasks_in(src, lo, hi):
    """the pod literals of a span, in source order"""
    out = []
    for m in re.finditer(r"pod_info\.NewTaskInfo\(&v1\.Pod\{", src[lo:hi]):
        brace = lo + m.end() - 1
        pod = G._j(G.ev(G.Parser(src, brace - len("&v1.Pod")).parse_expr()))
        meta, spec, status = pod.get("ObjectMeta") or {}, pod.get("Spec") or {}, pod.get("Status") or {}
        ann = meta.get("Annotations") or {}
        ann = ann.get("map", ann) if isinstance(ann, dict) else ann
        out.append({"name": meta.get("Name"), "job": ann.get("pod-group-name"), "node": spec.get("NodeName") or None, "phase": status.get("Phase") or None})
    return out


def jobs_in(src, lo, hi):
    """podgroup_info.NewPodGroupInfo(name, tasks...) calls of a span, in source order"""
    out = []
    for m in re.finditer(r'podgroup_info\.NewPodGroupInfo\("([^"]*)"', src[lo:hi]):
        p = lo + m.start() + len("podgroup_info.NewPodGroupInfo")
        out.append({"name": m.group(1), "tasks": tasks_in(src, p, match(src, p))})
    return out


def ident(tasks):
    return [[t["job"], t["name"]] for t in tasks]


def cases_of(path, func):
    src = open(path).read()
    at = src.index("func " + func + "(")
    t0 = src.index("tests := []struct", at)
    decl = src.index("{", t0); table = src.index("{", match(src, decl) + 1); table_end = match(src, table)
    i, out = table + 1, []
    while True:
        m = re.compile(r"\{").search(src, i, table_end)
        if not m:
            break
        lo, hi = m.start(), match(src, m.start())
        f = top_fields(src, lo, hi)
        name = re.match(r'\s*"([^"]*)"', src[f["name"][0]:f["name"][1]]).group(1)
        fl, fh = brace_of(src, f["fields"]); ff = top_fields(src, fl, fh)
        case = {"func": func, "name": name, "line": line_of(src, lo), "file": os.path.basename(path)}
        case["session_jobs"] = jobs_in(src, *ff["session"])
        pj = jobs_in(src, *ff["pendingTasksAsJob"]); assert len(pj) == 1 and not pj[0]["tasks"]
        case["ctor_potential"] = ident(tasks_in(src, *ff["potentialVictimsTasks"]))
        case["recorded_jobs"] = [{"name": j["name"], "tasks": ident(j["tasks"])} for j in jobs_in(src, *ff["recordedVictimsJobs"])] if "recordedVictimsJobs" in ff else []
        al, ah = brace_of(src, f["args"]); af = top_fields(src, al, ah)
        case["added"] = ident(tasks_in(src, *af["tasks"])) if "tasks" in af else []
        if "victimPodInfo" in af:
            (v,) = tasks_in(src, *af["victimPodInfo"]); case["victim"] = [v["job"], v["name"]]
        if "nodeNames" in af:
            case["node_names"] = re.findall(r'"([^"]*)"', src[af["nodeNames"][0]:af["nodeNames"][1]])
        # every task the case names, with what the scene needs of it
        pods = {}
        for j in case["session_jobs"]:
            for t in j["tasks"]:
                pods[(j["name"], t["name"])] = t
        for span in (ff["potentialVictimsTasks"], f["args"]) + ((ff["recordedVictimsJobs"],) if "recordedVictimsJobs" in ff else ()):
            for t in tasks_in(src, *span):
                pods.setdefault((t["job"], t["name"]), t)
        case["pods"] = [{"job": k[0], "name": k[1], "node": v["node"], "phase": v["phase"]} for k, v in sorted(pods.items())]
        if func.endswith("AddPotentialVictimsTasks"):
            el, eh = brace_of(src, f["expected"]); ef = top_fields(src, el, eh)
            groups = {}
            gl, gh = brace_of(src, ef["victimsJobsTaskGroups"])
            for k, span in top_fields_map(src, gl, gh).items():
                groups[k] = len(jobs_in(src, *span))
            case["want"] = {"potential": ident(tasks_in(src, *ef["potentialVictimsTasks"])), "groups_per_job": groups}
        elif func.endswith("GetVictimJobRepresentativeById") or func.endswith("LatestPotentialVictim"):
            w = src[f["want"][0]:f["want"][1]].strip()
            if w == "nil":
                case["want"] = None
            else:
                (j,) = jobs_in(src, *f["want"]); case["want"] = {"job": j["name"], "tasks": ident(j["tasks"])}
        else:
            case["want"] = ident(tasks_in(src, *f["want"]))
        out.append(case); i = hi + 1
    return out


def top_fields_map(src, lo, hi):
    """"key": value entries of a map literal"""
    out, i, start = {}, lo + 1, lo + 1
    while i <= hi:
        c = src[i]
        if i == hi or c == ",":
            m = re.match(r'\s*"([^"]*)":', src[start:i])
            if m:
                out[m.group(1)] = (start + m.end(), i)
            start = i + 1
        elif c in "({[":
            i = match(src, i)
        elif c == '"':
            i = skip_string(src, i)
        i += 1
    return out


def main():
    cases = []
    for fn in ("TestPodSimpleScenario_AddPotentialVictimsTasks", "TestPodSimpleScenario_GetVictimJobRepresentativeById", "TestPodSimpleScenario_LatestPotentialVictim"):
        cases += cases_of(DIR + "base_scenario_test.go", fn)
    cases += cases_of(DIR + "by_node_scenario_test.go", "TestPodByNodeScenario_VictimsTasksFromNodes")
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    with open(out, "w") as fh:
        json.dump({"source": "pkg/scheduler/actions/common/solvers/scenario/{base_scenario,by_node_scenario}_test.go", "cases": cases}, fh, indent=1, sort_keys=True); fh.write("\n")
    print(f"{out}: {len(cases)} cases")


if __name__ == "__main__":
    main()
