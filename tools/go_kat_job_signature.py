#!/usr/bin/env python3
"""Known answers of the job's scheduling-constraints signature → tests/golden/kat_job_signature.json.

Source: pkg/scheduler/api/podgroup_info/job_info_test.go TestPodGroupInfo_GetSchedulingConstraintsSignature :913-1416 — twelve pairs of pod groups built by closures (a tree of
SubGroupSets / PodSets with topology constraints, pending and running pods assigned to the pod-sets) and whether their signatures must be equal (job_info.go:547-570,
podset.go:150-196).  The closures use five kinds of statements (a constraint literal bound to a name, NewSubGroupSet, NewPodSet, AddPodSet / AddSubGroup, AssignTask of a pending
or running pod); this script interprets exactly those and writes each pod group as a tree: {name, kind, constraint, minAvailable, pods: [[name, "Pending" | "Running"]], children}.
Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/api/podgroup_info/job_info_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_job_signature.json")


def constraint(txt):
    f = dict(re.findall(r'(\w+):\s*"([^"]*)"', txt))
    return {"topology": f.get("Topology", ""), "required": f.get("RequiredLevel", ""), "preferred": f.get("PreferredLevel", "")}


def build(body):
    """interprets one closure; returns the root of its tree"""
    lits = []

    def stash(m):
        lits.append(constraint(m.group(1))); return f"TC{len(lits) - 1}"
    body = re.sub(r"&topology_info\.TopologyConstraintInfo\{([^{}]*)\}", stash, body, flags=re.S)
    body = re.sub(r",\s*\n\s*", ", ", body)  # (a call's arguments over several lines)
    env, nodes = {}, {}

    def tc(expr):
        expr = expr.strip()
        if expr == "nil": return None
        if re.fullmatch(r"TC\d+", expr): return lits[int(expr[2:])]
        return env[expr]

    def name(expr):
        expr = expr.strip()
        return {"subgroup_info.RootSubGroupSetName": "", "DefaultSubGroup": "default"}.get(expr, expr.strip('"'))
    root = None
    for ln in body.splitlines():
        ln = re.sub(r"\s*//.*$", "", ln).strip()  # (trailing comments)
        m = re.fullmatch(r"(\w+) := (TC\d+)", ln)
        if m: env[m.group(1)] = lits[int(m.group(2)[2:])]; continue
        m = re.fullmatch(r"(\w+) := subgroup_info\.NewSubGroupSet\((.+?),\s*([^,]+)\)", ln)
        if m:
            nodes[m.group(1)] = {"name": name(m.group(2)), "kind": "SubGroupSet", "constraint": tc(m.group(3)), "children": []}
            if m.group(2).strip() == "subgroup_info.RootSubGroupSetName": root = nodes[m.group(1)]
            continue
        m = re.fullmatch(r"(\w+) := subgroup_info\.NewPodSet\((.+?),\s*(\d+),\s*([^,]+)\)", ln)
        if m: nodes[m.group(1)] = {"name": name(m.group(2)), "kind": "PodSet", "constraint": tc(m.group(4)), "minAvailable": int(m.group(3)), "pods": []}; continue
        m = re.fullmatch(r"(\w+)\.(AddPodSet|AddSubGroup)\((\w+)\)", ln)
        if m: nodes[m.group(1)]["children"].append(nodes[m.group(3)]); continue
        m = re.fullmatch(r'(\w+)\.AssignTask\(create(Pending|Running)Task\("([^"]+)"\)\)', ln)
        if m: nodes[m.group(1)]["pods"].append([m.group(3), m.group(2)]); continue
        if "NewSubGroupSet" in ln or "NewPodSet" in ln or "AssignTask" in ln or ".Add" in ln:
            raise SyntaxError("statement not understood: " + ln)
    assert root is not None
    return root


def main():
    src = open(SRC).read()
    at = src.index("func TestPodGroupInfo_GetSchedulingConstraintsSignature")
    start = src.index("}{", src.index("expectEqual bool", at)) + 1; end = match(src, start)
    cases, i = [], start + 1
    while i < end:
        if src[i] == "{":
            j = match(src, i); body = src[i:j + 1]
            nm = re.search(r'name:\s*"([^"]*)"', body).group(1)
            trees = []
            for key in ("podGroupA", "podGroupB"):
                k = body.index(key + ": func()"); b = body.index("{", k); trees.append(build(body[b + 1:match(body, b)]))
            cases.append({"name": nm, "line": line_of(src, i), "a": trees[0], "b": trees[1], "equal": re.search(r"expectEqual:\s*(true|false)", body).group(1) == "true"})
            i = j
        elif src.startswith("//", i):
            i = src.index("\n", i)
        i += 1
    json.dump({"source": "api/podgroup_info/job_info_test.go TestPodGroupInfo_GetSchedulingConstraintsSignature", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases:
        print(c["line"], c["name"], c["equal"])


if __name__ == "__main__":
    main()
