#!/usr/bin/env python3
"""Known answers of the AccumulatedNodeAffinities scenario filter → tests/golden/kat_node_affinities.json.

Source: pkg/scheduler/actions/common/solvers/accumulated_scenario_filters/node_affinities/node_affinities_test.go — the three tests in which no filter is created
(:220-246: no scenario; no pending pod with a node affinity; a preferred-only affinity) and the table of TestNodeAffinitiesFilter_Filter (:248-424).  The table is
literal: nodes are newNodeInfo(newNode(name, labels)), pods come from six helpers of the same file (:79-218) whose bodies this script checks for the field that makes
them what their name says (a node selector, a required In term on a label, a required matchFields term on metadata.name, a preferred term, nothing, a node name).
A pending pod = {kind, …}; a victim = the node it runs on.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/actions/common/solvers/accumulated_scenario_filters/node_affinities/node_affinities_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_node_affinities.json")

HELPERS = {  # helper → (kind, what its body must contain)
    "podWithNodeSelector": ("selector", "NodeSelector: selector"),
    "podWithNodeAffinity": ("required_in", "RequiredDuringSchedulingIgnoredDuringExecution"),
    "podWithNodeAffinityMatchFields": ("required_match_fields", "MatchFields"),
    "podWithPreferredNodeAffinityOnly": ("preferred_only", "PreferredDuringSchedulingIgnoredDuringExecution"),
    "podWithoutAffinity": ("none", "Spec: v1.PodSpec{}"),
    "victimPodOnNode": ("victim", "NodeName: nodeName"),
}


def func_body(src, name):
    i = src.index("func " + name + "(")
    b = src.index("{", src.index(")", i))
    return src[b:match(src, b) + 1]


def labels_of(text):
    text = text.strip()
    if text == "nil":
        return {}
    m = re.fullmatch(r"map\[string\]string\{(.*)\}", text, re.S)
    assert m, text
    return dict(re.findall(r'"([^"]*)":\s*"([^"]*)"', m.group(1)))


def call_args(src, i):
    """the top-level arguments of the call whose '(' is at i"""
    end = match(src, i)
    out, depth, start, j = [], 0, i + 1, i + 1
    while j < end:
        c = src[j]
        if c in "({[":
            j = match(src, j)
        elif c == '"':
            j += 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
        elif c == ",":
            out.append(src[start:j].strip()); start = j + 1
        j += 1
    if src[start:end].strip():
        out.append(src[start:end].strip())
    return out, end


def unq(s):
    assert s[0] == s[-1] == '"', s
    return s[1:-1]


def pods_of(src, lo, hi):
    out = []
    for m in re.finditer(r"\b(" + "|".join(HELPERS) + r")\(", src[lo:hi]):
        kind = HELPERS[m.group(1)][0]
        args, _ = call_args(src, lo + m.end() - 1)
        if kind == "selector":
            out.append({"kind": kind, "uid": unq(args[0]), "selector": labels_of(args[3])})
        elif kind == "required_in":
            out.append({"kind": kind, "uid": unq(args[0]), "key": unq(args[3]), "values": [unq(args[4])]})
        elif kind == "required_match_fields":
            out.append({"kind": kind, "uid": unq(args[0]), "node_names": [unq(args[3])]})
        elif kind == "preferred_only":
            out.append({"kind": kind, "uid": unq(args[0]), "key": unq(args[3]), "values": [unq(args[4])], "weight": int(args[5])})
        elif kind == "none":
            out.append({"kind": kind, "uid": unq(args[0])})
        else:
            out.append({"kind": kind, "uid": unq(args[0]), "node": unq(args[3])})
    return out


def nodes_of(src, lo, hi):
    out = {}
    for m in re.finditer(r"newNode\(", src[lo:hi]):
        args, _ = call_args(src, lo + m.end() - 1)
        out[unq(args[0])] = labels_of(args[1])
    return out


def field_span(src, lo, hi, key):
    m = re.search(r"\b" + key + r":", src[lo:hi])
    if not m:
        return None
    b = src.index("{", lo + m.end())
    # the literal's type sits between the colon and the brace: make sure the brace belongs to this field
    assert "\n" not in src[lo + m.end():b], (key, src[lo + m.end():b])
    return b, match(src, b)


def main():
    src = open(SRC).read()
    for name, (_, needle) in HELPERS.items():
        assert needle in func_body(src, name), (name, needle)
    # the three tests without a filter
    no_filter = []
    for fn, scenario, pend_helper in (("TestNewNodeAffinitiesFilter_NilScenario", False, None),
                                      ("TestNewNodeAffinitiesFilter_NoPendingTasksWithNodeAffinity", True, "podWithoutAffinity"),
                                      ("TestNewNodeAffinitiesFilter_PreferredOnlyNodeAffinityReturnsNil", True, "podWithPreferredNodeAffinityOnly")):
        body = func_body(src, fn)
        assert "assert.Nil(t, filter)" in body, fn
        at = src.index("func " + fn + "(")
        if scenario:
            assert pend_helper + "(" in body and "NewByNodeScenario(" in body, fn
            lo = src.index(body)
            pending = pods_of(src, lo, lo + len(body))
        else:
            assert "NewNodeAffinitiesFilter(nil," in body, fn
            pending = []
        no_filter.append({"name": fn, "line": line_of(src, at), "scenario": scenario, "pending": pending, "want_filter": False})
    # the table
    at = src.index("func TestNodeAffinitiesFilter_Filter(")
    t0 = src.index("tests := []struct", at)
    decl = src.index("{", t0); table = src.index("{", match(src, decl) + 1); table_end = match(src, table)
    cases, i = [], table + 1
    while True:
        m = re.compile(r"\{").search(src, i, table_end)
        if not m:
            break
        lo, hi = m.start(), match(src, m.start())
        name = re.search(r'name:\s*"([^"]*)"', src[lo:hi]).group(1)
        want = re.search(r"wantFilterResult:\s*(true|false)", src[lo:hi]).group(1) == "true"
        a = field_span(src, lo, hi, "allNodes"); f = field_span(src, lo, hi, "feasibleNodes"); p = field_span(src, lo, hi, "pendingTasks"); v = field_span(src, lo, hi, "victimTasks")
        case = {"name": name, "line": line_of(src, lo), "all_nodes": nodes_of(src, *a), "feasible": sorted(nodes_of(src, *f)), "pending": pods_of(src, *p),
                "victims": [x["node"] for x in pods_of(src, *v)] if v else [], "want": want}
        assert all(x["kind"] != "victim" for x in case["pending"]) and case["pending"], name
        cases.append(case); i = hi + 1
    # the loop of the test (:401-423): the filter is created on the scenario without victims and must exist, then asked about the scenario that holds the victims as POTENTIAL victims
    loop = src[table_end:src.index("\n}\n", table_end)]
    assert "NewByNodeScenario(ssn, nil, pendingPG, []*pod_info.PodInfo{}, []*podgroup_info.PodGroupInfo{})" in loop and "assert.NotNil(t, filter" in loop
    assert "NewByNodeScenario(ssn, nil, pendingPG, tt.victimTasks, []*podgroup_info.PodGroupInfo{})" in loop and "filter.Filter(filterSc)" in loop
    doc = {"source": SRC.replace("/root/reference/", ""), "no_filter_cases": no_filter, "filter_cases": cases}
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True); fh.write("\n")
    print(f"{out}: {len(no_filter)} + {len(cases)} cases")


if __name__ == "__main__":
    main()
