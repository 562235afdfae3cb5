#!/usr/bin/env python3
"""Known answers of the topology plugin's bottom-up level order → tests/golden/kat_level_order.json.

Source: pkg/scheduler/plugins/topology/topology_utils_test.go (TestReverseLevelOrder :12-140): five trees as nested DomainInfo literals with the expected order of their ids
(reverseLevelOrder, topology_utils.go:20-55: levels from the root down, emitted from the deepest level up, every level left to right) — the order in which
getJobAllocatableDomains' result is tried (job_filtering.go:526-542).  A case holds the tree as parent / children lists over the ids in literal order.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402


def top_fields(src, lo, hi):
    """key -> (value start, value end) of the composite literal whose braces are at lo / hi (line comments between the fields skipped)"""
    out, i, start = {}, lo + 1, lo + 1
    while i <= hi:
        c = src[i]
        if i == hi or c == ",":
            m = re.match(r"(?:\s|//[^\n]*\n)*(\w+):", src[start:i])
            if m:
                out[m.group(1)] = (start + m.end(), i)
            start = i + 1
        elif c in "({[":
            i = match(src, i)
        elif c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
        elif src.startswith("//", i):
            i = src.index("\n", i)
        i += 1
    return out

SRC = "/root/reference/pkg/scheduler/plugins/topology/topology_utils_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_level_order.json")


def domain(src, lo, hi, ids, parent, children, me=None):
    """the DomainInfo literal whose braces are at lo / hi: appends its id, returns its index"""
    f = top_fields(src, lo, hi)
    idx = len(ids)
    ids.append(re.search(r'"([^"]*)"', src[f["ID"][0]:f["ID"][1]]).group(1))
    parent.append(-1 if me is None else me); children.append([])
    if "Children" in f:
        b = src.index("{", f["Children"][0]); e = match(src, b)   # []*DomainInfo{ ... }
        i = b + 1
        while i < e:
            if src[i] == "{":
                j = match(src, i)
                children[idx].append(domain(src, i, j, ids, parent, children, idx))
                i = j
            elif src.startswith("//", i):
                i = src.index("\n", i)
            i += 1
    return idx


def main():
    src = open(SRC).read()
    start = src.index("}{", src.index("func TestReverseLevelOrder")) + 1
    end = match(src, start)
    cases, i = [], start + 1
    while i < end:
        if src[i] == "{":
            j = match(src, i)
            f = top_fields(src, i, j)
            name = re.search(r'"([^"]*)"', src[f["name"][0]:f["name"][1]]).group(1)
            ids, parent, children = [], [], []
            root_txt = src[f["root"][0]:f["root"][1]].strip()
            if not root_txt.startswith("nil"):
                b = src.index("{", f["root"][0])
                domain(src, b, match(src, b), ids, parent, children)
            exp_txt = src[f["expected"][0]:f["expected"][1]].strip()
            expected = None if exp_txt.startswith("nil") else re.findall(r'"([^"]*)"', exp_txt[exp_txt.index("{"):])
            cases.append({"name": name, "line": line_of(src, i), "ids": ids, "parent": parent, "children": children, "expected": expected})
            i = j
        elif src.startswith("//", i):
            i = src.index("\n", i)
        i += 1
    json.dump({"source": "plugins/topology/topology_utils_test.go TestReverseLevelOrder", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases:
        print(c["line"], c["name"], c["ids"], c["parent"], c["expected"])


if __name__ == "__main__":
    main()
