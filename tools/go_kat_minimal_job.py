#!/usr/bin/env python3
"""Known answers of MinimalJobRepresentatives → tests/golden/kat_minimal_job.json.

Source: pkg/scheduler/actions/common/minimal_job_comparison_test.go (a Ginkgo suite, :33-400): five maps of cases — IsEasierToSchedule with single pods, several pods, pods of
different phases; UpdateRepresentative with several pods — each with the representative's pods, the job's pods (resource lists, optionally a phase) and the expected verdict
(minimal_job_comparison.go:15-112).  A pod is written [pending (1 / 0), milli-CPU, memory bytes, GPUs]; only pending pods take part (a Failed or Running pod has nothing to allocate).
Go ranges its maps in random order; the cases are written in source order.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402
from go_kat_level_order import top_fields  # noqa: E402

SRC = "/root/reference/pkg/scheduler/actions/common/minimal_job_comparison_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_minimal_job.json")
SUFFIX = {"m": 1e-3, "Ki": 2.0**10, "Mi": 2.0**20, "Gi": 2.0**30, "k": 1e3, "M": 1e6, "G": 1e9, "": 1.0}


def qty(s):
    m = re.fullmatch(r"([\d.]+)(m|Ki|Mi|Gi|k|M|G|)", s)
    return float(m.group(1)) * SUFFIX[m.group(2)]


def pod(txt, phase="Pending"):
    r = {k: qty(v) for k, v in re.findall(r"(v1\.ResourceCPU|v1\.ResourceMemory|resource_info\.GPUResourceName):\s*resource\.MustParse\(\"([^\"]*)\"\)", txt)}
    return [1.0 if phase == "Pending" else 0.0, r.get("v1.ResourceCPU", 0.0) * 1000.0, r.get("v1.ResourceMemory", 0.0), r.get("resource_info.GPUResourceName", 0.0)]


def pods(src, span, kind):
    if span is None: return []
    txt = src[span[0]:span[1]]
    b = txt.index("{"); inner = txt[b:match(txt, b) + 1]
    if kind == "v1.ResourceList": return [pod(inner)]
    out, i = [], 1
    while i < len(inner) - 1:
        if inner[i] == "{":
            j = match(inner, i); item = inner[i:j + 1]
            ph = re.search(r"status:\s*v1\.Pod(\w+)", item)
            out.append(pod(item, ph.group(1) if ph else "Pending")); i = j
        i += 1
    return out


def main():
    src = open(SRC).read()
    upd = src.index('Describe("UpdateRepresentative"')
    cases = []
    for m in re.finditer(r"range map\[string\]struct \{", src):
        s0 = m.end() - 1; s1 = match(src, s0)                      # the struct type
        kind = re.search(r"representativeResources\s+(\S+)", src[s0:s1]).group(1)   # v1.ResourceList | []v1.ResourceList | []podTestsInfo
        lit = src.index("{", s1); end = match(src, lit)
        i = lit + 1
        while i < end:
            mm = re.compile(r'"((?:[^"\\]|\\.)*)":\s*\{').match(src, i)
            if mm:
                b = mm.end() - 1; e = match(src, b); f = top_fields(src, b, e)
                cases.append({"fn": "UpdateRepresentative" if m.start() > upd else "IsEasierToSchedule", "line": line_of(src, i), "name": mm.group(1),
                              "representative": pods(src, f.get("representativeResources"), kind), "job": pods(src, f.get("jobResources"), kind), "want": src[f["result"][0]:f["result"][1]].strip() == "true"})
                i = e
            elif src.startswith("//", i):
                i = src.index("\n", i)
            i += 1
    json.dump({"source": "actions/common/minimal_job_comparison_test.go", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases: print(c["line"], c["fn"], c["name"], c["representative"], c["job"], c["want"])


if __name__ == "__main__":
    main()
