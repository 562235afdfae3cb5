#!/usr/bin/env python3
"""Known answers of proportion's queue attributes → tests/golden/kat_queue_attributes.json.

Source: pkg/scheduler/plugins/proportion/resource_share/queue_attributes_test.go — three Ginkgo case maps: GetRequestedResource (:42-203, per resource four cases of
ResourceShare.GetRequestableShare), DominantResource (:205-276, six cases of GetDominantResourceShare over a total capacity — what the queue order compares) and
GetAllocatableShare (:278-335, five cases).  All values are literal; a resource map entry left out is Go's zero, `allUnlimited` is UnlimitedResourceQuantity in every
resource (:30-34).  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/plugins/proportion/resource_share/queue_attributes_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_queue_attributes.json")
RES = {"CpuResource": 0, "MemoryResource": 1, "GpuResource": 2}
EXPR = {"commonconstants.UnlimitedResourceQuantity": -1.0}


def val(e, src):
    e = e.strip()
    if e in EXPR:
        return EXPR[e]
    m = re.fullmatch(r"([\d.]+) ([*/]) (\w+|[\d.]+)", e)
    if m:
        b = m.group(3)
        if not re.fullmatch(r"[\d.]+", b):
            b = re.search(r"\b" + b + r"\s*=\s*([\d.]+)", open(os.path.join(os.path.dirname(SRC), "queue_resource_share.go")).read()).group(1)
        return float(m.group(1)) * float(b) if m.group(2) == "*" else float(m.group(1)) / float(b)
    return float(e)


def quantities(e, src):
    e = e.strip()
    if e == "allUnlimited":
        return [-1.0, -1.0, -1.0]
    m = re.fullmatch(r"ResourceQuantities\{(.*)\}", e, re.S); assert m, e
    out = [0.0, 0.0, 0.0]
    for k, v in re.findall(r"(\w+):\s*([^,}]+)", m.group(1)):
        out[RES[k]] = val(v, src)
    return out


def table(src, describe):
    d = src.index(f'Describe("{describe}"'); db = src.index("{", d); de = match(src, db)
    t = src.index("tests := map[string]", d); assert t < de
    decl = src.index("struct {", t); tb = src.index("{", match(src, src.index("{", decl)) + 1)
    return tb, match(src, tb), src[db:de]


def entries(src, lo, hi):
    """"name": { … } entries directly inside the braces lo / hi"""
    out, i = [], lo + 1
    while True:
        m = re.compile(r'(?:"([^"]+)"|string\((\w+)\)): \{').search(src, i, hi)
        if not m:
            return out
        b = m.end() - 1; e = match(src, b)
        out.append((m.group(1) or m.group(2), m.start(), b, e)); i = e + 1


def main():
    src = open(SRC).read()
    au = re.search(r"allUnlimited = ResourceQuantities\{([^}]*)\}", src, re.S)
    assert au and au.group(1).count("commonconstants.UnlimitedResourceQuantity") == 3
    doc = {"source": SRC.replace("/root/reference/", "")}
    lo, hi, body = table(src, "GetRequestedResource")
    assert "Expect(resourceShare.GetRequestableShare()).To(Equal(testData.expected))" in body
    req = []
    for res, _, b, e in entries(src, lo, hi):
        for name, at, cb, ce in entries(src, b, e):
            c = src[cb:ce]
            share = re.search(r"\b(GPU|CPU|Memory): ResourceShare\{([^}]*)\}", c)
            assert {"GPU": "GpuResource", "CPU": "CpuResource", "Memory": "MemoryResource"}[share.group(1)] == res, (res, share.group(1))
            f = {k: val(v, src) for k, v in re.findall(r"(\w+):\s*([^,\n]+),", share.group(2))}
            req.append({"resource": RES[res], "name": name, "line": line_of(src, at), "request": f["Request"], "max_allowed": f["MaxAllowed"], "want": val(re.search(r"expected:\s*([^,\n]+),", c).group(1), src)})
    doc["requestable_share"] = req
    lo, hi, body = table(src, "DominantResource")
    assert "resourceShare.MaxAllowed = commonconstants.UnlimitedResourceQuantity" in body and "GetDominantResourceShare(testData.totalCapacity)" in body
    dom = []
    for name, at, b, e in entries(src, lo, hi):
        f = dict(re.findall(r"(\w+):\s*((?:ResourceQuantities\{[^}]*\})|[^,\n]+),", src[b:e]))
        dom.append({"name": name, "line": line_of(src, at), "deserved": quantities(f["deserved"], src), "fair_share": quantities(f["fairShare"], src), "allocated": quantities(f["allocated"], src),
                    "total": quantities(f["totalCapacity"], src) if "totalCapacity" in f else [0.0, 0.0, 0.0], "want": val(f["expected"], src)})
    doc["dominant_share"] = dom
    lo, hi, body = table(src, "GetAllocatableShare")
    assert "allocatableShare := queueAttributes.GetAllocatableShare()" in body
    al = []
    for name, at, b, e in entries(src, lo, hi):
        f = dict(re.findall(r"(\w+):\s*((?:ResourceQuantities\{[^}]*\})|[^,\n]+),", src[b:e]))
        al.append({"name": name, "line": line_of(src, at), "deserved": quantities(f["deserved"], src), "fair_share": quantities(f["fairShare"], src), "max_allowed": quantities(f["maxAllowed"], src),
                   "want": quantities(f["expected"], src)})
    doc["allocatable_share"] = al
    assert (len(req), len(dom), len(al)) == (12, 6, 5), (len(req), len(dom), len(al))
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True); fh.write("\n")
    print(f"{out}: {len(req)} + {len(dom)} + {len(al)} cases")


if __name__ == "__main__":
    main()
