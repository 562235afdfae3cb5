#!/usr/bin/env python3
"""Known answers of the max-node-resources pre-predicate → tests/golden/kat_max_node_resources.json.

Source: pkg/scheduler/k8s_internal/predicates/maxNodeResources_test.go Test_podToMaxNodeResourcesFiltering :26-415 (six cases: a pod that fits, and pods asking for more CPU / memory /
whole GPUs / a GPU fraction / ephemeral storage than any ONE node of the pool has allocatable).  A case holds the nodes' allocatable resource lists (k8s quantity strings), the pod's
annotations and container requests, and whether PreFilter answers nil (schedulable) or Unschedulable (maxNodeResources.go:59-96).  TestMaxNodeResourcesPredicateDRA (ResourceClaims) is
not transcribed: DRA pods carry the fallback flag.  The k8s constants are written out as the strings they stand for.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/k8s_internal/predicates/maxNodeResources_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_max_node_resources.json")
NAMES = {"v1.ResourceCPU": "cpu", "v1.ResourceMemory": "memory", "v1.ResourcePods": "pods", "v1.ResourceEphemeralStorage": "ephemeral-storage", "resource_info.GPUResourceName": "nvidia.com/gpu",
         "commonconstants.PodGroupAnnotationForPod": "pod-group-name", "common_info.GPUFraction": "gpu-fraction"}


def key(tok):
    tok = tok.strip()
    return tok[1:-1] if tok.startswith('"') else NAMES[tok]


def quantities(txt):
    return {key(k): v for k, v in re.findall(r'([\w.]+|"[^"]*"):\s*resource\.MustParse\("([^"]*)"\)', txt)}


def main():
    src = open(SRC).read()
    at = src.index("func Test_podToMaxNodeResourcesFiltering")
    start = src.index("}{", at) + 1; end = match(src, start)
    cases, i = [], start + 1
    while i < end:
        if src[i] == "{":
            j = match(src, i); body = src[i:j + 1]
            name = re.match(r'\{\s*"([^"]*)"', body).group(1)
            nm = body.index("nodesMap:"); nb = body.index("{", nm); ne = match(body, nb)
            nodes, k = {}, nb + 1
            while k < ne:  # "n1": { Allocatable: ...ResourceList{ ... } }
                m = re.compile(r'"([^"]*)":\s*\{').search(body, k, ne)
                if not m: break
                b = m.end() - 1; e = match(body, b)
                nodes[m.group(1)] = quantities(body[b:e]); k = e + 1
            pm = body.index("pod:"); pb = body.index("{", pm); pe = match(body, pb); pod = body[pb:pe]
            ann = {}
            am = re.search(r"Annotations:\s*map\[string\]string\{", pod)
            if am:
                ab = am.end() - 1; ae = match(pod, ab)
                ann = {key(k2): v for k2, v in re.findall(r'([\w.]+|"[^"]*"):\s*"([^"]*)"', pod[ab:ae])}
            reqs = []
            for cm in re.finditer(r"Requests:\s*map\[v1\.ResourceName\]resource\.Quantity\{", pod):
                rb = cm.end() - 1; reqs.append(quantities(pod[rb:match(pod, rb)]))
            n_containers = len(re.findall(r"\bName:\s*\"c\d+\"", pod))
            em = body.index("expected{", pe); eb = body.index("{", em); exp = body[eb:match(body, eb)]
            schedulable = bool(re.match(r"\{\s*nil\s*,\s*false", exp))
            cases.append({"name": name, "line": line_of(src, i), "nodes": nodes, "annotations": ann, "containers": reqs + [{}] * (n_containers - len(reqs)), "schedulable": schedulable})
            i = j
        elif src.startswith("//", i):
            i = src.index("\n", i)
        i += 1
    json.dump({"source": "k8s_internal/predicates/maxNodeResources_test.go Test_podToMaxNodeResourcesFiltering", "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases ->", OUT)
    for c in cases:
        print(c["line"], c["name"], c["nodes"], c["annotations"], c["containers"], c["schedulable"])


if __name__ == "__main__":
    main()
