#!/usr/bin/env python3
"""Known answers of the shared-GPU pieces → tests/golden/kat_gpu_sharing.json.

Sources: pkg/scheduler/plugins/gpupack/gpupack_test.go and plugins/gpuspread/gpuspread_test.go (a Ginkgo map of six cases each: the GPU-order score of one device group
from the node's GPU memory and the memory used on the group; two of the six expect the "invalid GPU memory" error), and pkg/scheduler/gpu_sharing/gpuSharing_test.go
(Test_getNodePreferableGpuForSharing, four cases: the groups a fraction pod takes out of the fitting GPUs of a node — the table's nodeSharingInfo argument is never
handed to the function, so the node carries no shared-GPU state: a numbered group counts as pipelined).  Node and pod literals go through the Go literal parser of
tools/go_fixtures.py.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import go_fixtures as G  # noqa: E402
from go_kat_resource_division import match, line_of  # noqa: E402

ROOT = "/root/reference/pkg/scheduler/"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_gpu_sharing.json")
G.CONSTS.update({"commonconstants.PodGroupAnnotationForPod": "pod-group-name", "commonconstants.GpuFraction": "gpu-fraction", "commonconstants.GpuFractionsNumDevices": "gpu-fraction-num-devices",
                 "pod_info.WholeGpuIndicator": "-2", "v1.ResourceCPU": "cpu", "v1.ResourceMemory": "memory", "v1.ResourcePods": "pods"})


def strip(m):
    return m.get("map", m) if isinstance(m, dict) else m


def order_cases(plugin):
    path = ROOT + f"plugins/{plugin}/{plugin}_test.go"
    src = open(path).read()
    assert "actualScore, err := gpuOrderFn(task, nodeInfo, caseSpec.gpuIdx)" in src and "MemoryOfEveryGpuOnNode: totalGpuMem" in src and "UsedSharedGPUsMemory: map[string]int64{" in src
    at = src.index("cases := map[string]struct"); decl = src.index("{", at); table = src.index("{", match(src, decl) + 1); end = match(src, table)
    out = []
    for m in re.finditer(r'"([^"]+)": \{', src[table:end]):
        lo = table + m.end() - 1; body = src[lo:match(src, lo)]
        f = dict(re.findall(r"(\w+):\s*([^,\n]+),", body))
        whole = f["gpuIdx"].strip() == "pod_info.WholeGpuIndicator"
        if not whole:  # the group the fake node holds memory for
            assert f["gpuIdx"].strip().strip('"') == re.search(r'UsedSharedGPUsMemory: map\[string\]int64\{\s*"([^"]+)": usedGpuMem', src).group(1)
        out.append({"plugin": plugin, "name": m.group(1), "line": line_of(src, table + m.start()), "total_mem": int(f["totalMem"]), "used_mem": int(f["usedMem"]), "whole_gpu": whole,
                    "want_score": float(f["expectedScore"]), "want_error": "expectedErr" in f})
    assert len(out) == 6, out
    return out


def sharing_cases():
    path = ROOT + "gpu_sharing/gpuSharing_test.go"
    src = open(path).read()
    loop = src[src.index("for _, tt := range tests {"):]
    assert "GetNodePreferableGpuForSharing(\n\t\t\t\ttt.args.fittingGPUsOnNode, tt.args.node, tt.args.pod, tt.args.isPipelineOnly)" in loop and "nodeSharingInfo" not in loop
    at = src.index("tests := []struct"); decl = src.index("{", at); table = src.index("{", match(src, decl) + 1); end = match(src, table)
    out, i = [], table + 1
    while True:
        m = re.compile(r"\{").search(src, i, end)
        if not m:
            break
        lo, hi = m.start(), match(src, m.start()); body = src[lo:hi]
        name = re.search(r'name:\s*"([^"]*)"', body).group(1)
        fit = re.search(r"fittingGPUsOnNode:\s*\[\]string\{([^}]*)\}", body).group(1)
        fitting = [("whole" if x.strip() == "pod_info.WholeGpuIndicator" else x.strip().strip('"')) for x in fit.split(",") if x.strip()]
        node = G._j(G.ev(G.Parser(src, lo + body.index("&v1.Node{") + 1).parse_expr())) if False else None
        nm = re.search(r"n := (&v1\.Node\{)", body); nb = lo + nm.start(1)
        node = G._j(G.ev(G.Parser(src, nb).parse_expr()))
        alloc = strip((node.get("Status") or {}).get("Allocatable") or {})
        pm = re.search(r"pod_info\.NewTaskInfo\((&v1\.Pod\{)", body)
        pod = G._j(G.ev(G.Parser(src, lo + pm.start(1)).parse_expr()))
        ann = strip((pod.get("ObjectMeta") or {}).get("Annotations") or {})
        req = {}
        for c in (pod.get("Spec") or {}).get("Containers") or []:
            req.update(strip(((c.get("Resources") or {}).get("Requests")) or {}))
        w = src[lo + body.index("want: want{"):hi]
        groups = re.search(r"expectedGroupsInList:\s*(make\(\[\]string, 0\)|\[\]string\{([^}]*)\})", w)
        out.append({"name": name, "line": line_of(src, lo), "fitting": fitting, "node_allocatable": {k: str(v) for k, v in alloc.items()},
                    "pod": {"gpu_request": req.get("nvidia.com/gpu"), "gpu_fraction": ann.get("gpu-fraction"), "num_devices": ann.get("gpu-fraction-num-devices")},
                    "pipeline_only": re.search(r"isPipelineOnly:\s*(true|false)", body).group(1) == "true",
                    "want": {"groups": int(re.search(r"groupLength:\s*(\d+)", w).group(1)), "includes": re.findall(r'"([^"]*)"', groups.group(2) or ""),
                             "releasing": re.search(r"isReleasing:\s*(true|false)", w).group(1) == "true"}})
        i = hi + 1
    assert len(out) == 4
    return out


def main():
    doc = {"sources": ["plugins/gpupack/gpupack_test.go", "plugins/gpuspread/gpuspread_test.go", "gpu_sharing/gpuSharing_test.go"],
           "gpu_order": order_cases("gpupack") + order_cases("gpuspread"), "preferable_gpu_for_sharing": sharing_cases()}
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True); fh.write("\n")
    print(f"{out}: {len(doc['gpu_order'])} + {len(doc['preferable_gpu_for_sharing'])} cases")


if __name__ == "__main__":
    main()
