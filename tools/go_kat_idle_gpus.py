#!/usr/bin/env python3
"""Known answers of the AccumulatedIdleGpus scenario filter → tests/golden/kat_idle_gpus.json.

Source: pkg/scheduler/actions/common/solvers/accumulated_scenario_filters/idle_gpus/idle_gpus_test.go — Test_orderedInsert (:28-104, the three cases ordered by
cmp.Compare; the fourth passes a comparison closure), TestAccumulatedIdleGpus_updateWithVictim (:201-305), _updateStateWithScenario (:307-945), _Filter (:947-1384).
The filter's state (`fields` / `want`) is literal; the scenarios are built by stereotyped Go code — pod_info.NewTaskInfo(&v1.Pod{…}) values, some bound to names inside a
closure, handed to scenario.NewByNodeScenario(session, nil, pendingJob, potentialVictims, recordedVictimJobs) — which this script walks: every pod literal goes through the
Go literal parser of tools/go_fixtures.py, the call's arguments are split at top-level commas.  A task = {uid, node, gpus}; a victim's AcceptedResource is its request
(node_info.AddTask → setAcceptedResources for whole devices).  The third Filter case asks for GPU MEMORY (gpu-memory annotation): kept, with `gpu_memory_mib` and the
node's `gpu_memory` label so that the oracle can turn it into a portion.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import go_fixtures as G  # noqa: E402
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/actions/common/solvers/accumulated_scenario_filters/idle_gpus/idle_gpus_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_idle_gpus.json")
G.CONSTS["commonconstants.PodGroupAnnotationForPod"] = "pod-group-name"
G.CONSTS["commonconstants.GpuMemory"] = "gpu-memory"
G.CONSTS["commonconstants.NvidiaGpuMemory"] = "nvidia.com/gpu.memory"


def split_args(src, lo, hi):
    """top-level comma-separated spans of src[lo:hi]"""
    out, i, start = [], lo, lo
    while i < hi:
        c = src[i]
        if c in "({[":
            i = match(src, i)
        elif c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
        elif src.startswith("//", i):
            i = src.index("\n", i)
        elif c == ",":
            out.append((start, i)); start = i + 1
        i += 1
    if src[start:hi].strip():
        out.append((start, hi))
    return out


def case_blocks(src, fn_name):
    """the elements of `tests := []struct{…}{ {…}, {…} }` inside func fn_name"""
    f = src.index("func " + fn_name + "(")
    m = re.compile(r"tests := \[\](struct|testCase\[string\])").search(src, f)
    if m.group(1) == "struct":
        type_open = src.index("{", m.end()); lit_open = src.index("{", match(src, type_open) + 1)
    else:
        lit_open = src.index("{", m.end())
    lit_close = match(src, lit_open)
    return [(a, b) for a, b in split_args(src, lit_open + 1, lit_close) if src[a:b].strip().startswith("{")]


def literal_at(src, key, lo, hi, ty):
    m = re.search(r"\b" + key + r":\s*" + ty + r"\{", src[lo:hi])
    if not m:
        return None
    return G._j(G.ev(G.Parser(src, lo + m.end() - 1).parse_composite(ty)))


def strip(d):
    return {k: v for k, v in (d or {}).items() if not k.startswith("_")} if isinstance(d, dict) else d


def task_from_pod(pod):
    meta, spec = pod.get("ObjectMeta") or {}, pod.get("Spec") or {}
    gpus = 0.0
    for c in (spec.get("Containers") or []):
        req = strip(((c.get("Resources") or {}).get("Requests")) or {})
        if "nvidia.com/gpu" in req:
            gpus += float(req["nvidia.com/gpu"])
    t = {"uid": meta.get("UID"), "node": spec.get("NodeName") or None, "gpus": gpus}
    ann = strip(meta.get("Annotations") or {})
    if "gpu-memory" in ann:
        t["gpu_memory_mib"] = int(ann["gpu-memory"])
    return t


def tasks_in(src, lo, hi, named):
    """tasks of a span, in source order: inline NewTaskInfo(&v1.Pod{…}) values and names bound earlier"""
    out = []
    for m in re.finditer(r"pod_info\.NewTaskInfo\(&v1\.Pod\{|\b(\w+)\b", src[lo:hi]):
        pos = lo + m.start()
        if m.group(0).startswith("pod_info.NewTaskInfo"):
            brace = lo + m.end() - 1
            out.append((pos, task_from_pod(G._j(G.ev(G.Parser(src, brace - len("&v1.Pod")).parse_expr())))))
        elif m.group(1) in named:
            out.append((pos, named[m.group(1)]))
    # names inside an inline literal's own span would be duplicates: keep items that do not lie inside an earlier inline literal
    spans, kept = [], []
    for m in re.finditer(r"pod_info\.NewTaskInfo\(", src[lo:hi]):
        spans.append((lo + m.start(), match(src, lo + m.end() - 1)))
    for pos, t in out:
        inner = any(a < pos <= b for a, b in spans)
        if not inner or src.startswith("pod_info.NewTaskInfo", pos):
            kept.append(t)
    return kept


def scenario_of(src, lo, hi):
    named = {}
    for m in re.finditer(r"(\w+)\s*:=\s*pod_info\.NewTaskInfo\(&v1\.Pod\{", src[lo:hi]):
        brace = lo + m.end() - 1
        named[m.group(1)] = task_from_pod(G._j(G.ev(G.Parser(src, brace - len("&v1.Pod")).parse_expr())))
    node_mem = {}
    for m in re.finditer(r"(\w+)\s*:=\s*&v1\.Node\{", src[lo:hi]):  # nodes with a gpu-memory label (the GPU-memory case)
        node = G._j(G.ev(G.Parser(src, lo + m.end() - len("&v1.Node{")).parse_expr()))
        labels = strip((node.get("ObjectMeta") or {}).get("Labels") or {})
        for k, v in labels.items():
            if "gpu.memory" in k or "gpu-memory" in k:
                node_mem[(node.get("ObjectMeta") or {}).get("Name")] = int(v)
    call = src.index("scenario.NewByNodeScenario(", lo)
    close = match(src, call + len("scenario.NewByNodeScenario"))
    args = split_args(src, call + len("scenario.NewByNodeScenario("), close)
    pending = tasks_in(src, args[2][0], args[2][1], named)
    potential = tasks_in(src, args[3][0], args[3][1], named)
    recorded = tasks_in(src, args[4][0], args[4][1], named)
    sc = {"pending": pending, "potential_victims": potential, "recorded_victims": recorded}
    if node_mem:
        sc["node_gpu_memory_mib"] = node_mem
    return sc


def main(out=OUT):
    src = open(SRC).read()
    doc = {"source": "pkg/scheduler/actions/common/solvers/accumulated_scenario_filters/idle_gpus/idle_gpus_test.go (Test_orderedInsert, TestAccumulatedIdleGpus_updateWithVictim / "
                     "_updateStateWithScenario / _Filter); `line` = the case's line; generated by tools/go_kat_idle_gpus.py"}
    # Test_orderedInsert: array / value / replace / want (cmp.Compare[string]; the case with a comparison closure is left out)
    oi = []
    for a, b in case_blocks(src, "Test_orderedInsert"):
        blk = src[a:b]
        if "cmp: func(" in blk:
            continue
        args = literal_at(src, "args", a, b, r"args\[string\]")
        oi.append({"line": line_of(src, a), "name": re.search(r'name:\s*"([^"]*)"', blk).group(1), "array": list(args["array"]), "value": args["value"], "replace": bool(args["replace"]),
                   "want": list(G._j(G.ev(G.Parser(src, a + re.search(r"want:\s*\[\]string\{", blk).end() - 1).parse_composite({"list": "string"}))))})
    doc["ordered_insert"] = oi
    uv = []
    for a, b in case_blocks(src, "TestAccumulatedIdleGpus_updateWithVictim"):
        blk = src[a:b]
        f = G._j(G.ev(G.Parser(src, a + re.search(r"fields:\s*fields\{", blk).end() - 1).parse_composite("fields")))
        elems = f.get("_elems") or []
        want = literal_at(src, "want", a, b, "want")
        # the victim is a &pod_info.PodInfo{NodeName, UID, AcceptedResource: …NewGpuResourceRequirementWithGpus(count, portion)} literal (a dereferenced call the literal parser does not take)
        g = re.search(r"NewGpuResourceRequirementWithGpus\(\s*([\d.]+)\s*,\s*([\d.]+)", blk)
        uv.append({"line": line_of(src, a), "name": re.search(r'name:\s*"([^"]*)"', blk).group(1), "idle": strip(elems[0]), "sorted": list(elems[1]),
                   "victim": {"uid": re.search(r'UID:\s*"([^"]*)"', blk).group(1), "node": re.search(r'NodeName:\s*"([^"]*)"', blk).group(1), "gpus": float(g.group(1)) + float(g.group(2))},
                   "min_relevant": re.search(r'minIdleGpusRelevant:\s*"([^"]*)"', blk).group(1),
                   "want_min_relevant": want["minIdleGpusRelevant"], "want_sorted": list(want["maxFreeGpuNodesSorted"])})
    doc["update_with_victim"] = uv
    for key, fn in (("update_state", "TestAccumulatedIdleGpus_updateStateWithScenario"), ("filter", "TestAccumulatedIdleGpus_Filter")):
        rows = []
        for a, b in case_blocks(src, fn):
            blk = src[a:b]
            f = strip(literal_at(src, "fields", a, b, "fields")); w = strip(literal_at(src, "want", a, b, "want") or {})
            row = {"line": line_of(src, a), "name": re.search(r'name:\s*"([^"]*)"', blk).group(1),
                   "fields": {"required": list(f.get("requiredGpusSorted") or []), "idle": strip(f.get("nodesNameToIdleGpus") or {}), "sorted": list(f.get("maxFreeGpuNodesSorted") or []),
                              "pending_in_state": sorted(strip(f.get("pendingTasksInState") or {})), "recorded_in_cache": sorted(strip(f.get("recordedVictimsInCache") or {})),
                              "potential_in_cache": sorted(strip(f.get("potentialVictimsInCache") or {}))},
                   "scenario": scenario_of(src, a, b)}
            if key == "update_state":
                row["first"] = bool(re.search(r"isFirstScenario:\s*true", blk))
                row["want"] = {"err": bool(w.get("wantErr", False))}
                for gk, jk in (("nodesNameToIdleGpus", "idle"), ("maxFreeGpuNodesSorted", "sorted"), ("pendingTasksInState", "pending_in_state"), ("recordedVictimsInCache", "recorded_in_cache"),
                               ("potentialVictimsInCache", "potential_in_cache")):
                    if gk in w:
                        v = w[gk]
                        row["want"][jk] = strip(v) if jk == "idle" else (list(v) if jk == "sorted" else sorted(strip(v)))
            else:
                row["want"] = {"valid": bool(w.get("validScenario", False)), "err": bool(w.get("err", False))}
            rows.append(row)
        doc[key] = rows
    json.dump(doc, open(out, "w"), indent=1)
    print({k: len(v) for k, v in doc.items() if isinstance(v, list)}, "→", out)


if __name__ == "__main__":
    main(*sys.argv[1:2])
