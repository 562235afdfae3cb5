#!/usr/bin/env python3
"""Known answers of the minruntime plugin's victim filters and scenario validators → tests/golden/kat_minruntime.json.

Source: pkg/scheduler/plugins/minruntime/minruntime_test.go — eleven Ginkgo specs (:98-339; the three parseMinRuntime specs are about the plugin's arguments) on the queue
tree of createTestQueues (resolver_test.go:346-431) with the defaults of the suite's BeforeEach (:78-95).  A spec builds a pending job and ONE victim job through
createPodGroup(uid, queue, lastStartTime | nil, minAvailable, pods) (:46-76: the pods are Running), maybe sets the resolve method, maybe lists pods of the victim
as the scenario's victims, calls one of preemptFilterFn / reclaimFilterFn / preemptScenarioValidatorFn / reclaimScenarioValidatorFn and expects true or false.  The
suite is imperative; this script reads each spec for exactly those pieces and fails on one it cannot account for.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

DIR = "/root/reference/pkg/scheduler/plugins/minruntime/"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_minruntime.json")
FNS = ("preemptFilterFn", "reclaimFilterFn", "preemptScenarioValidatorFn", "reclaimScenarioValidatorFn")


def queues():
    src = open(DIR + "resolver_test.go").read()
    at = src.index("func createTestQueues()"); b = src.index("{\n", at); body = src[b:match(src, b)]
    out = {}
    for m in re.finditer(r"&queue_info\.QueueInfo\{", body):
        lo = m.end() - 1; q = body[lo:match(body, lo)]
        dur = lambda k: (lambda x: None if x.group(1) == "nil" else int(re.search(r"Duration: (\d+) \* time\.Second", x.group(1)).group(1)))(re.search(k + r":\s*(nil|&metav1\.Duration\{[^}]*\})", q))
        out[re.search(r'UID:\s*"([^"]*)"', q).group(1)] = {"parent": re.search(r'ParentQueue:\s*"([^"]*)"', q).group(1), "preempt_s": dur("PreemptMinRuntime"), "reclaim_s": dur("ReclaimMinRuntime")}
    assert len(out) == 8
    return out


def main():
    src = open(DIR + "minruntime_test.go").read()
    be = src.index("BeforeEach(func() {"); bb = src.index("{", be); before = src[bb:match(src, bb)]
    d_pre = int(re.search(r"defaultPreemptDuration = metav1\.Duration\{Duration: (\d+) \* time\.Second\}", before).group(1))
    d_rec = int(re.search(r"defaultReclaimDuration = metav1\.Duration\{Duration: (\d+) \* time\.Second\}", before).group(1))
    assert "reclaimResolveMethod:     resolveMethodLCA" in before and "Status: pod_status.Running" in src[:be]
    specs = []
    for dm in re.finditer(r'Describe\("(\w+)", func\(\) \{', src):
        if dm.group(1) not in FNS:
            continue
        db = dm.end() - 1; de = match(src, db)
        for im in re.finditer(r'It\("([^"]*)", func\(\) \{', src[db:de]):
            ib = db + im.end() - 1; t = re.sub(r"//[^\n]*", "", src[ib:match(src, ib)])
            groups = re.findall(r'(\w+) := createPodGroup\("[^"]*", "([^"]*)", (nil|&\w+), (\d+), (\d+)\)', t)
            assert len(groups) == 2 and groups[0][2] == "nil", (im.group(1), groups)
            (pv, pq, _, _, _), (vv, vq, start, m_av, pods) = groups
            ago = None
            if start != "nil":
                ago = int(re.search(re.escape(start[1:]) + r" := (?:now|time\.Now\(\))\.Add\(-(\d+) \* time\.Second\)", t).group(1))
            method = re.search(r"plugin\.reclaimResolveMethod = resolveMethod(\w+)", t)
            call = re.findall(r"result := plugin\.(\w+)\(([^)]*)\)", t); assert len(call) == 1 and call[0][0] == dm.group(1), (im.group(1), call)
            victims = 0
            if "Validator" in dm.group(1):
                assert re.search(r"Preemptor: " + pv + r",", t) and re.search(r"Victims:\s+map\[common_info\.PodGroupID\]\*api\.VictimInfo\{" + vv + r"\.UID: \{Job: " + vv + r", Tasks: \w+\}\}", t)
                victims = len(re.findall(r'UID:\s+"victim-job-pod-\d+"', t)); assert victims >= 1
            else:
                assert call[0][1].replace(" ", "") == f"{pv},{vv}"
            want = re.search(r"Expect\(result\)\.To\(Be(True|False)\(\)", t).group(1) == "True"
            assert t.count("Expect(") == 1
            specs.append({"fn": dm.group(1), "name": im.group(1), "line": line_of(src, db + im.start()), "pending_queue": pq, "victim_queue": vq, "victim_started_ago_s": ago,
                          "victim_min_available": int(m_av), "victim_pods": int(pods), "scenario_victim_tasks": victims, "resolve_method": (method.group(1) if method else "LCA").lower(), "want": want})
    assert len(specs) == 11, len(specs)
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    with open(out, "w") as fh:
        json.dump({"source": "pkg/scheduler/plugins/minruntime/minruntime_test.go (+ createTestQueues of resolver_test.go)", "queues": queues(), "default_preempt_s": d_pre, "default_reclaim_s": d_rec,
                   "specs": specs}, fh, indent=1, sort_keys=True); fh.write("\n")
    print(f"{out}: {len(specs)} specs")


if __name__ == "__main__":
    main()
