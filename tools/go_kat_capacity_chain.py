#!/usr/bin/env python3
"""Known answers of the capacity policy over a queue CHAIN → tests/golden/kat_capacity_chain.json.

Source: pkg/scheduler/plugins/proportion/capacity_policy/capacity_policy_test.go — a Ginkgo suite of four case maps, fourteen cases: IsJobOverQueueCapacity
(:25-171 max allowed, :173-327 non-preemptible quota), IsNonPreemptibleJobOverQuota (:330-485), IsTaskAllocationOnNodeOverCapacity (:488-1080; every case asks
for CPU only, so the node's share of the request is the request).  A case is three queues top → mid → leaf with literal shares (a field left out is Go's zero), a
job of one pending task in the leaf, and whether the result is schedulable.  tools/go_kat_capacity.py pins the two checks on ONE queue; this one pins the walk up the
chain.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/plugins/proportion/capacity_policy/capacity_policy_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_capacity_chain.json")
CALLS = {"IsJobOverQueueCapacity": "capacityPolicy.IsJobOverQueueCapacity(testData.job, tasksToAllocate)",
         "IsNonPreemptibleJobOverQuota": "capacityPolicy.IsNonPreemptibleJobOverQuota(testData.job, tasksToAllocate)",
         "IsTaskAllocationOnNodeOverCapacity": 'capacityPolicy.IsTaskAllocationOnNodeOverCapacity(testData.job.GetAllPodsMap()["task-a"],'}


def num(v):
    v = v.strip()
    return -1.0 if v == "commonconstants.UnlimitedResourceQuantity" else float(v)


def main():
    src = open(SRC).read()
    cases = []
    for dm in re.finditer(r'Describe\("(Is\w+)", func\(\) \{', src):
        fn = dm.group(1); db = dm.end() - 1; de = match(src, db)
        assert CALLS[fn] in src[db:de], fn
        for tm in re.finditer(r"tests := map\[string\]struct \{", src[db:de]):
            decl = db + tm.end() - 1; table = src.index("{", match(src, decl) + 1); tend = match(src, table)
            i = table + 1
            while True:
                m = re.compile(r'"([^"]+)": \{').search(src, i, tend)
                if not m:
                    break
                lo = m.end() - 1; hi = match(src, lo); body = src[lo:hi]
                qm = re.search(r"queues: map\[common_info\.QueueID\]\*rs\.QueueAttributes\{", body); qb = lo + qm.end() - 1; qe = match(src, qb)
                queues, k = {}, qb + 1
                while True:
                    q = re.compile(r'"([^"]+)": \{').search(src, k, qe)
                    if not q:
                        break
                    ql = q.end() - 1; qh = match(src, ql); qbody = src[ql:qh]
                    shares = {}
                    for rm in re.finditer(r"\b(GPU|CPU|Memory): rs\.ResourceShare\{", qbody):
                        rl = ql + rm.end() - 1
                        shares[rm.group(1)] = {a: num(b) for a, b in re.findall(r"(\w+):\s*([^,\n]+),", src[rl + 1:match(src, rl)])}
                    queues[q.group(1)] = {"parent": re.search(r'ParentQueue:\s*"([^"]*)"', qbody).group(1), "shares": shares}
                    k = qh + 1
                jb = src[lo + body.index("job: &podgroup_info.PodGroupInfo{"):hi]
                pre = re.search(r"Preemptibility:\s*v2alpha2\.(\w+)", jb)
                r1 = re.search(r"NewResourceRequirementsWithGpus\(([\d.]+)\)", jb); r3 = re.search(r"NewResourceRequirements\(([\d.]+), ([\d.]+), ([\d.]+)\)", jb)
                req = [0.0, 0.0, float(r1.group(1))] if r1 else [float(r3.group(2)), float(r3.group(3)), float(r3.group(1))]  # [cpu, memory, gpu]
                assert len(re.findall(r"ResReq:", jb)) == 1 and "Status:    pod_status.Pending" in jb
                cases.append({"fn": fn, "name": m.group(1), "line": line_of(src, m.start()), "queues": queues, "job_queue": re.search(r'Queue:\s*"([^"]*)"', jb).group(1),
                              "preemptible": bool(pre) and pre.group(1) == "Preemptible", "requested": req,
                              "want_schedulable": re.search(r"expectedResult:\s*(true|false)", body).group(1) == "true"})
                i = hi + 1
    assert len(cases) == len(re.findall(r"expectedResult:\s*(?:true|false)", src)) == 14, len(cases)
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    with open(out, "w") as fh:
        json.dump({"source": SRC.replace("/root/reference/", ""), "cases": cases}, fh, indent=1, sort_keys=True); fh.write("\n")
    print(f"{out}: {len(cases)} cases")


if __name__ == "__main__":
    main()
