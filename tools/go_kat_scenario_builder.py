#!/usr/bin/env python3
"""Known answers of PodAccumulatedScenarioBuilder → tests/golden/kat_scenario_builder.json.

Source: pkg/scheduler/actions/common/solvers/pod_scenario_builder_test.go — a Ginkgo suite of nine specs (:36-285) over three helpers (:287-434): initializeSession(jobs,
tasksPerJob) puts `jobs` running jobs of `tasksPerJob` one-GPU pods on one node that is exactly full, one queue per job; createJobWithTasks builds the pending reclaimer;
the specs then set the gangs' minAvailable, pick recorded victims (two whole jobs "by index" of a Go map range — any two, the jobs are alike — or ONE named pod of the only
job, handed over as CloneWithTasks), build the victims queue over the session's jobs and walk GetValidScenario / GetNextScenario.  What a spec expects is counts: how many
scenarios, potential victims per scenario, recorded victim jobs per scenario, the size of the job representative of every potential victim of the last scenario.
The suite is imperative; this script reads each spec (its Context's BeforeEach in front of it) for exactly those calls and expectations and fails on a spec it cannot
account for.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/actions/common/solvers/pod_scenario_builder_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_scenario_builder.json")


def blocks(src, lo, hi, head):
    """(title, body start, body end) of every `head("title", func() {` between lo and hi"""
    out = []
    for m in re.finditer(re.escape(head) + r'\("([^"]*)", func\(\) \{', src[lo:hi]):
        b = lo + m.end() - 1
        out.append((m.group(1), lo + m.start(), b + 1, match(src, b)))
    return out


def spec(src, ctx_title, before, title, at, text):
    t = re.sub(r"//[^\n]*", "", before + text)
    d = {"context": ctx_title, "name": title, "line": line_of(src, at)}
    (j, k), = set(re.findall(r"initializeSession\((\d+), (\d+)\)", t)); d["jobs"], d["tasks_per_job"] = int(j), int(k)
    (n, jid, gpu), = set(re.findall(r'createJobWithTasks\((\d+), (\d+), "team-a", v1\.PodPending, \[\]v1\.ResourceRequirements\{(requireOneGPU\(\))?\}\)', t))
    d["reclaimer_tasks"], d["reclaimer_gpus_per_task"] = int(n), 1 if gpu else 0
    if "SetMinAvailable(int32(len(podGroupInfo.GetAllPodsMap())))" in t:
        d["min_available"] = "all"
    elif re.search(r"minAvailable := (\d+)", t):
        assert "SetMinAvailable(int32(minAvailable))" in t
        d["min_available"] = int(re.search(r"minAvailable := (\d+)", t).group(1))
    else:
        assert "SetMinAvailable" not in t
        d["min_available"] = 1  # createJobWithTasks: MinMember 1 (:375-378)
    m = re.search(r"recordedVictimIndexes := \[\]int\{([^}]*)\}", t)
    e = re.search(r'podInfo\.Name == "pod-(\d+)"', t)
    if m:
        assert "slices.Contains(recordedVictimIndexes, podGroupIndex)" in t
        d["recorded"] = {"whole_jobs": len(m.group(1).split(","))}
    elif e:
        assert "podGroupInfo.CloneWithTasks(partialTasks)" in t and re.search(r"CloneWithTasks\(partialTasks\)\)\s*break", t)
        d["recorded"] = {"pod_of_first_job": int(e.group(1))}
    else:
        assert "recordedVictimsJobs := []*podgroup_info.PodGroupInfo{}" in t or "[]*podgroup_info.PodGroupInfo{},\n" in t, title
        d["recorded"] = None
    assert len(re.findall(r"NewPodAccumulatedScenarioBuilder\(", t)) == 1 and "utils.GetVictimsQueue(ssn, nil)" in t
    w = {}
    if "Expect(scenarioBuilder.GetValidScenario()).To(Not(BeNil()))" in t: w["first_scenario"] = True
    if "Expect(scenarioBuilder.GetValidScenario()).To(BeNil())" in t: w["first_scenario"] = False
    if "Expect(scenarioBuilder.GetNextScenario()).To(BeNil())" in t: w["next_of_empty_queue_is_nil"] = True
    m = re.search(r"Expect\(numberOfGeneratedScenarios\)\.To\(Equal\((\d+)\)\)", t)
    if m: w["scenarios"] = int(m.group(1))
    m = re.search(r"potentialVictimsPerScenario := \[\]int\{([^}]*)\}", t)
    if m:
        assert "Expect(len(sn.PotentialVictimsTasks())).To(Equal(potentialVictimsPerScenario[numberOfGeneratedScenarios]))" in t and "Expect(numberOfGeneratedScenarios).To(Equal(len(potentialVictimsPerScenario)))" in t
        w["potential_per_scenario"] = [int(x) for x in m.group(1).split(",")]
    if "Expect(len(sn.RecordedVictimsJobs())).To(Equal(len(recordedVictimsJobs)))" in t: w["recorded_jobs_in_every_scenario"] = True
    m = re.search(r"Expect\(len\(lastScenario\.PotentialVictimsTasks\(\)\)\)\.To\(Equal\((\d+)\)\)", t)
    if m:
        assert "Expect(lastScenario).NotTo(BeNil())" in t
        w["last_potential"] = int(m.group(1))
        w["last_representative_size"] = int(re.search(r"Expect\(len\(matchingJob\.GetAllPodsMap\(\)\)\)\.To\(Equal\((\d+)\)\)", t).group(1))
    assert w and t.count("Expect(") == sum((1 if k in ("first_scenario", "next_of_empty_queue_is_nil", "scenarios", "recorded_jobs_in_every_scenario") else 0) for k in w) \
        + (3 if "potential_per_scenario" in w else 0) + (3 if "last_potential" in w else 0), (title, w, t.count("Expect("))
    d["want"] = w
    return d


def main():
    src = open(SRC).read()
    for fn, needle in (("initializeSession", "node.Idle.Add(newJob.Allocated)"), ("initializeSession", 'queueName := fmt.Sprintf("team-%d", jobID)'), ("createJobWithTasks", "MinMember: 1,"),
                       ("createQueue", 'ParentQueue: "default"'), ("buildPod", 'pod.Spec.NodeName = "node-1"'), ("requireOneGPU", 'resource_info.GPUResourceName: resource.MustParse("1")')):
        i = src.index("func " + fn + "("); b = src.index("{\n", i)
        assert needle in src[b:match(src, b)], (fn, needle)
    d0 = src.index('Describe("PodAccumulatedScenarioBuilder"'); db = src.index("{", d0); de = match(src, db)
    specs = []
    for ctitle, _, cb, ce in blocks(src, db, de, "Context"):
        m = re.search(r"BeforeEach\(func\(\) \{", src[cb:ce])
        before = ""
        if m:
            bb = cb + m.end() - 1; before = src[bb + 1:match(src, bb)]
        for title, at, ib, ie in blocks(src, cb, ce, "It"):
            specs.append(spec(src, ctitle, before, title, at, src[ib:ie]))
    assert len(specs) == len(re.findall(r"\bIt\(", src)), (len(specs), len(re.findall(r"\bIt\(", src)))
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    with open(out, "w") as fh:
        json.dump({"source": SRC.replace("/root/reference/", ""), "specs": specs}, fh, indent=1, sort_keys=True); fh.write("\n")
    print(f"{out}: {len(specs)} specs")


if __name__ == "__main__":
    main()
