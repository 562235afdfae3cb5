#!/usr/bin/env python3
"""Known answers of the TopologyAwareIdleGpus scenario filter → tests/golden/kat_topo_idle_gpus.json.

Source: pkg/scheduler/actions/common/solvers/accumulated_scenario_filters/idle_gpus/topology_aware_idle_gpus_test.go — the twelve tests that build a cluster, a
pending job with topology-constrained sub-groups, optionally running victims, one or two scenarios, create the filter on the first and ask it about each
(TestTopologyDomainKey_Equality is about Go map keys and has no counterpart).  The tests are imperative but stereotyped; this script walks each function body:
  * string variables (`x := "…"`, `x := topology + "/rack"`, `xs := []string{…}`) and the label table `map[string]map[string]string{node: {levelExpr: value}}`,
  * createTestNodes(ctrl, map[string]int{node: gpus}, labels | nil)                                                   (:574-620),
  * createRunningPodWithGpus(name, ns, node, gpus) + nodes[node].AddTask(victimTask): a running pod that holds devices (:622-641),
  * newConstrainedSubGroup(name, topology, requiredLevel, podCount): a SubGroupSet with that constraint over one PodSet (:542-551),
  * buildJob(name, tasks{SubGroupName, RequiredGPUs}, root, …)                                                         (:553-563),
  * scenario.NewByNodeScenario(session | nil, job, job, potential victims | nil, recorded victim jobs | nil),
  * NewTopologyAwareIdleGpusFilter(scenario, nodes), then runFilterCheck(t, filter, scenario, want, reason) in order — or the nil-filter assertion of the first test.
Anything a body holds beyond that shape makes the script fail rather than guess.  Only the reference is read."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from go_kat_resource_division import match, line_of  # noqa: E402

SRC = "/root/reference/pkg/scheduler/actions/common/solvers/accumulated_scenario_filters/idle_gpus/topology_aware_idle_gpus_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_topo_idle_gpus.json")
CONSTS = {"podgroup_info.DefaultSubGroup": "default"}  # api/podgroup_info/job_info.go: DefaultSubGroup = "default"
SKIP = {"TestTopologyDomainKey_Equality"}


def strip_comments(s):
    return re.sub(r"//[^\n]*", "", s)


def split_top(s):
    """top-level comma-separated pieces of s"""
    out, i, start = [], 0, 0
    while i < len(s):
        c = s[i]
        if c in "({[":
            i = match(s, i)
        elif c == '"':
            i += 1
            while s[i] != '"':
                i += 2 if s[i] == "\\" else 1
        elif c == ",":
            out.append(s[start:i].strip()); start = i + 1
        i += 1
    if s[start:].strip():
        out.append(s[start:].strip())
    return out


class Body:
    def __init__(self, text):
        self.t = strip_comments(text)
        self.vars = dict(CONSTS)
        for m in re.finditer(r"^\s*(\w+) := (\[\]string\{[^}]*\}|[^\n{]+)$", self.t, re.M):
            name, rhs = m.group(1), m.group(2).strip()
            if rhs.startswith("[]string{"):
                self.vars[name] = [self.expr(x) for x in split_top(rhs[len("[]string{"):-1])]
            elif re.fullmatch(r'"[^"]*"|\w+( \+ "[^"]*")+|"[^"]*"( \+ \w+)+', rhs):
                self.vars[name] = self.expr(rhs)

    def expr(self, e):
        e = e.strip()
        if " + " in e:
            return "".join(self.expr(x) for x in e.split(" + "))
        if e.startswith('"'):
            assert e.endswith('"') and '"' not in e[1:-1], e
            return e[1:-1]
        m = re.fullmatch(r"(\w+)\[(\d+)\]", e)
        if m:
            return self.vars[m.group(1)][int(m.group(2))]
        assert e in self.vars, ("unknown expression", e)
        return self.vars[e]

    def calls(self, fn):
        """argument lists of every call of fn, in order"""
        out = []
        for m in re.finditer(r"\b" + re.escape(fn) + r"\(", self.t):
            i = m.end() - 1
            out.append((m.start(), split_top(self.t[i + 1:match(self.t, i)])))
        return out

    def label_tables(self):
        out = {}
        for m in re.finditer(r"(\w+) := map\[string\]map\[string\]string\{", self.t):
            b = m.end() - 1
            table = {}
            for item in split_top(self.t[b + 1:match(self.t, b)]):
                node, inner = item.split(":", 1)
                inner = inner.strip(); assert inner[0] == "{" and inner[-1] == "}", item
                table[self.expr(node)] = {self.expr(kv.rsplit(":", 1)[0]): self.expr(kv.rsplit(":", 1)[1]) for kv in split_top(inner[1:-1])}
            out[m.group(1)] = table
        return out


def int_of(e):
    m = re.fullmatch(r"ptr\.To\(int64\((\d+)\)\)|(\d+)", e.strip())
    assert m, e
    return int(m.group(1) or m.group(2))


def walk(name, text, line):
    b = Body(text)
    labels = b.label_tables()
    (_, args), = b.calls("createTestNodes")
    caps = args[1]; assert caps.startswith("map[string]int{"), caps
    nodes = {}
    for item in split_top(caps[len("map[string]int{"):-1]):
        n, g = item.split(":"); nodes[b.expr(n)] = {"gpus": int(g), "labels": {}}
    if args[2] != "nil":
        for n, l in labels[args[2]].items():
            nodes[n]["labels"] = l
    # victims: pod, task, AddTask — names of the three variables tie them together
    victims, task_of = [], {}
    for at, a in b.calls("createRunningPodWithGpus"):
        var = re.search(r"(\w+) := $", b.t[:at]).group(1)
        victims.append({"name": b.expr(a[0]), "node": b.expr(a[2]), "gpus": int(a[3]), "_pod_var": var})
    for at, a in b.calls("pod_info.NewTaskInfo"):
        var = re.search(r"(\w+) := $", b.t[:at]).group(1)
        v, = [v for v in victims if v["_pod_var"] == a[0]]
        task_of[var] = v["name"]
        assert re.search(r'nodes\["' + re.escape(v["node"]) + r'"\]\.AddTask\(' + var + r"\)", b.t), (name, var)
    for v in victims:
        del v["_pod_var"]
    # recorded victim jobs: NewPodGroupInfo + AddTaskInfo(task)
    job_tasks = {}
    for m in re.finditer(r"(\w+)\.AddTaskInfo\((\w+)\)", b.t):
        job_tasks.setdefault(m.group(1), []).append(task_of[m.group(2)])
    doc = {"name": name, "line": line, "nodes": nodes, "victims": victims}
    subgroups = []
    for _, a in b.calls("newConstrainedSubGroup"):
        subgroups.append({"name": b.expr(a[0]), "topology": b.expr(a[1]), "required_level": b.expr(a[2]), "pods": int(a[3])})
    doc["subgroups"] = subgroups
    if b.calls("buildJob"):
        (_, a), = b.calls("buildJob")
        assert a[1].startswith("[]*tasks_fake.TestTaskBasic{") and a[2] == "rootSubGroupSet", a
        tasks = []
        for item in split_top(a[1][len("[]*tasks_fake.TestTaskBasic{"):-1]):
            f = dict(kv.split(":", 1) for kv in split_top(item.strip()[1:-1]))
            f = {k.strip(): v.strip() for k, v in f.items()}
            assert f["State"] == "pod_status.Pending" and set(f) == {"SubGroupName", "State", "RequiredGPUs"}, f
            tasks.append({"subgroup": b.expr(f["SubGroupName"]), "gpus": int_of(f["RequiredGPUs"])})
        doc["tasks"] = tasks
    else:  # the first test: jobs_fake.TestJobBasic without a sub-group tree
        m = re.search(r"RequiredGPUsPerTask:\s*(\d+)", b.t); assert m and "RootSubGroupSet:     nil" in text and len(re.findall(r"\{State: pod_status\.Pending\}", b.t)) == 1, name
        doc["tasks"] = [{"subgroup": "default", "gpus": int(m.group(1))}]
    scenarios = {}
    for at, a in b.calls("scenario.NewByNodeScenario"):
        var = re.search(r"(\w+) := $", b.t[:at]).group(1)
        assert a[1] == a[2] and len(a) == 5, a
        pot = [] if a[3] == "nil" else [task_of[x] for x in split_top(re.fullmatch(r"\[\]\*pod_info\.PodInfo\{(.*)\}", a[3], re.S).group(1))]
        rec = [] if a[4] == "nil" else [t for x in split_top(re.fullmatch(r"\[\]\*podgroup_info\.PodGroupInfo\{(.*)\}", a[4], re.S).group(1)) for t in job_tasks[x]]
        scenarios[var] = {"potential": pot, "recorded": rec}
    (_, a), = b.calls("NewTopologyAwareIdleGpusFilter")
    doc["created_on"] = a[0]; assert a[0] in scenarios and a[1] == "nodes"
    checks = b.calls("runFilterCheck")
    if checks:
        assert 'if filter == nil {' in b.t and "Expected non-nil filter" in b.t
        doc["want_filter"] = True
        doc["calls"] = []
        for _, c in checks:
            assert c[0] == "t" and c[1] == "filter" and c[3] in ("true", "false"), c
            doc["calls"].append(dict(scenarios[c[2]], scenario=c[2], want=c[3] == "true"))
        assert doc["calls"][0]["scenario"] == doc["created_on"]
    else:
        assert "if filter != nil {" in b.t and "Expected nil filter" in b.t
        doc["want_filter"] = False
        doc["calls"] = [dict(scenarios[a[0]], scenario=a[0], want=None)]
    return doc


def main():
    src = open(SRC).read()
    for fn, needle in (("newConstrainedSubGroup", "subGroup.AddPodSet(subgroup_info.NewPodSet(name, podCount, nil))"), ("createRunningPodWithGpus", "Status: v1.PodStatus{Phase: v1.PodRunning}"),
                       ("createTestNodes", '"nvidia.com/gpu": resource.MustParse(strconv.Itoa(nodeCapacities[nodeName]))'), ("runFilterCheck", "if valid != wantValid {")):
        i = src.index("func " + fn + "("); bb = src.index("{\n", i)
        assert needle in src[bb:match(src, bb)], fn
    cases = []
    for m in re.finditer(r"^func (Test\w+)\(t \*testing\.T\) \{", src, re.M):
        if m.group(1) in SKIP:
            continue
        b = m.end() - 1
        cases.append(walk(m.group(1), src[b + 1:match(src, b)], line_of(src, m.start())))
    out = sys.argv[1] if len(sys.argv) > 1 else OUT
    with open(out, "w") as fh:
        json.dump({"source": SRC.replace("/root/reference/", ""), "cases": cases}, fh, indent=1, sort_keys=True); fh.write("\n")
    print(f"{out}: {len(cases)} cases, {sum(len(c['calls']) for c in cases)} filter calls")


if __name__ == "__main__":
    main()
