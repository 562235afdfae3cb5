#!/usr/bin/env python3
"""Known answers of the reference's reclaim arithmetic → tests/golden/kat_reclaimable.json.

Source: pkg/scheduler/plugins/proportion/reclaimable/reclaimable_test.go.  The two `CanReclaimResources` tables (:34-531) are Go
composite literals and are parsed with the literal parser of tools/go_fixtures.py.  The `Reclaimable` specs (:533-1185) are a BeforeEach
fixture plus a few assignments per spec; those are restated below by hand, one entry per `It`, each with its line.  Only the reference is
read; this script and the JSON are committed.

Layout of a case: queues {name: [parent, {res: [Deserved, FairShare, MaxAllowed, Allocated, AllocatedNotPreemptible]}]}, res in cpu / memory / gpu,
reclaimer [queue, [milli-cpu, memory, gpus], preemptible], reclaimees [[queue, [milli-cpu, memory, gpus]]], want.
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import go_fixtures as G  # noqa: E402

SRC = "/root/reference/pkg/scheduler/plugins/proportion/reclaimable/reclaimable_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_reclaimable.json")
UNL = -1.0  # commonconstants.UnlimitedResourceQuantity


def share(d):
    f = (d or {}).get("fields", {}) if isinstance(d, dict) and "fields" in d else (d or {})
    def num(k):
        v = f.get(k, 0)
        if isinstance(v, (dict, str)): return UNL  # the constant's identifier
        return float(v)
    return [num("Deserved"), num("FairShare"), num("MaxAllowed"), num("Allocated"), num("AllocatedNotPreemptible")]


def can_reclaim_cases():
    src = open(SRC).read()
    out = []
    for m in re.finditer(r'name:\s*"([^"]+)",\s*reclaimerInfo:\s*&ReclaimerInfo\{', src):
        line = src.count("\n", 0, m.start()) + 1
        p = G.Parser(src, m.end() - len("ReclaimerInfo{"))
        info = G._j(G.ev(p.parse_expr()))
        rest = src[p.peek()[2]:]
        q0 = rest.index("rs.QueueAttributes{")
        p2 = G.Parser(rest, q0)
        qa = G._j(G.ev(p2.parse_expr()))
        want = re.search(r"canReclaim:\s*(true|false)", rest[p2.peek()[2]:]).group(1) == "true"
        req = info["RequiredResources"]["args"]
        qrs = qa["QueueResourceShare"]
        out.append({"name": m.group(1), "line": line, "mode": "can_reclaim",
                    "queues": {"queue1": ["", {"cpu": share(qrs.get("CPU")), "memory": share(qrs.get("Memory")), "gpu": share(qrs.get("GPU"))}]},
                    "reclaimer": ["queue1", [float(x) for x in req], bool(info["IsPreemptable"])], "reclaimees": [], "want": want})
    return out


Z = [0.0] * 5


def gpu_queue(parent, deserved, fair, allocated, np_=0.0):  # the fixtures set GPU only: CPU / Memory are zero-valued ResourceShare{}
    return [parent, {"cpu": list(Z), "memory": list(Z), "gpu": [float(deserved), float(fair), UNL, float(allocated), float(np_)]}]


def spec(name, line, queues, reclaimer, reclaimees, want):
    return {"name": name, "line": line, "mode": "reclaimable", "queues": queues, "reclaimer": reclaimer, "reclaimees": reclaimees, "want": want}


def single_department():  # :533-683
    def base():
        return {"p1": gpu_queue("default", 3, 3, 2), "p2": gpu_queue("default", 2, 2, 3), "default": gpu_queue("", 5, 5, 5)}
    G1 = [0.0, 0.0, 1.0]
    out = []
    def add(name, line, edit, want, preemptible=True):
        q = base(); edit(q)
        out.append(spec(name, line, q, ["p1", G1, preemptible], [["p2", G1]], want))
    GPU = lambda q, n: q[n][1]["gpu"]
    add("Reclaimer is below fair share, reclaimee above fair share", 616, lambda q: None, True)
    def e(q): GPU(q, "p2")[3] = 2; GPU(q, "default")[3] = 4
    add("Reclaimer is below fair share, reclaimer exactly at fair share", 620, e, False)
    def e(q): GPU(q, "p2")[3] = 1; GPU(q, "default")[3] = 3
    add("Reclaimer and reclaimee are below fair share", 626, e, False)
    def e(q): GPU(q, "p2")[1] += 3 - GPU(q, "p2")[1]
    add("Reclaimer below deserved and reclaimee above deserved (within fair share)", 632, e, True)
    def e(q): GPU(q, "p1")[3] = 3; GPU(q, "default")[3] = 6; GPU(q, "default")[0] = 7; GPU(q, "default")[1] += 7 - GPU(q, "p2")[1]
    add("Reclaimer at fair share, reclaimee above fair share, department below fair share", 638, e, False)
    def e(q): GPU(q, "p1")[3] = 3; GPU(q, "default")[3] = 6
    add("Reclaimer at fair share, reclaimee above fair share, department above fair share", 646, e, False)
    def e(q): GPU(q, "p1")[0] = 2
    add("Reclaimer above deserved, attempting to reclaim for non preemptible job", 652, e, True, preemptible=False)
    def e(q): GPU(q, "p1")[0] = 2; GPU(q, "default")[0] = 3
    add("Reclaimer department only preemptible above deserved, attempting to reclaim for non preemptible job", 658, e, True, preemptible=False)
    def e(q): GPU(q, "p1")[0] = 2; GPU(q, "default")[0] = 3; GPU(q, "default")[4] = 3
    add("Reclaimer department nonpreemtible equal to deserved, attempting to reclaim for non preemptible job", 665, e, False, preemptible=False)
    def e(q): GPU(q, "p1")[0] = 2; GPU(q, "default")[0] = 1; GPU(q, "default")[1] = 1; GPU(q, "default")[3] = 3
    add("Reclaimer department allocated above fair share, attempting to reclaim for job", 673, e, True)
    return out


def multiple_departments():  # :685-797
    def base():
        return {"p1": gpu_queue("d1", 3, 3, 2), "p2": gpu_queue("d2", 2, 2, 3), "d1": gpu_queue("", 3, 3, 2), "d2": gpu_queue("", 2, 2, 3)}
    G1 = [0.0, 0.0, 1.0]
    GPU = lambda q, n: q[n][1]["gpu"]
    out = []
    q = base(); out.append(spec("Reclaimer is below fair share, reclaimee above fair share - sanity", 779, q, ["p1", G1, True], [["p2", G1]], True))
    q = base(); GPU(q, "p2")[3] = 2; GPU(q, "d2")[3] = 2
    out.append(spec("Reclaimee department goes below fair share", 783, q, ["p1", G1, True], [["p2", G1]], False))
    q = base(); GPU(q, "p1")[3] = 1; GPU(q, "d1")[3] = 1; GPU(q, "p2")[1] = 4; GPU(q, "d2")[1] = 4
    out.append(spec("Reclaimer department is below deserved and reclaimee department is above deserved but within fair share", 789, q, ["p1", G1, True], [["p2", G1]], True))
    return out


def multiple_levels():  # :799-1163 — queuesTestData {parent, deserved, fairShare, allocated} through buildQueues (:1165-1185)
    def build(d): return {n: gpu_queue(*v) for n, v in d.items()}
    G = lambda g: [0.0, 0.0, float(g)]
    out = []
    out.append(spec("Reclaimer is below fair share, reclaimee above fair share - sanity", 830,
                    build({"left-top": ("", 1, 1, 0), "left-mid": ("left-top", 1, 1, 0), "left-leaf": ("left-mid", 1, 1, 0),
                           "right-top": ("", 1, 1, 2), "right-mid": ("right-top", 1, 1, 2), "right-leaf": ("right-mid", 1, 1, 2)}),
                    ["left-leaf", G(1), True], [["right-leaf", G(2)]], True))
    out.append(spec("Reclaimer top queue will go over quota - don't reclaim", 876,
                    build({"left-top": ("", 1, 1, 1), "left-top-oq-leaf": ("left-top", 0, 0, 1), "left-mid": ("left-top", 1, 1, 0), "left-leaf": ("left-mid", 1, 1, 0),
                           "right-top": ("", 1, 1, 2), "right-mid": ("right-top", 1, 1, 2), "right-leaf": ("right-mid", 1, 1, 2)}),
                    ["left-leaf", G(1), True], [["right-leaf", G(2)]], False))
    branch = {"top": ("", 2, 2, 2), "mid1": ("top", 1, 1, 0.5), "mid2": ("top", 1, 1, 1.5), "left-leaf1": ("mid1", 1, 1, 0), "left-leaf2": ("mid1", 0, 0, 0.5), "right-leaf": ("mid2", 1, 1, 1.5)}
    out.append(spec("Reclaimer in the same tree branch and will go over fair share - don't reclaim", 928, build(branch), ["left-leaf1", G(1), True], [["right-leaf", G(1.5)]], False))
    out.append(spec("Reclaimer in the same tree branch - reclaim", 979, build(branch), ["left-leaf1", G(1), True], [["right-leaf", G(1.5)], ["left-leaf2", G(0.5)]], True))
    out.append(spec("Reclaimer has lower utilization ratio than reclaimee but over 1", 1045,
                    build({"d1": ("", 4, 4, 4), "d1-project-1": ("d1", 3, 1, 0), "d1-project-2": ("d1", 1, 3, 4), "d2": ("", 3, 3, 7), "d2-project-1": ("d2", 3, 3, 7)}),
                    ["d1-project-1", G(1), True], [["d2-project-1", G(1)]], True))
    q = build({"d1": ("", 4, 4, 4), "d1-project-1": ("d1", 1, 1, 1), "d2": ("", 3, 3, 7), "d2-project-1": ("d2", 3, 3, 7)})
    q["d1"][1]["cpu"][3] = 3000.0; q["d1"][1]["cpu"][1] = 1000.0; q["d2"][1]["cpu"][3] = 1000.0; q["d2"][1]["cpu"][1] = 1000.0
    out.append(spec("Reclamation with uninvolved resources", 1104, q, ["d1-project-1", G(1), True], [["d2-project-1", G(1)]], True))
    return out


def main():
    cases = can_reclaim_cases() + single_department() + multiple_departments() + multiple_levels()
    json.dump({"source": "plugins/proportion/reclaimable/reclaimable_test.go", "saturation_multiplier": 1.0, "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases →", OUT)


if __name__ == "__main__":
    main()
