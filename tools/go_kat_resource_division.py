#!/usr/bin/env python3
"""Known answers of the proportion plugin's resource division → tests/golden/kat_resource_division.json (replaces the hand transcription of rounds 1-3).

Source: pkg/scheduler/plugins/proportion/resource_division/resource_division_test.go — a Ginkgo suite, not a table: Describe / Context / When containers with nested
BeforeEach blocks, It bodies that patch `queues["1"].GPU.MaxAllowed = 2`, call setResourceShare / divideOverQuotaResource and Expect results, one DescribeTable.  This
script is a small interpreter for exactly that dialect: it walks the container tree, replays the BeforeEach chain of every It (outermost first, as Ginkgo does),
executes the It's assignments, records the call and the expectations.  Queue literals go through the Go literal parser of tools/go_fixtures.py.  Only the reference is
read.  A case = one call: {fn, total, k_value, resource, queues (state before the call, ids ascending), remaining, fair (expected FairShare per queue, null where the
It does not look)}.  `created`: the suite stamps every queue with v1.Now() (equal or ascending in literal order); the index in the literal stands for it."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import go_fixtures as G  # noqa: E402

SRC = "/root/reference/pkg/scheduler/plugins/proportion/resource_division/resource_division_test.go"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kat_resource_division.json")
G.CONSTS["commonconstants.UnlimitedResourceQuantity"] = -1.0
RES = {"rs.GpuResource": "GPU", "rs.CpuResource": "CPU", "rs.MemoryResource": "Memory"}


def match(src, i):
    """index of the bracket that closes the one at src[i] (strings, runes and // comments skipped)"""
    open_c = src[i]; close_c = {"(": ")", "{": "}", "[": "]"}[open_c]; depth = 0
    while i < len(src):
        c = src[i]
        if c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
        elif c == "`":
            i = src.index("`", i + 1)
        elif src.startswith("//", i):
            i = src.index("\n", i)
        elif c == open_c:
            depth += 1
        elif c == close_c:
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ValueError("unbalanced")


def line_of(src, i):
    return src.count("\n", 0, i) + 1


class Scope:
    def __init__(self, parent=None):
        self.parent, self.vars, self.before = parent, {}, []

    def chain(self):
        return (self.parent.chain() if self.parent else []) + [self]

    def lookup(self, name):
        s = self
        while s:
            if name in s.vars:
                return s.vars[name]
            s = s.parent
        raise KeyError(name)


def value(expr, env, scope):
    expr = expr.strip()
    m = re.fullmatch(r"float64\((.+)\)", expr)
    if m:
        return float(value(m.group(1), env, scope))
    if re.fullmatch(r"-?\d+(\.\d+)?", expr):
        return float(expr)
    if expr in G.CONSTS:
        return float(G.CONSTS[expr])
    m = re.fullmatch(r'queues\["(\w+)"\]\.(GPU|CPU|Memory)\.(\w+)', expr)
    if m:
        return float(env["queues"][m.group(1)][m.group(2)].get(m.group(3), 0))
    if expr in env:
        return env[expr]
    return scope.lookup(expr)


def share_of(q, res):
    d = (q.get("QueueResourceShare") or {}).get(res)
    if not isinstance(d, dict) or d.get("_call"):  # rs.EmptyResource()
        d = {}
    return {k: float(v) for k, v in d.items() if not k.startswith("_")}


def queues_from_literal(lit):
    out = {}
    for i, (qid, q) in enumerate((k, v) for k, v in lit.items() if not k.startswith("_")):
        out[qid] = {"GPU": share_of(q, "GPU"), "CPU": share_of(q, "CPU"), "Memory": share_of(q, "Memory"), "Priority": int(q.get("Priority", 0)), "created": i}
    return out


def parse_literal(src, i):
    p = G.Parser(src, i)
    return G._j(G.ev(p.parse_expr()))


def statements(src, lo, hi):
    """top-level statements of src[lo:hi] as (start, end) spans: a statement ends at a newline outside every bracket"""
    out, i = [], lo
    while i < hi:
        while i < hi and src[i] in " \t\n":
            i += 1
        if i >= hi:
            break
        if src.startswith("//", i):
            i = src.index("\n", i); continue
        j = i
        while j < hi and src[j] != "\n":
            if src[j] in "({[":
                j = match(src, j)
            elif src[j] == '"':
                j += 1
                while src[j] != '"':
                    j += 2 if src[j] == "\\" else 1
            elif src.startswith("//", j):
                break
            j += 1
        out.append((i, j)); i = j + 1 if j < hi and src[j] == "\n" else src.index("\n", j) + 1
    return out


CASES = []


def run_body(src, lo, hi, env, scope, it=None):
    """assignments, calls and expectations of a BeforeEach / It body"""
    for a, b in statements(src, lo, hi):
        st = src[a:b].strip()
        m = re.match(r"queues\s*:?=\s*(map\[)", st)
        if m:
            env["queues"] = queues_from_literal(parse_literal(src, a + m.start(1))); continue
        m = re.match(r"queues\s*:=\s*getQueues\(\)", st)
        if m:
            env["queues"] = json.loads(json.dumps(scope.lookup("getQueues"))); continue
        m = re.match(r'queues\["(\w+)"\]\.(GPU|CPU|Memory)\.(\w+)\s*=\s*(.+)$', st)
        if m:
            env["queues"][m.group(1)][m.group(2)][m.group(3)] = value(m.group(4), env, scope); continue
        m = re.match(r"(\w+)\s*:=\s*map\[", st)
        if m:
            env[m.group(1)] = {k: float(v) for k, v in parse_literal(src, a + st.index("map[")).items() if not k.startswith("_")}; continue
        m = re.match(r"remaining\s*:=\s*(setResourceShare|divideOverQuotaResource)\((.+)\)$", st)
        if m:
            args = [x.strip() for x in m.group(2).split(",")]
            fn = m.group(1)
            total, k = value(args[0], env, scope), value(args[1], env, scope)
            res = RES[args[2] if fn == "setResourceShare" else args[3]]
            env["_call"] = {"fn": fn, "total": total, "k_value": k, "resource": res, "before": json.loads(json.dumps(env["queues"])), "fair": {}, "remaining": None}
            continue
        m = re.match(r"(\w+)\s*:=\s*(.+)$", st)
        if m and not st.startswith("remaining"):
            env[m.group(1)] = value(m.group(2), env, scope); continue
        m = re.match(r"Expect\(remaining\)\.To\(Equal\(([^,]+?)\)[,)]", st)
        if m:
            env["_call"]["remaining"] = value(re.sub(r"\)+$", "", m.group(1)) + (")" if m.group(1).count("(") > re.sub(r"\)+$", "", m.group(1)).count(")") else ""), env, scope); continue
        m = re.match(r'Expect\(queues\["(\w+)"\]\.(GPU|CPU|Memory)\.FairShare\)\.To\(Equal\((.+?)\)\)\s*(//.*)?$', st)
        if m:
            env["_call"]["fair"][m.group(1)] = value(m.group(3), env, scope); continue
        m = re.match(r"for (\w+), (\w+) := range (\w+) \{", st)
        if m and "Expect(queues[" in st:  # for uuid, expectedShare := range expectedShare { Expect(queues[uuid].GPU.FairShare).To(Equal(expectedShare), ...) }
            for qid, v in env[m.group(3)].items():
                env["_call"]["fair"][qid] = float(v)
            continue
        raise SystemExit(f"resource_division_test.go:{line_of(src, a)}: statement not understood: {st[:100]}")


def emit(name, line, env):
    c = env.pop("_call")
    ids = sorted(c["before"], key=lambda x: c["before"][x]["created"])
    r = c["resource"]
    qs = []
    for qid in ids:
        q = c["before"][qid]; s = q[r]
        num = lambda v: int(v) if float(v) == int(v) else float(v)
        qs.append({"deserved": num(s.get("Deserved", 0)), "request": num(s.get("Request", 0)), "fair": num(s.get("FairShare", 0)), "oqw": num(s.get("OverQuotaWeight", 0)),
                   "max_allowed": num(s.get("MaxAllowed", 0)), "priority": q["Priority"], "created": q["created"]})
    num = lambda v: None if v is None else (int(v) if float(v) == int(v) else float(v))
    CASES.append({"line": line, "name": name, "fn": c["fn"], "resource": r, "total": num(c["total"]), "k_value": float(c["k_value"]), "queues": qs,
                  "remaining": num(c["remaining"]), "fair": [num(c["fair"].get(qid)) for qid in ids]})


def run_it(src, name, line, lo, hi, scope):
    env = {}
    for s in scope.chain():
        for blo, bhi in s.before:
            run_body(src, blo, bhi, env, s)
    run_body(src, lo, hi, env, scope)
    emit(name, line, env)


def walk(src, lo, hi, scope):
    for a, b in statements(src, lo, hi):
        st = src[a:b]
        m = re.match(r'(Describe|Context|When)\("((?:[^"\\]|\\.)*)",\s*func\(\)\s*\{', st)
        if m:
            body = a + m.end() - 1
            walk(src, body + 1, match(src, body), Scope(scope)); continue
        m = re.match(r"BeforeEach\(func\(\)\s*\{", st)
        if m:
            body = a + m.end() - 1
            scope.before.append((body + 1, match(src, body))); continue
        m = re.match(r'It\("((?:[^"\\]|\\.)*)",\s*func\(\)\s*\{', st)
        if m:
            body = a + m.end() - 1
            run_it(src, m.group(1), line_of(src, a), body + 1, match(src, body), scope); continue
        m = re.match(r"var\s*\(", st)
        if m:
            for ln in src[a + m.end():match(src, a + m.end() - 1)].split("\n"):
                mm = re.match(r"\s*(\w+)\s+[\w.\[\]*]+\s*=\s*(.+)$", ln)
                if mm:
                    scope.vars[mm.group(1)] = value(mm.group(2), {}, scope)
            continue
        m = re.match(r"getQueues\s*:=\s*func\(\)[^{]*\{", st)
        if m:
            scope.vars["getQueues"] = queues_from_literal(parse_literal(src, src.index("map[", src.index("return", a))))
            continue
        m = re.match(r'DescribeTable\("((?:[^"\\]|\\.)*)",\s*func\(testData testMetadata\)\s*\{', st)
        if m:
            # the table's body (:274-293) patches MaxAllowed / OverQuotaWeight / Request / Priority from the entry's maps, calls setResourceShare(totalGPUs, 0, GPU) and expects
            # expectedRemaining and expectedShare — written out here, the entries are literals
            end = match(src, a + len("DescribeTable"))
            for em in re.finditer(r'Entry\("((?:[^"\\]|\\.)*)",\s*(testMetadata\{)', src[a:end]):
                lit = G._j(G.ev(G.Parser(src, a + em.end(2) - 1).parse_composite("testMetadata")))
                env = {"queues": json.loads(json.dumps(scope.lookup("getQueues")))}
                for field, key in (("maxAllowed", "MaxAllowed"), ("gpuOverQuotaWeights", "OverQuotaWeight"), ("request", "Request")):
                    for qid, v in (lit.get(field) or {}).items():
                        if not qid.startswith("_"):
                            env["queues"][qid]["GPU"][key] = float(v)
                for qid, v in (lit.get("overQuotaPriority") or {}).items():
                    if not qid.startswith("_"):
                        env["queues"][qid]["Priority"] = int(v)
                env["_call"] = {"fn": "setResourceShare", "total": float(lit.get("totalGPUs", 0)), "k_value": 0.0, "resource": "GPU", "before": json.loads(json.dumps(env["queues"])),
                                "remaining": float(lit.get("expectedRemaining", 0)), "fair": {k: float(v) for k, v in (lit.get("expectedShare") or {}).items() if not k.startswith("_")}}
                emit(em.group(1), line_of(src, a + em.start()), env)
            continue
        # type declarations and anything else at container level carry no case


def main(out=OUT):
    src = open(SRC).read()
    m = re.search(r'var _ = Describe\("Proportion", func\(\)\s*\{', src)
    body = m.end() - 1
    walk(src, body + 1, match(src, body), Scope())
    doc = {"source": "pkg/scheduler/plugins/proportion/resource_division/resource_division_test.go (Ginkgo: setResourceShare / divideOverQuotaResource per resource); `line` = the It / Entry line; "
                     "generated by tools/go_kat_resource_division.py (an interpreter for the suite's BeforeEach / It dialect)",
           "cases": CASES}
    json.dump(doc, open(out, "w"), indent=1)
    print(len(CASES), "cases →", out)


if __name__ == "__main__":
    main(*sys.argv[1:2])
