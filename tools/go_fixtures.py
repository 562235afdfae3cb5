#!/usr/bin/env python3
"""Transcribe the reference's golden test tables into JSON fixtures.

The reference cannot be compiled here (no Go toolchain), so its declarative action tests are the
golden vectors that pin the oracle.  Every table is a Go composite literal of
`test_utils.TestTopologyBasic` (pkg/scheduler/test_utils/test_utils.go:40-58).  This script parses
those literals straight out of the Go sources under /root/reference and writes tests/golden/*.json.
It only reads the reference; fixtures + this script are committed, the Go sources are not copied.

Usage:  python tools/go_fixtures.py            (regenerates tests/golden/)
"""
from __future__ import annotations

import json
import os
import re
import sys

REF = "/root/reference/pkg/scheduler"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# (relative Go file, action list the test runs, rounds).  Unit action tests run ONE action once
# (e.g. actions/allocate/allocate_test.go:49-53); integration tests run all actions for N rounds
# (actions/integration_tests/integration_tests_utils/integration_tests_utils.go:40-59).
SOURCES = [
    ("actions/allocate/allocate_test.go", ["allocate"]),
    ("actions/allocate/allocateGang_test.go", ["allocate"]),
    ("actions/allocate/allocateElastic_test.go", ["allocate"]),
    ("actions/allocate/allocate_subgroups_test.go", ["allocate"]),
    ("actions/allocate/allocateTopology_test.go", ["allocate"]),
    ("actions/allocate/allocateFractionalGpu_test.go", ["allocate"]),
    ("actions/allocate/allocateGpuMemory_test.go", ["allocate"]),
    ("actions/allocate/allocateMIG_test.go", ["allocate"]),
    ("actions/reclaim/reclaim_test.go", ["reclaim"]),
    ("actions/reclaim/reclaimDepartments_test.go", ["reclaim"]),
    ("actions/reclaim/reclaimGang_test.go", ["reclaim"]),
    ("actions/reclaim/reclaim_elastic_test.go", ["reclaim"]),
    ("actions/reclaim/reclaim_sub_group_test.go", ["reclaim"]),
    ("actions/preempt/preempt_test.go", ["preempt"]),
    ("actions/preempt/preemptGang_test.go", ["preempt"]),
    ("actions/preempt/preempt_elastic_test.go", ["preempt"]),
    ("actions/preempt/preempt_subgroups_test.go", ["preempt"]),
    ("actions/consolidation/consolidation_test.go", ["consolidation"]),
    ("actions/consolidation/consolidation_subgroups_test.go", ["consolidation"]),
    ("actions/reclaim/reclaimGpuMemory_test.go", ["reclaim"]),
    ("actions/reclaim/reclaimMIG_test.go", ["reclaim"]),
    ("actions/preempt/preemptGpuMemory_test.go", ["preempt"]),
    ("actions/preempt/preemptMIG_test.go", ["preempt"]),
    ("actions/consolidation/consolidationGpuMemory_test.go", ["consolidation"]),
    ("actions/integration_tests/allocate/allocate_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/allocate/allocate_topology_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/reclaim/reclaim_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/preempt/preempt_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/preempt/preemptGang_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/consolidation/consolidation_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/consolidation/consolidationGang_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/consolidation_and_reclaim/consolidation_and_reclaim_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/allocate/allocateFractionalGpu_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/allocate/allocateMIG_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/consolidation/consolidationFractional_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/preempt/preemptFractional_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/preempt/preemptMIG_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/reclaim/reclaimFractional_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
    ("actions/integration_tests/reclaim/reclaimMIG_test.go", ["allocate", "consolidation", "reclaim", "preempt", "stalegangeviction"]),
]


# --------------------------------------------------------------------------------------------- lexer
TOKEN_RE = re.compile(
    r"""
    (?P<ws>\s+) |
    (?P<lc>//[^\n]*) |
    (?P<bc>/\*.*?\*/) |
    (?P<raw>`[^`]*`) |
    (?P<str>"(?:\\.|[^"\\])*") |
    (?P<rune>'(?:\\.|[^'\\])') |
    (?P<num>(?:0[xX][0-9a-fA-F_]+|(?:\d[\d_]*)?\.\d[\d_]*(?:[eE][+-]?\d+)?|\d[\d_]*(?:[eE][+-]?\d+)?)) |
    (?P<id>[A-Za-z_][A-Za-z0-9_]*) |
    (?P<op>:=|\.\.\.|&&|\|\||==|!=|<=|>=|[{}\[\]().,:;&*+\-/<>=!%|^])
    """,
    re.X | re.S,
)


def lex(src: str, start: int):
    pos = start
    n = len(src)
    while pos < n:
        m = TOKEN_RE.match(src, pos)
        if not m:
            raise SyntaxError(f"lex error at {pos}: {src[pos:pos+40]!r}")
        pos = m.end()
        k = m.lastgroup
        if k in ("ws", "lc", "bc"):
            continue
        yield (k, m.group(), m.start())


def unquote(tok: str) -> str:
    if tok[0] == "`":
        return tok[1:-1]
    return json.loads(tok.replace("\\'", "'")) if "\\x" not in tok else bytes(tok[1:-1], "utf-8").decode("unicode_escape")


# --------------------------------------------------------------------------------------------- parser
class Parser:
    """Parses one Go expression starting at a byte offset; stops when the expression is complete."""

    def __init__(self, src: str, start: int):
        self.src = src
        self.toks = lex(src, start)
        self.buf = []

    def peek(self, k=0):
        while len(self.buf) <= k:
            try:
                self.buf.append(next(self.toks))
            except StopIteration:
                self.buf.append(("eof", "", len(self.src)))
        return self.buf[k]

    def next(self):
        t = self.peek()
        self.buf.pop(0)
        return t

    def expect(self, val):
        t = self.next()
        if t[1] != val:
            raise SyntaxError(f"expected {val!r} got {t[1]!r} at {t[2]}: {self.src[max(0,t[2]-60):t[2]+40]!r}")
        return t

    # type expressions ----------------------------------------------------------------------------
    def parse_type(self):
        t = self.peek()
        if t[1] == "*":
            self.next()
            return self.parse_type()
        if t[1] == "[":
            self.next()
            if self.peek()[1] != "]":
                self.next()  # array length
            self.expect("]")
            return {"list": self.parse_type()}
        if t[1] == "map":
            self.next()
            self.expect("[")
            k = self.parse_type()
            self.expect("]")
            return {"map": (k, self.parse_type())}
        if t[1] == "func":
            self.next()
            self.expect("(")
            depth = 1
            while depth:
                x = self.next()[1]
                depth += x == "("
                depth -= x == ")"
            if self.peek()[1] not in ("{",):
                self.parse_type()
            return {"func": True}
        if t[1] == "interface":
            self.next(); self.expect("{"); self.expect("}")
            return "interface{}"
        if t[0] == "id":
            name = self.next()[1]
            while self.peek()[1] == "." and self.peek(1)[0] == "id":
                self.next()
                name += "." + self.next()[1]
            return name
        raise SyntaxError(f"type? {t}")

    # expressions ---------------------------------------------------------------------------------
    def parse_expr(self, elem_type=None):  # Go's precedence for the arithmetic the fixtures use: * / bind tighter than + -, both left-associative
        lhs = self.parse_term(elem_type)
        while self.peek()[1] in ("+", "-") and self.peek()[0] == "op":
            op = self.next()[1]
            rhs = self.parse_term(None)
            lhs = {"_bin": op, "l": lhs, "r": rhs}
        return lhs

    def parse_term(self, elem_type=None):
        lhs = self.parse_unary(elem_type)
        while self.peek()[1] in ("*", "/") and self.peek()[0] == "op":
            op = self.next()[1]
            rhs = self.parse_unary(None)
            lhs = {"_bin": op, "l": lhs, "r": rhs}
        return lhs

    def parse_unary(self, elem_type):
        t = self.peek()
        if t[1] == "-":
            self.next()
            return {"_neg": self.parse_unary(None)}
        if t[1] == "&":
            self.next()
            return self.parse_unary(elem_type)
        return self.parse_postfix(self.parse_primary(elem_type))

    def parse_primary(self, elem_type):
        t = self.peek()
        if t[1] == "{":  # elided element type inside a slice / map literal
            return self.parse_composite(elem_type)
        if t[0] == "num":
            self.next()
            s = t[1].replace("_", "")
            return float(s) if any(c in s for c in ".eE") and not s.lower().startswith("0x") else int(s, 0)
        if t[0] in ("str", "raw"):
            self.next()
            return unquote(t[1])
        if t[1] == "(":
            self.next()
            e = self.parse_expr()
            self.expect(")")
            return e
        if t[1] in ("[", "map") or (t[1] == "*" ):
            ty = self.parse_type()
            if self.peek()[1] == "{":
                return self.parse_composite(ty)
            if self.peek()[1] == "(":  # conversion like []string(x)
                self.next(); e = self.parse_expr(); self.expect(")")
                return e
            return {"_type_expr": ty}
        if t[1] == "func":
            return self.parse_func_literal()
        if t[0] == "id":
            name = self.next()[1]
            if name in ("true", "false"):
                return name == "true"
            if name == "nil":
                return None
            while self.peek()[1] == "." and self.peek(1)[0] == "id":
                self.next()
                name += "." + self.next()[1]
            if self.peek()[1] == "{" and (name[0].isupper() or "." in name):
                return self.parse_composite(name)
            return {"_ident": name}
        raise SyntaxError(f"primary? {t} :: {self.src[max(0,t[2]-80):t[2]+40]!r}")

    def parse_postfix(self, e):
        while True:
            t = self.peek()
            if t[1] == "(":
                self.next()
                args = []
                while self.peek()[1] != ")":
                    args.append(self.parse_expr())
                    if self.peek()[1] == ",":
                        self.next()
                self.expect(")")
                e = {"_call": e.get("_ident") if isinstance(e, dict) and "_ident" in e else e, "args": args}
            elif t[1] == "." and self.peek(1)[0] == "id":
                self.next()
                e = {"_sel": self.next()[1], "x": e}
            elif t[1] == "[":
                self.next(); idx = self.parse_expr(); self.expect("]")
                e = {"_index": idx, "x": e}
            else:
                return e

    def parse_composite(self, ty):
        self.expect("{")
        elem_t = None
        if isinstance(ty, dict) and "list" in ty:
            elem_t = ty["list"]
        if isinstance(ty, dict) and "map" in ty:
            elem_t = ty["map"][1]
        fields, elems, items = {}, [], []
        while self.peek()[1] != "}":
            e = self.parse_expr(elem_t)
            if self.peek()[1] == ":":
                self.next()
                v = self.parse_expr(elem_t)
                if isinstance(ty, dict) and "map" in ty:
                    items.append((e, v))
                elif isinstance(e, dict) and "_ident" in e:
                    fields[e["_ident"]] = v
                else:
                    items.append((e, v))
            else:
                elems.append(e)
            if self.peek()[1] == ",":
                self.next()
        self.expect("}")
        if isinstance(ty, dict) and "map" in ty:
            return {"_map": items}
        if isinstance(ty, dict) and "list" in ty:
            return {"_list": elems}
        return {"_struct": ty, "fields": fields, "elems": elems, "items": items}

    def parse_func_literal(self):
        """func() T { stmts }()  — only the sub-group-set builder idiom is interpreted."""
        self.parse_type()  # consumes 'func(...) T'
        self.expect("{")
        stmts = []
        while self.peek()[1] != "}":
            if self.peek()[1] == "return":
                self.next()
                stmts.append(("return", self.parse_expr()))
            elif self.peek()[0] == "id" and self.peek(1)[1] == ":=":
                name = self.next()[1]; self.next()
                stmts.append(("assign", name, self.parse_expr()))
            else:
                stmts.append(("expr", self.parse_expr()))
            if self.peek()[1] == ";":
                self.next()
        self.expect("}")
        e = {"_func": stmts}
        return e


# --------------------------------------------------------------------------------------------- evaluation
CONSTS = {
    "constants.PriorityTrainNumber": 50, "constants.PriorityBuildNumber": 100,
    "constants.PriorityInteractivePreemptibleNumber": 75, "constants.PriorityInferenceNumber": 125,
    "common_info.NoMaxAllowedResource": -1.0, "commonconstants.UnlimitedResourceQuantity": -1.0,
    "subgroup_info.RootSubGroupSetName": "", "podgroup_info.DefaultSubGroup": "default",
    "time.Second": 1_000_000_000, "time.Minute": 60_000_000_000, "time.Hour": 3_600_000_000_000, "time.Millisecond": 1_000_000,
    "enginev2alpha2.Preemptible": "preemptible", "enginev2alpha2.NonPreemptible": "non-preemptible",
    "v2alpha2.Preemptible": "preemptible", "v2alpha2.NonPreemptible": "non-preemptible",
    "node_info.MigStrategySingle": "single", "node_info.MigStrategyMixed": "mixed", "node_info.MigStrategyNone": "none",
    "node_info.DefaultGpuMemory": 100,
}
IDENTITY_CALLS = {"ptr.To", "pointer.Int", "pointer.Int32", "pointer.Int64", "pointer.Float64", "pointer.Duration", "pointer.Bool",
                  "test_utils.CreateFloat64Pointer", "CreateFloat64Pointer", "int64", "int32", "int", "float64", "uint64", "string",
                  "pointer.String", "v1.ResourceName", "common_info.PodID", "lo.ToPtr", "ptr.Int", "intPtr", "float64Ptr"}


class SubGroupSet:
    def __init__(self, name, constraint):
        self.name, self.constraint, self.groups, self.podsets = name, constraint, [], []

    def to_json(self):
        return {"Name": self.name, "TopologyConstraint": self.constraint,
                "PodSets": [{"Name": n, "MinAvailable": m, "TopologyConstraint": c} for (n, m, c) in self.podsets],
                "SubGroups": [g.to_json() for g in self.groups]}


def ev(e, env=None):
    env = env or {}
    if isinstance(e, (int, float, str, bool)) or e is None:
        return e
    if "_ident" in e:
        n = e["_ident"]
        if n in env:
            return env[n]
        if n in CONSTS:
            return CONSTS[n]
        if n.startswith("pod_status."):
            return n.split(".", 1)[1]
        return n
    if "_neg" in e:
        return -ev(e["_neg"], env)
    if "_bin" in e:
        l, r = ev(e["l"], env), ev(e["r"], env)
        return {"*": lambda: l * r, "/": lambda: l / r, "+": lambda: l + r, "-": lambda: l - r}[e["_bin"]]()
    if "_list" in e:
        return [ev(x, env) for x in e["_list"]]
    if "_map" in e:
        return {str(ev(k, env)): ev(v, env) for k, v in e["_map"]}
    if "_struct" in e:
        d = {k: ev(v, env) for k, v in e["fields"].items()}
        if e["elems"]:
            d["_elems"] = [ev(x, env) for x in e["elems"]]
        for k, v in e["items"]:
            d[str(ev(k, env))] = ev(v, env)
        d["_type"] = e["_struct"] if isinstance(e["_struct"], str) else None
        return d
    if "_func" in e:
        loc = dict(env)
        for st in e["_func"]:
            if st[0] == "assign":
                loc[st[1]] = ev(st[2], loc)
            elif st[0] == "expr":
                ev(st[1], loc)
            elif st[0] == "return":
                return ev(st[1], loc)
        return None
    if "_call" in e:
        f = e["_call"]
        args = [ev(a, env) for a in e["args"]]
        if isinstance(f, dict) and "_func" in f:  # immediately-invoked function literal
            return ev(f, env)
        if isinstance(f, dict) and "_sel" in f:   # method call: x.AddPodSet(...)
            recv = ev(f["x"], env)
            if f["_sel"] == "AddPodSet" and isinstance(recv, SubGroupSet):
                recv.podsets.append(args[0]); return None
            if f["_sel"] == "AddSubGroup" and isinstance(recv, SubGroupSet):
                recv.groups.append(args[0]); return None
            return {"_method": f["_sel"], "args": _j(args)}
        if isinstance(f, str) and "." in f and f.split(".", 1)[0] in env:  # method call on a local: root.AddPodSet(...)
            recv, meth = env[f.split(".", 1)[0]], f.split(".", 1)[1]
            if meth == "AddPodSet" and isinstance(recv, SubGroupSet):
                recv.podsets.append(args[0]); return None
            if meth == "AddSubGroup" and isinstance(recv, SubGroupSet):
                recv.groups.append(args[0]); return None
        if f in IDENTITY_CALLS:
            return args[0]
        if f == "subgroup_info.NewSubGroupSet":
            return SubGroupSet(args[0], args[1])
        if f == "subgroup_info.NewPodSet":
            return (args[0], args[1], args[2])
        if f == "jobs_fake.DefaultSubGroup":
            s = SubGroupSet("", None); s.podsets.append(("default", args[0], None)); return s
        if f in ("resource.MustParse",):
            return args[0]
        return {"_call": f if isinstance(f, str) else "?", "args": _j(args)}
    if "_sel" in e:
        return {"_sel": e["_sel"]}
    if "_type_expr" in e or "_index" in e:
        return None
    raise ValueError(f"cannot evaluate {e}")


def _j(x):
    if isinstance(x, SubGroupSet):
        return x.to_json()
    if isinstance(x, dict):
        return {k: _j(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_j(v) for v in x]
    return x


def extract(path: str):
    src = open(path).read()
    out = []
    for m in re.finditer(r"(?:test_utils\.)?TestTopologyBasic\{", src):
        # skip the field name `TestTopologyBasic:` matches — we start at the literal's type token
        start = m.start()
        if src[max(0, start - 30):start].rstrip().endswith(("func", "type")):
            continue
        line = src.count("\n", 0, start) + 1
        try:
            p = Parser(src, start)
            node = p.parse_expr()
            val = _j(ev(node))
        except Exception as ex:  # keep going; report
            out.append({"_error": f"{type(ex).__name__}: {ex}", "_line": line})
            continue
        if isinstance(val, dict) and ("Jobs" in val or "Nodes" in val):
            val["_line"] = line
            # RoundsUntilMatch / RoundsAfterMatch live on the wrapping TestTopologyMetadata literal (integration tests): the
            # fields that follow this literal up to the wrapper's closing brace
            end = p.peek()[2]
            tail = src[end:end + 400]
            nxt = tail.find("TestTopologyBasic")
            if nxt >= 0:
                tail = tail[:nxt]
            for fld in ("RoundsUntilMatch", "RoundsAfterMatch"):
                mm = re.search(fld + r":\s*(\d+)", tail)
                if mm:
                    val["_" + fld] = int(mm.group(1))
            out.append(val)
    # RoundsUntilMatch etc. live on the wrapping TestTopologyMetadata literal
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    summary = {}
    for rel, actions in SOURCES:
        path = os.path.join(REF, rel)
        if not os.path.exists(path):
            print("missing", rel); continue
        cases = extract(path)
        errs = [c for c in cases if "_error" in c]
        good = [c for c in cases if "_error" not in c]
        name = rel.replace("actions/", "").replace("/", "__").replace("_test.go", "")
        doc = {"source": f"pkg/scheduler/{rel}", "actions": actions, "cases": good, "parse_errors": errs}
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
        summary[name] = (len(good), len(errs))
        print(f"{rel}: {len(good)} cases, {len(errs)} parse errors")
        for e in errs[:3]:
            print("   ", e)
    return summary


if __name__ == "__main__":
    main()
