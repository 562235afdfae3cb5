"""Debug aid: one seed of the broad randomized campaign on the GPU against the oracle (prints the first differing operation per case)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import kai_testlib as T
from test_gpu_parity import run_gpu
seed = int(sys.argv[1])
for ci, (snap, cfg, acts) in enumerate(T.broad_case(seed)):
    for mode in (0, 1, 2):
        cfg.engine_mode = mode
        o = T.Oracle.run(snap, cfg, acts); g = run_gpu(snap, cfg, acts)
        same = o.ops == g.ops and (o.pod_status == g.pod_status).all() and (o.pod_node == g.pod_node).all()
        d = next((i for i, (a, b) in enumerate(zip(o.ops, g.ops)) if a != b), None)
        print("case", ci, acts, "mode", mode, "nodes", snap.n_nodes, "pods", snap.n_pods, "same", same, "nops", len(o.ops), len(g.ops), "first diff", d, (o.ops[d], g.ops[d]) if d is not None else "")
    cfg.engine_mode = 0
