"""Broad randomized campaign on the CPU: seeds lo..hi of tests/kai_testlib.py::broad_case, oracle vs the host-compiled engine (tests/host_sim)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import kai_testlib as T
from test_engine_hostsim import HostSim
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = tot = 0; t0 = time.time()
for seed in range(lo, hi):
    for ci, (snap, cfg, acts) in enumerate(T.broad_case(seed)):
        cfg.engine_mode = seed % 3 if ci % 2 else 0
        o = T.Oracle.run(snap, cfg, acts); tot += 1
        try: g = HostSim.run(snap, cfg, acts)
        except RuntimeError as e:
            bad += 1; print("ENGINE ERROR seed", seed, "case", ci, acts, "mode", cfg.engine_mode, e, flush=True); continue
        ok = o.ops == g.ops and (o.pod_status == g.pod_status).all() and (o.pod_node == g.pod_node).all() and all(np.array_equal(o.nodes[k], g.nodes[k]) for k in o.nodes) \
            and all(np.array_equal(o.shares_final[k], g.shares_final[k]) for k in o.shares_final)
        if not ok:
            bad += 1; print("MISMATCH seed", seed, "case", ci, acts, "mode", cfg.engine_mode, flush=True)
    if time.time() - t0 > float(os.environ.get("CAMPAIGN_SECONDS", "150")):
        print("time budget reached at seed", seed); break
print("runs", tot, "mismatch", bad, f"{time.time()-t0:.0f}s")
