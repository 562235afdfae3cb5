"""Broad randomized campaign on the CPU: seeds lo..hi of tests/kai_testlib.py::broad_case, oracle vs the host-compiled engine (tests/host_sim).
CAMPAIGN_FRACTIONS=1 turns a share of the one-GPU pods of every case into fraction pods on shared GPUs (synth.add_fractions, portions in 1/4 so that
quota sums stay exact); queue shares are then held to 1e-9 instead of bit for bit (the order of addition of non-integers)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import kai_testlib as T
from test_engine_hostsim import HostSim
lo, hi = int(sys.argv[1]), int(sys.argv[2])
FRAC = os.environ.get("CAMPAIGN_FRACTIONS", "0") == "1"
MW = int(os.environ.get("CAMPAIGN_MW", "1"))  # > 1: the victim actions of the host-compiled engine on that many engines (threads over replicas, kai_engine_solver.inc solve_partial_multi)
HostSim.lib(); HostSim._raw.kai_hostsim_set_multi(MW)
same = (lambda a, b: np.allclose(a, b, rtol=0.0, atol=1e-9)) if FRAC else np.array_equal
bad = tot = 0; t0 = time.time()
for seed in range(lo, hi):
    for ci, (snap, cfg, acts) in enumerate(T.broad_case(seed)):
        cfg.engine_mode = seed % 3 if ci % 2 else 0
        if FRAC: snap, cfg, acts = T.broad_case(seed)[ci]; cfg.engine_mode = seed % 3 if ci % 2 else 0  # cases of one seed share snapshot objects: take a fresh one
        if FRAC: T.pkg.synth.add_fractions(snap, seed * 7 + ci, frac=(0.3, 0.6, 0.9)[(seed + ci) % 3], portions=((0.25, 0.5, 0.75), (0.5,), (0.25, 0.25, 0.5))[seed % 3])
        o = T.Oracle.run(snap, cfg, acts); tot += 1
        try: g = HostSim.run(snap, cfg, acts)
        except RuntimeError as e:
            bad += 1; print("ENGINE ERROR seed", seed, "case", ci, acts, "mode", cfg.engine_mode, e, flush=True); continue
        ok = o.ops == g.ops and (o.pod_status == g.pod_status).all() and (o.pod_node == g.pod_node).all() and all(np.array_equal(o.nodes[k], g.nodes[k]) for k in o.nodes) \
            and all(same(o.shares_final[k], g.shares_final[k]) for k in o.shares_final)
        if not ok:
            bad += 1; print("MISMATCH seed", seed, "case", ci, acts, "mode", cfg.engine_mode, flush=True)
    if time.time() - t0 > float(os.environ.get("CAMPAIGN_SECONDS", "150")):
        print("time budget reached at seed", seed); break
print("runs", tot, "mismatch", bad, f"{time.time()-t0:.0f}s")
