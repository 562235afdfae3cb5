"""BASELINE config 4 at a given scale with queueDepthPerAction for the victim actions: the ORACLE end to end (8 threads) against the host-compiled engine, operations hashed
into profiles/full_size_pins.json (what bench.py's parity_full compares the MI355X's run with).  usage: pin_c4_depth.py <scale> <depth> [--no-oracle]"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import kai_testlib as T
from test_engine_hostsim import HostSim
scale, depth = float(sys.argv[1]), int(sys.argv[2])
acts = ("allocate", "consolidation", "reclaim")
snap, cfg, desc = T.pkg.synth.config(3, scale)
for a in ("consolidation", "reclaim", "preempt"):
    cfg.queue_depth[T.abi.ACTIONS[a]] = depth
desc += f", queueDepthPerAction {depth} for the victim actions"
t = time.time(); r = HostSim.run(snap, cfg, acts); th = time.time() - t
sha = T.ops_sha256(r.ops)
print(desc, "host-compiled engine %.1f s" % th, "ops", len(r.ops), "sha", sha, flush=True)
entry = {"actions": list(acts), "workload": desc, "nodes": snap.n_nodes, "pods": snap.n_pods, "jobs": snap.n_jobs, "queues": snap.n_queues, "ops": len(r.ops),
         "host_compiled_engine_s": round(th, 1), "host_engine_ops_sha256": sha}
if "--no-oracle" not in sys.argv:
    t = time.time(); o = T.Oracle.run(snap, cfg, acts, threads=8); to = time.time() - t
    osha = T.ops_sha256(o.ops)
    print("oracle %.1f s" % to, "ops", len(o.ops), "sha", osha, "equal", osha == sha, flush=True)
    entry.update({"ops_sha256": osha, "oracle_s": round(to, 1), "oracle_threads": 8, "engine_equals_oracle": str({"ops": osha == sha, "pod_status": bool((o.pod_status == r.pod_status).all()), "pod_node": bool((o.pod_node == r.pod_node).all())})})
    path = os.path.join(ROOT, "profiles", "full_size_pins.json")
    d = json.load(open(path)); d["C4_%gpct_depth%d" % (scale * 100, depth)] = entry
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)
