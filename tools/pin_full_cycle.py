"""BASELINE config 5 at a small scale under the reference's DEFAULT action list minus stalegangeviction (conf_util/scheduler_conf_util.go:37: allocate, consolidation, reclaim, preempt) with
queueDepthPerAction for the victim actions: the ORACLE end to end against the host-compiled engine, operations hashed into profiles/full_size_pins.json.
usage: pin_full_cycle.py <scale> <depth>     (0.005 8: 328 nodes x 5 000 pods — the largest size at which every implementation here finishes the four actions in seconds)"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import kai_testlib as T
from test_engine_hostsim import HostSim
scale, depth = float(sys.argv[1]), int(sys.argv[2])
acts = ("allocate", "consolidation", "reclaim", "preempt")
snap, cfg, desc = T.pkg.synth.config(4, scale)
for a in ("consolidation", "reclaim", "preempt"):
    cfg.queue_depth[T.abi.ACTIONS[a]] = depth
desc += f", queueDepthPerAction {depth} for the victim actions"  # (the string bench.py builds for --queue-depth: its parity_full looks the pin up by workload + actions)
t = time.time(); r = HostSim.run(snap, cfg, acts); th = time.time() - t
t = time.time(); o = T.Oracle.run(snap, cfg, acts, threads=8); to = time.time() - t
sha, osha = T.ops_sha256(r.ops), T.ops_sha256(o.ops)
print(desc, "host-compiled engine %.1f s, oracle %.1f s, ops %d, equal %s" % (th, to, len(o.ops), sha == osha), flush=True)
entry = {"actions": list(acts), "workload": desc, "nodes": snap.n_nodes, "pods": snap.n_pods, "jobs": snap.n_jobs, "queues": snap.n_queues, "ops": len(o.ops), "evictions": sum(1 for x in o.ops if x[0] == 2),
         "host_compiled_engine_s": round(th, 1), "ops_sha256": osha, "oracle_s": round(to, 1), "oracle_threads": 8, "scale": scale, "queue_depth": depth, "note": "the reference's default action list without stalegangeviction (conf_util/scheduler_conf_util.go:37) on one session",
         "engine_equals_oracle": str({"ops": osha == sha, "pod_status": bool((o.pod_status == r.pod_status).all()), "pod_node": bool((o.pod_node == r.pod_node).all())})}
path = os.path.join(ROOT, "profiles", "full_size_pins.json")
d = json.load(open(path)); d["C5_%gpct_cycle_depth%d" % (scale * 100, depth)] = entry
json.dump(d, open(path, "w"), indent=1, sort_keys=True)
