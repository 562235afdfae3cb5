"""Pins the benched sizes against the oracle END TO END on the CPU: BASELINE config 3 and config 5 at full size, the oracle
(node scoring on 8 threads, same results as one thread: tests/test_oracle_threads.py) against the host-compiled engine (tests/host_sim = the
device engine's source under g++) — every committed operation, pod state, node and queue share — and writes the SHA-256 of the operation
stream to profiles/full_size_pins.json.  bench.py prints the same hash of the MI355X's operations (`ops_sha256`) and compares it with this file.

    python tools/pin_full_sizes.py C3 C5        (C5: about 10-20 minutes of oracle time on 8 cores)
"""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import kai_testlib as T
from test_engine_hostsim import HostSim

CONFIGS = {"C1": 0, "C2": 1, "C3": 2, "C5": 4, "C5mixed": 4, "C3fractions30": 2}  # C5mixed: the shape SURVEY 8d writes down (bench.py --mixed); C3fractions30: bench.py --config C3 --fractions 0.3
OUT = os.path.join(T.ROOT, "profiles", "full_size_pins.json")
pins = json.load(open(OUT)) if os.path.exists(OUT) else {}
for name in sys.argv[1:] or ["C3"]:
    snap, cfg, desc = T.pkg.synth.config(CONFIGS[name], 1.0, mixed=(name == "C5mixed"))
    if name == "C3fractions30":
        T.pkg.synth.add_fractions(snap, 7, frac=0.3); desc += " + 30 % of the one-GPU pods as fractions of one device"
    same_f = (lambda a, b: np.allclose(a, b, rtol=0.0, atol=1e-9)) if name == "C3fractions30" else np.array_equal  # fractions: shares to 1e-9 (the order of addition of non-integers)
    t0 = time.time(); ref = T.Oracle.run(snap, cfg, ("allocate",), threads=min(8, os.cpu_count() or 1)); t_or = time.time() - t0
    c3 = T.abi.KaiConfig.from_buffer_copy(cfg); c3.engine_mode = 3
    t0 = time.time(); res = HostSim.run(snap, c3, ("allocate",)); t_eng = time.time() - t0
    same = {"ops": res.ops == ref.ops, "stmts": res.stmts == ref.stmts, "pod_status": bool((res.pod_status == ref.pod_status).all()), "pod_node": bool((res.pod_node == ref.pod_node).all()),
            "nodes": all(np.array_equal(res.nodes[k], ref.nodes[k]) for k in ref.nodes), "shares_open": all(same_f(res.shares_open[k], ref.shares_open[k]) for k in ref.shares_open),
            "shares_final": all(same_f(res.shares_final[k], ref.shares_final[k]) for k in ref.shares_final), "decisions": int(res.stats.decisions) == int(ref.stats.decisions)}
    pins[name] = {"workload": desc, "nodes": snap.n_nodes, "pods": snap.n_pods, "jobs": snap.n_jobs, "queues": snap.n_queues, "actions": ["allocate"],
                  "ops": len(ref.ops), "decisions": int(ref.stats.decisions), "ops_sha256": T.ops_sha256(ref.ops), "state_sha256": T.state_sha256(ref),
                  "oracle_s": round(t_or, 1), "oracle_threads": min(8, os.cpu_count() or 1), "host_compiled_engine_s": round(t_eng, 2), "engine_equals_oracle": same}
    print(name, json.dumps(pins[name]), flush=True)
    json.dump(pins, open(OUT, "w"), indent=1, sort_keys=True)
    # (the decisions COUNTER may differ on snapshots with topology gangs: k_drain books one failing allocateTask per drained job, the reference none when subSetNodesFn finds no domain)
    assert all(v for k, v in same.items() if k != "decisions"), f"{name}: the host-compiled engine differs from the oracle: {same}"
