#!/usr/bin/env python3
"""Host clocks of kai_session_open's host preparation (SharedPods + HostPrep, kai_host_prep.hpp) on THIS machine's cores — no GPU involved.

tests/host_sim exports the preparation as kai_hostsim_prep_ms (objects kept between calls, as a kai_core handle keeps them between sessions); this tool runs it `--reps` times on a
BASELINE configuration and prints the best run: SharedPods, HostPrep, and HostPrep's eight phases (range checks / nodes / pods / task order / queues + job lists / shares + topology /
classes / batch shape).  KAI_HOST_THREADS sets the thread count (default: the machine's cores, at most 16), KAI_HOST_POOL=0 starts threads per loop instead of using the pool.

    KAI_HOST_THREADS=8 python tools/host_prep_timing.py --config C5 [--reps 7]
"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C5"); ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--reps", type=int, default=7); ap.add_argument("--mixed", action="store_true")
    args = ap.parse_args()
    import kai_testlib as T
    from test_engine_hostsim import HostSim
    HostSim.lib(); raw = HostSim._raw
    raw.kai_hostsim_prep_ms.restype = C.c_int
    snap, cfg, desc = T.pkg.synth.config({"C1": 0, "C2": 1, "C3": 2, "C4": 3, "C5": 4}[args.config], args.scale, mixed=args.mixed)
    st = snap.as_struct(); out = (C.c_double * 16)()
    best = None
    for _ in range(args.reps):
        rc = raw.kai_hostsim_prep_ms(C.byref(cfg), C.byref(st), out, 16)
        assert rc == 0, rc
        v = [float(x) for x in out[:10]]
        if best is None or v[0] + v[1] < best[0] + best[1]:
            best = v
    names = ("range checks", "nodes", "pods", "task order", "queues + job lists", "shares + topology", "classes", "batch shape")
    print(json.dumps({"workload": desc, "nodes": snap.n_nodes, "pods": snap.n_pods, "jobs": snap.n_jobs, "host_threads": os.environ.get("KAI_HOST_THREADS", "default"),
                      "pool": os.environ.get("KAI_HOST_POOL", "1") != "0", "shared_pods_ms": round(best[0], 2), "host_prep_ms": round(best[1], 2), "total_ms": round(best[0] + best[1], 2),
                      "phases_ms": {n: round(x, 2) for n, x in zip(names, best[2:])}, "reps": args.reps}))


if __name__ == "__main__":
    main()
