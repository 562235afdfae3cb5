#!/usr/bin/env python3
"""How fast is the bucket-fill ALGORITHM on one host core?  (`cpu_same_algorithm`.)

tests/host_sim/native_bucket_fill.hpp is the fill of the batch path — sets of nodes by free devices with two summary levels, whole nodes of a one-class gang per step, dead
gangs decided from the levels' populations — as plain scalar C++.  With KAI_HOSTSIM_NATIVE_FILL=1 the host simulation runs it as a shadow of every emulated fill launch of the
allocate action (same plan, same sets), compares every output with the emulated kernel's and sums its time; the plan and apply kernels still run on the emulator (they are
data-parallel kernels, not a loop a core would run this way), so the figure is the FILL alone — the part of the cycle that is one dependency chain on the MI355X.

    python tools/native_fill_timing.py [--config C5] [--scale 1.0] [--out profiles/r05_native_fill.json]
"""
import argparse, ctypes as C, json, os, platform, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["KAI_HOSTSIM_NATIVE_FILL"] = "1"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def measure(idx, scale):
    import kai_testlib as T
    from test_engine_hostsim import HostSim
    snap, cfg, desc = T.pkg.synth.config(idx, scale)
    HostSim.lib()
    ms, a, b, d = C.c_double(), C.c_int64(), C.c_int64(), C.c_int64()
    HostSim._raw.kai_hostsim_native_fill(C.byref(ms), C.byref(a), C.byref(b), C.byref(d))  # (clears the sums)
    t0 = time.time(); res = HostSim.run(snap, cfg, ("allocate",)); wall = time.time() - t0
    HostSim._raw.kai_hostsim_native_fill(C.byref(ms), C.byref(a), C.byref(b), C.byref(d))
    return {"workload": desc, "nodes": snap.n_nodes, "pods": snap.n_pods, "native_fill_ms": ms.value, "fill_launches": a.value, "fill_decisions": b.value,
            "ns_per_decision": ms.value * 1e6 / max(b.value, 1), "outputs_differing_from_the_emulated_kernel": d.value, "operations": len(res.ops), "ops_sha256": T.ops_sha256(res.ops),
            "on_sets_by_free_devices": int(res.stats.reserved[6]), "emulated_cycle_wall_s": round(wall, 1), "cpu": cpu_model(), "threads": 1,
            "what": "tests/host_sim/native_bucket_fill.hpp: the fill of the allocate action as scalar C++ on one core, shadowing every emulated launch (outputs compared); plan and apply are not in this figure"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C5"); ap.add_argument("--scale", type=float, default=1.0); ap.add_argument("--out", default="")
    args = ap.parse_args()
    row = measure({"C1": 0, "C2": 1, "C3": 2, "C4": 3, "C5": 4}[args.config], args.scale)
    print(json.dumps(row), flush=True)
    if args.out:
        json.dump(row, open(args.out, "w"), indent=1)
