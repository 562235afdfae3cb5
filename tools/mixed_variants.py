"""Where the sequential engine's time goes on config 5's shape: the same cluster (65 536 nodes x 1 M pending pods, half full) with (a) plain gangs only, the batch path off
(engine_mode 3), (b) + 5 % elastic gangs, (c) + node labels and 5 % required-rack gangs, (d) both = bench.py --mixed.  One cycle each after a warm-up; KAI_PROF=1 prints the
engine's phase clocks of every action on stderr.  usage: mixed_variants.py [scale]   (device run; not a test)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kai_testlib as T
pkg = T.pkg; S = pkg.synth; abi = pkg.abi
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = lambda x: max(1, int(round(x * scale)))
seed = S.SEED0 + 4


def build(elastic, topo):
    s = S.make_snapshot(n(65536), n(1000000), seed, queue_levels=(8, 16, 16), prefill=0.5, zipf=True, limits_frac=0.2, queue_prios=(100, 200), oqws=(1.0, 2.0, 4.0),
                        nonpreempt_frac=0.1, usage_max=0.3, elastic_frac=0.05 if elastic else 0.0)
    if topo: S.add_topology(s, seed, zones=min(8, max(1, n(8))), racks_per_zone=max(1, min(64, n(65536) // (8 * 4))), req_rack_frac=0.05, pref_rack_frac=0.0)
    cfg = abi.default_config(k_value=0.5)
    cfg.plugins |= abi.PLUGINS.get("minruntime", 0)
    return s, cfg


for name, elastic, topo, mode in (("plain gangs, sequential engine (engine_mode 3)", False, False, 3), ("+ 5 % elastic gangs", True, False, 0),
                                  ("+ rack labels, 5 % required-rack gangs", False, True, 0), ("both (bench.py --mixed)", True, True, 0)):
    snap, cfg = build(elastic, topo); cfg.engine_mode = mode
    with pkg.KaiCore(cfg) as core:
        ssn = core.open_session(snap)
        for it in range(2):
            ssn.reset(); t0 = time.perf_counter(); ops = ssn.execute("allocate"); dt = time.perf_counter() - t0; st = ssn.stats()
        print(f"{name}: {snap.n_jobs} jobs, {len(ops)} operations, {int(st.decisions)} decisions, {int(st.jobs_attempted)} jobs attempted, {int(st.node_scans)} node passes: "
              f"{dt * 1e3:.1f} ms (device {st.kernel_ms:.1f} ms), batch path {int(st.reserved[4])}", flush=True)
        print(f"{name}: done", file=sys.stderr, flush=True)
        ssn.close()
