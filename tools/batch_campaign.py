"""Randomized campaign of the BATCH path (plan / fill / apply): plain-gang clusters of every shape the bucket fill and the general fill kernel take — queue trees of
1 .. 3 levels, 4 / 8 / 16 devices per node, requests of 1 .. 8 (or 1, 3, 5) devices, gangs of 1 .. 100 tasks, empty to nearly full clusters, limits, zipf weights — each
against the oracle: operations, Statement numbers, pod / node state, shares and the (decisions, attempted, committed, rollbacks) counters.
usage: batch_campaign.py <seed lo> <seed hi> [gpu]     (default: the host-compiled engine with the kernels on the emulator; `gpu`: the device through the C ABI)
CAMPAIGN_SECONDS bounds the run.  Every seed also runs with one placement per step (KAI_FILL_UNBATCHED) when the seed is odd, with the general kernel when seed % 5 == 0, with the
two-worker kernel of kai_fill_counts.hpp instead of kai_fill_levels.hpp when seed % 7 == 3, with the round loop on the host (KAI_BATCH_HOST_LOOP) instead of on the device when
seed % 4 == 2, with first plans of 8 jobs per leaf (KAI_BATCH_H0: many rounds) when seed % 9 == 4."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import kai_testlib as T
lo, hi = int(sys.argv[1]), int(sys.argv[2]); GPU = len(sys.argv) > 3 and sys.argv[3] == "gpu"
if GPU:
    from test_gpu_parity import run_gpu as run
else:
    from test_engine_hostsim import HostSim
    HostSim.lib(); run = HostSim.run
S = T.pkg.synth
bad = tot = buckets = counts = levels = 0; t0 = time.time()
for seed in range(lo, hi):
    rng = np.random.default_rng(910000 + seed)
    sizes, probs = [((1, 2, 4, 8), (.4, .2, .2, .2)), ((1, 4, 8, 16, 64, 100), (.1, .2, .3, .2, .1, .1)), ((1, 2, 3, 24), (.3, .2, .2, .3)), ((1,), (1.0,))][seed % 4]
    snap = S.make_snapshot(int(rng.integers(1, 500)), int(rng.integers(1, 3000)), 910000 + seed, queue_levels=[(1,), (2, 2), (3, 4), (2, 2, 2), (4,)][seed % 5],
                           prefill=float(rng.random()) * 0.95, gpu_mix=[((8, 1.0),), ((8, .6), (4, .4)), ((16, .5), (8, .5))][seed % 3], zipf=bool(seed % 2), limits_frac=0.3 if seed % 3 == 0 else 0.0,
                           gpus_per_pod=(1, 2, 4, 8) if seed % 4 else (1, 3, 5), gang_sizes=sizes, gang_p=probs, mem_per_gpu=(8, 32)[seed % 2] * S.GIB, cpu_per_gpu=(2000.0, 4000.0)[seed % 2],
                           lexi_names=bool(seed % 7 == 0), queue_prios=(100, 200) if seed % 2 else (100,), oqws=(1.0, 2.0) if seed % 3 else (1.0,), nonpreempt_frac=0.1 * (seed % 3), usage_max=0.2 * (seed % 2))
    cfg = T.abi.default_config(gpu_strategy=T.abi.BINPACK if seed % 6 else T.abi.SPREAD, k_value=(0.0, 0.5, 1.0)[seed % 3])
    for k in ("KAI_FILL_UNBATCHED", "KAI_FILL_GENERAL", "KAI_FILL_TWO_WORKERS", "KAI_BATCH_HOST_LOOP", "KAI_BATCH_H0"): os.environ.pop(k, None)
    if seed % 4 == 2: os.environ["KAI_BATCH_HOST_LOOP"] = "1"
    if seed % 9 == 4: os.environ["KAI_BATCH_H0"] = "8"
    if seed % 7 == 3: os.environ["KAI_FILL_TWO_WORKERS"] = "1"
    if seed % 2: os.environ["KAI_FILL_UNBATCHED"] = "1"
    if seed % 5 == 0: os.environ["KAI_FILL_GENERAL"] = "1"
    o = T.Oracle.run(snap, cfg); g = run(snap, cfg); tot += 1
    buckets += int((int(g.stats.reserved[1]) >> 62) & 1) if GPU else int(g.stats.reserved[6])
    counts += int((int(g.stats.reserved[1]) >> 61) & 1) if GPU else int(g.stats.reserved[7]) & 0xffffffff  # the fill behind a counting machine (kai_fill_counts.hpp / kai_fill_levels.hpp)
    levels += int((int(g.stats.reserved[1]) >> 60) & 1) if GPU else int(g.stats.reserved[7]) >> 32         # ... with a wavefront per level and a bookkeeper (kai_fill_levels.hpp)
    ok = o.ops == g.ops and getattr(g, "stmts", o.stmts) == o.stmts and (o.pod_status == g.pod_status).all() and (o.pod_node == g.pod_node).all() and all(np.array_equal(o.nodes[k], g.nodes[k]) for k in o.nodes) \
        and all(np.array_equal(o.shares_final[k], g.shares_final[k]) for k in o.shares_final) \
        and (int(o.stats.decisions), int(o.stats.jobs_attempted), int(o.stats.jobs_committed), int(o.stats.rollbacks)) == (int(g.stats.decisions), int(g.stats.jobs_attempted), int(g.stats.jobs_committed), int(g.stats.rollbacks))
    if not ok:
        bad += 1; print("MISMATCH seed", seed, dict(os.environ).get("KAI_FILL_UNBATCHED"), dict(os.environ).get("KAI_FILL_GENERAL"), flush=True)
    if time.time() - t0 > float(os.environ.get("CAMPAIGN_SECONDS", "150")):
        print("time budget reached at seed", seed); break
print("batch campaign", "(device)" if GPU else "(host twin)", "runs", tot, "mismatch", bad, "on the sets by free devices %d, of which behind a counting machine %d, of which on k_fill_levels %d" % (buckets, counts, levels), f"{time.time()-t0:.0f}s")
