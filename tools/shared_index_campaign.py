"""Random MID-SIZE clusters with shared GPUs (40 .. 400 nodes, CPU-only / whole-GPU classes of several sizes / fractions of a device / gpu-memory requests): the sizes at which a
session's class index, the staged job path and its follow step all engage beside the brute-force passes of the fraction pods (kai_engine.hpp key_shared_layout, allocate_job_fast).
Every third seed runs a full cycle (allocate, consolidation, reclaim, preempt), the others allocate alone; KAI_SHARED_INDEX cycles through 2 / 1 / 0.  Against the oracle.
usage: shared_index_campaign.py <seed lo> <seed hi> [gpu]   (default: the host twin; CAMPAIGN_SECONDS bounds the run)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import kai_testlib as T
GPU = len(sys.argv) > 3 and sys.argv[3] == "gpu"
if GPU:
    from test_gpu_parity import run_gpu as run
else:
    from test_engine_hostsim import HostSim
    run = HostSim.run
S = T.pkg.synth
lo, hi = int(sys.argv[1]), int(sys.argv[2]); t0 = time.time(); bad = tot = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    snap = S.make_snapshot(int(rng.integers(40, 400)), int(rng.integers(300, 3000)), 7000 + seed, queue_levels=((2, 3), (3,), (2, 2, 2), (4, 4))[seed % 4], prefill=float(rng.uniform(0.1, 0.9)),
                           gpu_mix=((8, .6), (4, .2), (0, .2)), cpu_only_frac=float(rng.uniform(0.0, 0.4)), limits_frac=0.2 if seed % 2 else 0.0, zipf=bool(seed % 3 == 0),
                           elastic_frac=0.1 if seed % 5 == 0 else 0.0)
    S.add_fractions(snap, seed, frac=float(rng.uniform(0.1, 0.9)), portions=(0.25, 0.5, 0.75), memory_requests=0.3 if seed % 4 == 3 else 0.0)
    cfg = T.abi.default_config(k_value=(0.0, 0.5, 1.0)[seed % 3], gpu_strategy=T.abi.SPREAD if seed % 6 == 5 else T.abi.BINPACK, cpu_strategy=T.abi.SPREAD if seed % 7 == 6 else T.abi.BINPACK,
                               max_consolidation_preemptees=(16, -1, 4)[seed % 3])
    if seed % 4 == 3: cfg.min_node_gpu_memory = 100
    if seed % 5 == 0: cfg.plugins = (cfg.plugins & ~T.abi.PLUGINS["gpupack"]) | T.abi.PLUGINS["gpuspread"]
    if seed % 11 == 3: cfg.plugins &= ~T.abi.PLUGINS["gpusharingorder"]
    os.environ["KAI_SHARED_INDEX"] = str((2, 1, 0, 2)[seed % 4])
    acts = ("allocate", "consolidation", "reclaim", "preempt") if seed % 3 == 0 else ("allocate",)
    o = T.Oracle.run(snap, cfg, acts); tot += 1
    try: g = run(snap, cfg, acts)
    except RuntimeError as e:
        bad += 1; print("ENGINE ERROR seed", seed, acts, e, flush=True); continue
    ok = o.ops == g.ops and (o.pod_status == g.pod_status).all() and (o.pod_node == g.pod_node).all() and all(np.array_equal(o.nodes[k], g.nodes[k]) for k in o.nodes) \
        and all(np.allclose(o.shares_final[k], g.shares_final[k], rtol=0.0, atol=1e-9) for k in o.shares_final)
    if not ok: bad += 1; print("MISMATCH seed", seed, acts, "level", os.environ["KAI_SHARED_INDEX"], flush=True)
    if time.time() - t0 > float(os.environ.get("CAMPAIGN_SECONDS", "120")): break
print("shared-index campaign", "(device)" if GPU else "(host twin)", "runs", tot, "mismatch", bad, f"{time.time() - t0:.0f}s")
