"""Randomized campaign on the GPU with shared-GPU, gpu-memory and MIG pods: the seeds of the CPU campaigns (tests/test_engine_hostsim.py fuzz families) through the C ABI
against the oracle.  usage: gpu_campaign_mig.py lo hi"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import kai_testlib as T
from test_gpu_parity import run_gpu
from test_engine_hostsim import FRAC_ACTS
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = tot = 0; t0 = time.time()
for seed in range(lo, hi):
    snap = T.pkg.synth.make_crowded_snapshot(3 + seed % 11, 884000 + seed, fill=0.3 + 0.5 * (seed % 5) / 4, n_pending_jobs=6 + seed % 13, elastic_frac=0.2, hog_frac=0.5,
                                             queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], cpu_only_frac=0.3 if seed % 4 == 0 else 0.0)
    if seed % 2: T.pkg.synth.add_mig(snap, seed, node_frac=(0.3, 0.6, 1.0)[seed % 3], pod_frac=(0.5, 0.9)[seed % 2], legacy_frac=(0.0, 0.05, 0.2)[seed % 3])
    else: T.pkg.synth.add_fractions(snap, seed, frac=0.7, memory_requests=(0.0, 0.5, 1.0)[seed % 3], gpu_memory=(100, 200, 16300)[seed % 3], portions=(0.25, 0.5, 0.75))
    cfg = T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], k_value=(0.0, 0.5, 1.0)[seed % 3], max_consolidation_preemptees=(-1, 16, 2)[seed % 3])
    cfg.min_node_gpu_memory = (100, 200, 16300)[seed % 3] if seed % 5 else 100
    acts = FRAC_ACTS[seed % len(FRAC_ACTS)] if seed % 4 else ("allocate", "consolidation", "reclaim", "preempt")
    o = T.Oracle.run(snap, cfg, acts); g = run_gpu(snap, cfg, acts); tot += 1
    ok = o.ops == g.ops and (o.pod_status == g.pod_status).all() and (o.pod_node == g.pod_node).all() and all(np.array_equal(o.nodes[k], g.nodes[k]) for k in o.nodes) \
        and all(np.allclose(o.shares_final[k], g.shares_final[k], rtol=0, atol=1e-9) for k in o.shares_final)
    if not ok:
        bad += 1; print("MISMATCH seed", seed, acts, flush=True)
    if time.time() - t0 > float(os.environ.get("CAMPAIGN_SECONDS", "150")):
        print("time budget reached at seed", seed); break
print("runs", tot, "mismatch", bad, f"{time.time()-t0:.0f}s")
