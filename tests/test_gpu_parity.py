"""Parity proper: the HIP path, called through the C ABI on a real MI355X, against the oracle and the reference's goldens.

Placements (integer pod→node indices, in commit order), pod states and node accounting must be bit-identical;
DRF shares are compared bit-exactly as well (north_star tolerance is 1e-6 — both sides sum in the same order).
"""
import numpy as np
import pytest

import kai_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "the -m gpu suite needs a MI355X"
    return True


def run_gpu(snap, cfg, actions=("allocate",)):
    with T.pkg.KaiCore(cfg) as core:
        ssn = core.open_session(snap)
        shares_open = ssn.queue_shares()
        ops, stmts = [], []
        for a in actions:
            arr = ssn.execute(a)
            base = (stmts[-1] + 1) if stmts else 0  # kai_op.stmt counts from 0 in every action; the oracle numbers a whole cycle
            ops += [(int(o["kind"]), int(o["pod"]), int(o["node"]), int(o["job"])) for o in arr]
            stmts += [int(o["stmt"]) + base for o in arr]
        st, nd = ssn.pod_states()
        groups = ssn.gpu_groups()
        res = T.Result(gpu_groups=groups, ops=ops, stmts=stmts, pod_status=st, pod_node=nd, shares_open=shares_open, shares_final=ssn.queue_shares(), nodes=ssn.node_states(), stats=ssn.stats())
        ssn.close()
    return res


def assert_same(res, ref):
    assert len(res.ops) == len(ref.ops)
    assert res.ops == ref.ops
    if getattr(res, "stmts", None) is not None and getattr(ref, "stmts", None) is not None:
        assert res.stmts == ref.stmts  # Statement boundaries (kai_op.stmt)
    assert (res.pod_status == ref.pod_status).all() and (res.pod_node == ref.pod_node).all()
    for k in ref.shares_open:
        assert np.array_equal(res.shares_open[k], ref.shares_open[k]), f"open {k}"
        assert np.array_equal(res.shares_final[k], ref.shares_final[k]), f"final {k}"
    for k in ref.nodes:
        assert np.array_equal(res.nodes[k], ref.nodes[k]), k


GOLD_FILES = ("allocate__allocate", "allocate__allocateGang", "allocate__allocateElastic", "allocate__allocate_subgroups", "allocate__allocateTopology",
              "reclaim__reclaim", "reclaim__reclaimDepartments", "reclaim__reclaimGang", "reclaim__reclaim_elastic", "reclaim__reclaim_sub_group",
              "preempt__preempt", "preempt__preemptGang", "preempt__preempt_elastic", "preempt__preempt_subgroups",
              "consolidation__consolidation", "consolidation__consolidation_subgroups")
GOLD = [(n, i, c, T.load_golden(n)["actions"]) for n in GOLD_FILES
        for i, c in enumerate(T.load_golden(n)["cases"])]


@pytest.mark.parametrize("name,i,case,actions", GOLD, ids=[f"{n}[{i}]" for n, i, _, _ in GOLD])
def test_gpu_reference_goldens(gpu, name, i, case, actions):
    try:
        snap, cfg, meta = T.case_to_snapshot(case)
    except T.Unsupported as e:
        pytest.skip(str(e))
    res = run_gpu(snap, cfg, actions)
    assert not T.check_expectations(snap, meta, res.pod_status, res.pod_node, res.nodes), meta["name"]
    assert_same(res, T.Oracle.run(snap, cfg, actions))


@pytest.mark.parametrize("idx,scale", [(0, 1.0), (1, 0.1), (1, 1.0), (2, 0.05), (2, 1.0), (4, 0.005)])
def test_gpu_synthetic_configs(gpu, idx, scale):
    """(2, 1.0) = BASELINE config 3 at full size end to end against the oracle (node scoring on 8 threads: about a minute)."""
    snap, cfg, _ = T.pkg.synth.config(idx, scale)
    assert_same(run_gpu(snap, cfg), T.Oracle.run(snap, cfg, threads=8 if snap.n_nodes >= 2048 else 1))


@pytest.mark.parametrize("name,idx", [("C3", 2), ("C5", 4)])
def test_gpu_full_size_operations_hash_to_the_oracles(gpu, name, idx):
    """The benched sizes pinned END TO END: profiles/full_size_pins.json holds the SHA-256 of the operation stream and of the final state (pod statuses and nodes, node accounting,
    queue shares) of the ORACLE's full-size run (tools/pin_full_sizes.py; C5: 37 minutes of oracle time on 8 cores, where the host-compiled engine agreed on every
    operation, pod, node and share) — the MI355X must produce the same two hashes."""
    import json, os
    with open(os.path.join(T.ROOT, "profiles", "full_size_pins.json")) as f:
        pin = json.load(f)[name]
    snap, cfg, desc = T.pkg.synth.config(idx, 1.0)
    assert (desc, snap.n_nodes, snap.n_pods) == (pin["workload"], pin["nodes"], pin["pods"])
    res = run_gpu(snap, cfg)
    assert len(res.ops) == pin["ops"] and T.ops_sha256(res.ops) == pin["ops_sha256"]
    assert T.state_sha256(res) == pin["state_sha256"]


@pytest.mark.parametrize("key,scale,depth", [("C4_10pct_depth0", 0.1, 0), ("C4_30pct_depth8", 0.3, 8)])  # (the FULL size, 270 s on the device, runs last of the whole suite: tests/test_zz_gpu_config4_full_size.py)
def test_gpu_config4_cycle_hashes_to_the_oracles(gpu, key, scale, depth):
    """BASELINE config 4 (allocate + consolidation + reclaim on one session: the victim search) pinned END TO END against the oracle's run of the same cycle
    (tools/pin_c4_depth.py → profiles/full_size_pins.json): 10 % with the reference's default queue depth (unlimited, conf_util/scheduler_conf_util.go:89-90), 30 % and the FULL
    size (10 000 nodes x 110 000 pods) with queueDepthPerAction 8 for the victim actions (framework/session.go:398-404; the operator docs configure 5 .. 15)."""
    import json, os
    with open(os.path.join(T.ROOT, "profiles", "full_size_pins.json")) as f:
        pin = json.load(f)[key]
    snap, cfg, desc = T.pkg.synth.config(3, scale)
    for a in ("consolidation", "reclaim", "preempt"):
        cfg.queue_depth[T.abi.ACTIONS[a]] = depth
    assert (snap.n_nodes, snap.n_pods, snap.n_jobs) == (pin["nodes"], pin["pods"], pin["jobs"])
    res = run_gpu(snap, cfg, tuple(pin["actions"]))
    assert len(res.ops) == pin["ops"] and T.ops_sha256(res.ops) == pin["ops_sha256"]


@pytest.mark.parametrize("nodes", [30, 60, 200])
def test_gpu_reclaim_large_jobs_walks_the_victims_log(gpu, nodes):
    """The reference's BenchmarkReclaimLargeJobs shape (integration_tests/reclaim/reclaim_benchmark_test.go:62-160; tools/ref_benchmarks.py): hundreds of scenarios per partial job
    that end at the AccumulatedIdleGpus filter, recorded victims in front of them — the victims log, the filter as a running sum, task groups only for the scenario that gets
    through (kai_engine_solver.inc vl_*), on 32 workgroups.  Operations, scenarios simulated and simulations run must be the oracle's (200 nodes: 26 690 dropped scenarios)."""
    import sys, os, ctypes as C
    sys.path.insert(0, os.path.join(T.ROOT, "tools"))
    import ref_benchmarks as RB
    snap, cfg, _ = T.case_to_snapshot(RB.reclaim_large(nodes), ("reclaim",))
    ref = T.Oracle.run(snap, cfg, ("reclaim",))
    out = (C.c_int64 * 3)(); T.Oracle.lib().kai_oracle_last_victim_stats(out)
    res = run_gpu(snap, cfg, ("reclaim",))
    assert_same(res, ref)
    assert (int(res.stats.reserved[2]), int(res.stats.reserved[3])) == (int(out[0]), int(out[1]))  # scenarios, simulations of the (last) victim action


def test_gpu_sessions_with_victim_actions_keep_device_memory_flat(gpu):
    """One handle, a scheduling cycle per session (scheduler.go:112-138): the slabs of a closed session serve the next one and the victim actions' replica memory (32 workgroups, a
    replica of the session arrays each) is its own allocation kept between sessions — device memory in use must not grow from the second session on (round 4's advisor finding: the
    replica buffer was pushed into the slab list without its size)."""
    import torch
    snap, cfg, _ = T.pkg.synth.config(3, 0.02)
    acts = ("allocate", "consolidation", "reclaim")
    free = []
    with T.pkg.KaiCore(cfg) as core:
        first = None
        for i in range(5):
            ssn = core.open_session(snap)
            ops = [tuple(int(o[k]) for k in ("kind", "pod", "node", "job")) for a in acts for o in ssn.execute(a)]
            ssn.close()
            torch.cuda.synchronize()
            free.append(torch.cuda.mem_get_info()[0])
            first = first or ops
            assert ops == first
    assert max(free[1:]) - min(free[1:]) <= 8 << 20, f"device memory in use moved between sessions: {[f >> 20 for f in free]} MiB free"


def test_gpu_default_cycle_on_config5_hashes_to_the_oracles(gpu):
    """The cycle the reference runs by default — allocate, consolidation, reclaim, preempt on one session (conf_util/scheduler_conf_util.go:37, without stalegangeviction) — on
    BASELINE config 5's shape at 0.5 % (328 nodes x 5 000 pods; queueDepthPerAction 8 for the victim actions): after allocate most pods are still pending and the three victim
    actions search victims for them.  Operations pinned by the oracle's run (tools/pin_full_cycle.py)."""
    import json, os
    with open(os.path.join(T.ROOT, "profiles", "full_size_pins.json")) as f:
        pin = json.load(f)["C5_0.5pct_cycle_depth8"]
    snap, cfg, _ = T.pkg.synth.config(4, pin["scale"])
    for a in ("consolidation", "reclaim", "preempt"):
        cfg.queue_depth[T.abi.ACTIONS[a]] = pin["queue_depth"]
    assert (snap.n_nodes, snap.n_pods, snap.n_jobs) == (pin["nodes"], pin["pods"], pin["jobs"])
    res = run_gpu(snap, cfg, tuple(pin["actions"]))
    assert len(res.ops) == pin["ops"] and T.ops_sha256(res.ops) == pin["ops_sha256"]


@pytest.mark.parametrize("seed", range(8))
def test_gpu_random_small(gpu, seed):
    rng = np.random.default_rng(seed)
    snap = T.pkg.synth.make_snapshot(int(rng.integers(1, 40)), int(rng.integers(0, 300)), 1000 + seed, queue_levels=(2, 3), prefill=float(rng.random()) * 0.8,
                                     gpu_mix=((8, .5), (4, .3), (0, .2)), cpu_only_frac=0.3, zipf=True, limits_frac=0.3, queue_prios=(100, 200),
                                     oqws=(1.0, 2.0), nonpreempt_frac=0.2, usage_max=0.2, lexi_names=bool(seed % 2))
    for strat in (T.abi.BINPACK, T.abi.SPREAD):
        cfg = T.abi.default_config(gpu_strategy=strat, cpu_strategy=strat, k_value=float(seed % 3) * 0.5)
        assert_same(run_gpu(snap, cfg), T.Oracle.run(snap, cfg))


def test_gpu_empty_and_ragged(gpu):
    """Edge cases: no pending pods, no nodes, a queue without jobs, a job whose queue is missing."""
    cfg = T.abi.default_config()
    for n_nodes, n_pods in ((4, 0), (0, 10), (1, 1), (3, 200)):
        snap = T.pkg.synth.make_snapshot(n_nodes, n_pods, 77, queue_levels=(1, 3), prefill=0.5)
        if snap.n_jobs:
            snap.arrays["job_queue"][0] = -1
        assert_same(run_gpu(snap, cfg), T.Oracle.run(snap, cfg))


def test_gpu_reset_replays_identically(gpu):
    snap, cfg, _ = T.pkg.synth.config(1, 0.1)
    with T.pkg.KaiCore(cfg) as core:
        ssn = core.open_session(snap)
        a = ssn.execute("allocate"); s1 = ssn.queue_shares()
        ssn.reset()
        b = ssn.execute("allocate"); s2 = ssn.queue_shares()
        ssn.close()
    assert (a == b).all() and all(np.array_equal(s1[k], s2[k]) for k in s1)


def test_gpu_best_node_matches_first_placement(gpu):
    """kai_best_node (OrderedNodesByTask + FittingNode for one task) agrees with the action's first decision for that pod."""
    snap, cfg, _ = T.pkg.synth.config(0)
    ref = T.Oracle.run(snap, cfg)
    with T.pkg.KaiCore(cfg) as core:
        ssn = core.open_session(snap)
        kind, pod, node, _ = ref.ops[0]
        got, pipe = ssn.best_node(pod)
        ssn.close()
    assert got == node and pipe == (kind == 1)


def test_gpu_full_size_properties(gpu):
    """C2 at full size: size-independent invariants (every committed pod is on exactly one node, node accounting balances,
    queue allocations equal the sum of their jobs, gangs are all-or-nothing) plus bit-exact oracle parity."""
    snap, cfg, _ = T.pkg.synth.config(1, 1.0)
    res = run_gpu(snap, cfg)
    a = snap.arrays
    placed = np.array([p for (_, p, _, _) in res.ops])
    assert len(set(placed.tolist())) == len(placed)
    # node accounting: allocatable == idle + used for every resource when nothing is releasing
    assert not res.nodes["releasing"].any()
    assert np.array_equal(a["node_allocatable"].T, res.nodes["idle"] + res.nodes["used"])
    used = np.zeros_like(res.nodes["used"])
    active = (res.pod_status & T.abi.ACTIVE_USED) != 0
    for r in range(snap.n_res):
        np.add.at(used[:, r], res.pod_node[active], a["pod_req"][r, active])
    assert np.array_equal(used, res.nodes["used"])
    assert (res.nodes["idle"] >= 0).all()
    # gang all-or-nothing: per pod-set either 0 or >= minAvailable active pods
    cnt = np.bincount(a["pod_podset"][active], minlength=snap.n_podsets)
    assert ((cnt == 0) | (cnt >= a["podset_min_available"])).all()
    assert_same(res, T.Oracle.run(snap, cfg))


@pytest.mark.parametrize("idx,scale", [(1, 0.3), (2, 0.03), (4, 0.01)])
def test_gpu_engine_modes_agree(gpu, idx, scale):
    """class index + staged job path (0), brute-force scans (1) and class index with the general job path (2): same results, same counters."""
    snap, cfg, _ = T.pkg.synth.config(idx, scale)
    ref = T.Oracle.run(snap, cfg)
    want = (ref.stats.decisions, ref.stats.jobs_attempted, ref.stats.jobs_committed, ref.stats.rollbacks)
    for mode in (0, 1, 2):
        c = T.abi.KaiConfig.from_buffer_copy(cfg); c.engine_mode = mode
        res = run_gpu(snap, c)
        assert_same(res, ref)
        assert (res.stats.decisions, res.stats.jobs_attempted, res.stats.jobs_committed, res.stats.rollbacks) == want, mode


@pytest.mark.parametrize("seed", range(4))
def test_gpu_elastic_and_subgroups(gpu, seed):
    """elastic jobs (re-pushed with a changed order key), two-pod-set gangs and task-order labels through the C ABI, all engine modes"""
    rng = np.random.default_rng(300 + seed)
    snap = T.pkg.synth.make_snapshot(int(rng.integers(4, 120)), int(rng.integers(20, 900)), 3000 + seed, queue_levels=(2, 2), prefill=float(rng.random()) * 0.6,
                                     gpu_mix=((8, .6), (4, .4)), zipf=bool(seed % 2), limits_frac=0.2, queue_prios=(100, 200), oqws=(1.0, 2.0),
                                     nonpreempt_frac=0.1, elastic_frac=0.4, multi_podset_frac=0.3, task_prio_frac=0.3, lexi_names=bool(seed % 3 == 0))
    cfg = T.abi.default_config(k_value=float(seed % 2))
    ref = T.Oracle.run(snap, cfg)
    for mode in (0, 1, 2):
        c = T.abi.KaiConfig.from_buffer_copy(cfg); c.engine_mode = mode
        assert_same(run_gpu(snap, c), ref)


def test_gpu_best_node_with_node_sets(gpu):
    """kai_best_node over node subsets (what SubsetNodesFn hands to OrderedNodesByTask) against the oracle, bin-pack and spread"""
    import ctypes as C
    lib = T.Oracle.lib(); lib.kai_oracle_best_node.restype = C.c_int
    rng = np.random.default_rng(11)
    snap = T.pkg.synth.make_snapshot(150, 400, 4242, queue_levels=(2, 2), prefill=0.5, gpu_mix=((8, .5), (4, .3), (0, .2)), cpu_only_frac=0.3, lexi_names=True)
    pending = np.nonzero(snap.arrays["pod_status"] == T.abi.POD_STATUS["Pending"])[0]
    for strat in (T.abi.BINPACK, T.abi.SPREAD):
        cfg = T.abi.default_config(gpu_strategy=strat, cpu_strategy=strat)
        s = snap.as_struct()
        with T.pkg.KaiCore(cfg) as core:
            ssn = core.open_session(snap)
            for trial in range(24):
                pod = int(rng.choice(pending))
                subset = None if trial % 4 == 0 else np.nonzero(rng.random(snap.n_nodes) < rng.choice([0.02, 0.2, 0.7]))[0].tolist()
                got = ssn.best_node(pod, pipeline_only=bool(trial % 3 == 0), nodeset=subset)
                words = None
                if subset is not None:
                    w = np.zeros((snap.n_nodes + 31) // 32, np.uint32)
                    for n in subset: w[n >> 5] |= np.uint32(1 << (n & 31))
                    words = w.ctypes.data_as(C.POINTER(C.c_uint32))
                node, pipe = C.c_int(-1), C.c_int(0)
                assert lib.kai_oracle_best_node(C.byref(cfg), C.byref(s), pod, words, int(trial % 3 == 0), C.byref(node), C.byref(pipe)) == 0
                assert got == (node.value, bool(pipe.value)), (strat, trial, pod, got, node.value, pipe.value)
            ssn.close()


def crowded(seed):
    snap = T.pkg.synth.make_crowded_snapshot(6 + seed % 17, 1000 + seed, fill=0.85 + 0.1 * (seed % 2), queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3],
                                             cpu_only_frac=0.2 if seed % 5 == 0 else 0.0)
    cfg = T.abi.default_config(max_consolidation_preemptees=-1 if seed % 2 else 16)
    cfg.use_scheduling_signatures = seed % 2; cfg.allow_consolidating_reclaim = int(seed % 3 != 0)  # MinimalJobRepresentatives on for odd seeds
    return snap, cfg


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("actions", [("reclaim",), ("preempt",), ("consolidation",), ("allocate", "consolidation", "reclaim", "preempt")], ids=lambda a: "+".join(a))
def test_gpu_victim_actions_crowded_cluster(gpu, seed, actions):
    """reclaim / preempt / consolidation (scenario solver) on a nearly full cluster: committed evictions, pipelines and allocations,
    pod states, node accounting and queue shares identical to the oracle; the full action list runs in ONE session like a scheduling cycle."""
    snap, cfg = crowded(seed)
    assert_same(run_gpu(snap, cfg, actions), T.Oracle.run(snap, cfg, actions))


@pytest.mark.parametrize("scale", [0.003, 0.01])
def test_gpu_config4_topology_consolidation_reclaim(gpu, scale):
    """BASELINE config 4 (scaled): topology-constrained gangs on a cluster 85 % full of preemptible jobs; allocate, consolidation, reclaim in one session."""
    snap, cfg, _ = T.pkg.synth.config(3, scale)
    acts = ("allocate", "consolidation", "reclaim")
    ref = T.Oracle.run(snap, cfg, acts)
    assert any(o[0] == 2 for o in ref.ops)  # the cycle really evicts
    assert_same(run_gpu(snap, cfg, acts), ref)


@pytest.mark.parametrize("scale", [0.02, 0.1])
def test_gpu_config5_in_the_mixed_shape(gpu, scale):
    """BASELINE config 5 as SURVEY 8d writes it down (bench.py --config C5 --mixed), scaled: 521 topology domains (the domain loops of subSetNodesFn run on the scan
    lanes: TopoScan ops 5 .. 14), 5 % of the gangs with a required rack, 5 % elastic gangs, half the cluster running — the sequential engine against the oracle."""
    snap, cfg, _ = T.pkg.synth.config(4, scale, mixed=True)
    assert_same(run_gpu(snap, cfg, ("allocate",)), T.Oracle.run(snap, cfg, ("allocate",), threads=8 if snap.n_nodes >= 2048 else 1))


@pytest.mark.parametrize("wgs", (2, 7, 64, 128))
def test_gpu_scan_grid_on_several_workgroups(gpu, wgs, monkeypatch):
    """The allocate action of the sequential engine with its passes over the nodes spread over 2 / 7 / 64 / 128 workgroups (ScanGrid, kai_kernels.hpp: pre-order range, best
    node of a decision that has no class index — fractions of a device, node sets of the topology DFS —, the node loops of subSetNodesFn): shared-GPU clusters, replica
    groups with a required rack, the mixed config 5 at 2 % (521 topology domains) and sub-group jobs, each against the oracle."""
    from test_engine_hostsim import _same_groups
    monkeypatch.setenv("KAI_SCAN_WGS", str(wgs))
    for seed in (1, 4, 12, 23):
        snap = T.pkg.synth.make_crowded_snapshot(2 + seed % 9, 9300 + seed, fill=0.3 + 0.5 * (seed % 5) / 4, n_pending_jobs=6 + seed % 13, elastic_frac=0.2,
                                                 hog_frac=0.5, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], cpu_only_frac=0.3 if seed % 4 == 0 else 0.0)
        T.pkg.synth.add_fractions(snap, seed, frac=0.6, portions=(0.25, 0.5, 0.75))
        cfg = T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], k_value=(0.0, 0.5, 1.0)[seed % 3])
        ref = T.Oracle.run(snap, cfg, ("allocate",)); res = run_gpu(snap, cfg, ("allocate",))
        assert_same_tol(res, ref); _same_groups(snap, res, ref)
    for seed in (2, 7, 19):
        snap = T.pkg.synth.make_crowded_snapshot(6 + seed % 13, 3000 + seed, fill=0.8 + 0.1 * (seed % 2), queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], two_podsets_frac=0.6, n_pending_jobs=14)
        T.pkg.synth.add_replica_topology(snap, seed, zones=2, nodes_per_rack=2 + seed % 2)
        cfg = T.abi.default_config(max_consolidation_preemptees=-1)
        for acts in (("allocate",), ("allocate", "consolidation", "reclaim", "preempt")):
            assert_same(run_gpu(snap, cfg, acts), T.Oracle.run(snap, cfg, acts))
    snap, cfg, _ = T.pkg.synth.config(4, 0.02, mixed=True)
    res = run_gpu(snap, cfg, ("allocate",))
    assert_same(res, T.Oracle.run(snap, cfg, ("allocate",)))
    assert 2 <= (int(res.stats.reserved[1]) >> 48) <= wgs  # the engine's workgroup + the helpers that signed on (those on the engine's own XCD stay out)
    snap, cfg, _ = T.pkg.synth.config(2, 0.05)
    T.pkg.synth.add_fractions(snap, 7, frac=0.3)
    ref = T.Oracle.run(snap, cfg, ("allocate",)); res = run_gpu(snap, cfg, ("allocate",))
    assert_same_tol(res, ref); _same_groups(snap, res, ref)


def test_gpu_scan_grid_at_full_size_against_the_host_compiled_engine(gpu):
    """The two workloads the scan grid is for, at the benched sizes, against the host-compiled engine (same source, g++, passes over the nodes as plain loops): config 5 in the
    mixed shape (65 536 nodes, 521 topology domains) and config 3 with 30 % of the one-GPU pods as fractions of a device (every decision a pass over 10 000 nodes' GPU groups)."""
    from test_engine_hostsim import HostSim, _same_groups
    snap, cfg, _ = T.pkg.synth.config(4, 1.0, mixed=True)
    res, ref = run_gpu(snap, cfg, ("allocate",)), HostSim.run(snap, cfg, ("allocate",))
    assert_same(res, ref)
    snap, cfg, _ = T.pkg.synth.config(2, 1.0)
    T.pkg.synth.add_fractions(snap, 7, frac=0.3)
    res, ref = run_gpu(snap, cfg, ("allocate",)), HostSim.run(snap, cfg, ("allocate",))
    assert_same_tol(res, ref); _same_groups(snap, res, ref)


@pytest.mark.parametrize("scale,depth", [(0.02, 3), (0.05, 8)])
def test_gpu_config4_with_queue_depth(gpu, scale, depth):
    """BASELINE config 4 with queueDepthPerAction for the victim actions (the reference's operator docs configure 5 .. 15): allocate, consolidation, reclaim on the device
    (victim actions on 32 workgroups) against the oracle.  bench.py --config C4 --scale 0.3 --queue-depth 8 is the benched form (profiles/full_size_pins.json holds the oracle's hash)."""
    snap, cfg, _ = T.pkg.synth.config(3, scale)
    for a in ("consolidation", "reclaim", "preempt"):
        cfg.queue_depth[T.abi.ACTIONS[a]] = depth
    acts = ("allocate", "consolidation", "reclaim")
    ref = T.Oracle.run(snap, cfg, acts)
    assert any(o[0] == 2 for o in ref.ops)
    assert_same(run_gpu(snap, cfg, acts), ref)


import test_oracle_golden as _G
INTEG_FILES = _G.INTEG_FILES  # all 113 scenarios, incl. the fraction, GPU-memory and MIG tables
INTEG = [(n, i, c) for n in INTEG_FILES for i, c in enumerate(T.load_golden(n)["cases"])]


@pytest.mark.parametrize("name,i,case", INTEG, ids=[f"{n}[{i}]" for n, i, _ in INTEG])
def test_gpu_integration_rounds(gpu, name, i, case):
    """The reference's integration tests: full cycles (allocate, consolidation, reclaim, preempt) over several rounds with state fed back;
    every round identical to the oracle and the final cluster state as the reference expects."""
    from test_engine_hostsim import _same_groups
    def run_both(snap, cfg, actions):
        res = run_gpu(snap, cfg, actions)
        ref = T.Oracle.run(snap, cfg, actions)
        if "pod_gpu_portion" in snap.arrays:
            assert_same_tol(res, ref); _same_groups(snap, res, ref)
        else:
            assert_same(res, ref)
        return res
    try:
        errs = T.run_integration(case, run_both, rounds_after=1, fractions=True)
    except T.Unsupported as e:
        pytest.skip(str(e))
    assert not errs, errs[:4]


@pytest.mark.parametrize("seed", range(12))
def test_gpu_minruntime_protection(gpu, seed):
    """minruntime plugin (plugins/minruntime): start times up to 2 h before "now", min-runtimes of 0 / 30 / 60 min on half of the queues, both
    reclaim resolve methods — victim filters and the elastic-victim scenario validators must match the oracle, and protection must matter."""
    snap = T.pkg.synth.make_crowded_snapshot(6 + seed % 17, 1000 + seed, fill=0.9, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], minruntime=True)
    cfg = T.abi.default_config(max_consolidation_preemptees=-1); cfg.use_scheduling_signatures = seed % 2
    cfg.now_ns = T.pkg.synth.NOW_NS; cfg.default_preempt_min_runtime_ns = 0 if seed % 4 else 900 * 10**9
    cfg.default_reclaim_min_runtime_ns = 600 * 10**9 if seed % 3 == 0 else 0; cfg.reclaim_resolve_method = seed % 2
    for actions in (("reclaim",), ("preempt",), ("allocate", "consolidation", "reclaim", "preempt")):
        assert_same(run_gpu(snap, cfg, actions), T.Oracle.run(snap, cfg, actions))


@pytest.mark.parametrize("seed", range(12))
def test_gpu_replica_groups_with_required_rack(gpu, seed):
    """Two sub-groups of one job on the same required topology level (the replica fixtures of allocateTopology_test.go) in a crowded
    cluster: allocation through the sub-group DFS and the victim actions with the TopologyAwareIdleGpus scenario filter in play."""
    snap = T.pkg.synth.make_crowded_snapshot(6 + seed % 13, 3000 + seed, fill=0.8 + 0.1 * (seed % 2), queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3],
                                             two_podsets_frac=0.6, n_pending_jobs=14)
    T.pkg.synth.add_replica_topology(snap, seed, zones=2, nodes_per_rack=2 + seed % 2)
    cfg = T.abi.default_config(max_consolidation_preemptees=-1); cfg.use_scheduling_signatures = seed % 2
    for actions in (("allocate",), ("reclaim",), ("preempt",), ("consolidation",), ("allocate", "consolidation", "reclaim", "preempt")):
        assert_same(run_gpu(snap, cfg, actions), T.Oracle.run(snap, cfg, actions))


@pytest.mark.parametrize("seed", [9, 36, 123, 234, 1623] + list(range(3000, 3012)))
def test_gpu_broad_random_cycles(gpu, seed):
    """Seeds of the broad randomized campaign (tests/kai_testlib.py::broad_case), incl. the ones that exposed real divergences: staged job
    path after a victim action, candidate-node order under lexicographic node names, the session job's own tasks cache, leaf heaps
    whose keys change while a job waits.  Every case: operations, pod states, node accounting and queue shares identical to the oracle."""
    for snap, cfg, actions in T.broad_case(seed):
        assert_same(run_gpu(snap, cfg, actions), T.Oracle.run(snap, cfg, actions))


# ------------------------------------------------------------------------------------------------ batch path (kai_batch.hpp) on the MI355X
def stats_tuple(s):
    return (s.decisions, s.jobs_attempted, s.jobs_committed, s.rollbacks)


@pytest.mark.parametrize("seed", range(24))
def test_gpu_batch_random_regular(gpu, seed):
    """plain-gang clusters over every queue-tree shape: the allocate action must take the batch path and equal the oracle and the sequential engine"""
    from test_batch_path import regular_snapshot
    snap = regular_snapshot(seed)
    cfg = T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], cpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[(seed // 2) % 2], k_value=float(seed % 3) * 0.5)
    ref = T.Oracle.run(snap, cfg)
    res = run_gpu(snap, cfg)
    assert_same(res, ref)
    assert stats_tuple(res.stats) == stats_tuple(ref.stats)
    assert res.stats.reserved[4] >= 1 or snap.n_jobs == 0 or ref.stats.jobs_attempted == 0, "the allocate action did not take the batch path"
    seq = T.abi.KaiConfig.from_buffer_copy(cfg); seq.engine_mode = 3
    res3 = run_gpu(snap, seq)
    assert res3.stats.reserved[4] == 0
    assert_same(res3, ref)


def on_buckets(stats):
    """the fill of the batch path ran on k_fill_buckets (sets of nodes by free devices in LDS; kai_core.hip puts bit 62 of reserved[1])"""
    return bool((int(stats.reserved[1]) >> 62) & 1)


@pytest.mark.parametrize("seed", range(16))
def test_gpu_bucket_fill_against_oracle_and_general_kernel(gpu, seed, monkeypatch):
    """bin-packed GPU classes on nodes where only the devices can bind: k_fill_buckets against the oracle, then the general k_fill on the same snapshot"""
    rng = np.random.default_rng(5100 + seed)
    snap = T.pkg.synth.make_snapshot(int(rng.integers(1, 400)), int(rng.integers(1, 1500)), 5100 + seed, queue_levels=[(1,), (2, 2), (3, 4), (2, 2, 2)][seed % 4],
                                     prefill=float(rng.random()) * 0.9, gpu_mix=((8, .6), (4, .4)) if seed % 2 else ((8, 1.0),), zipf=bool(seed % 2),
                                     limits_frac=0.3 if seed % 3 == 0 else 0.0, lexi_names=bool(seed % 5 == 0), gpus_per_pod=(1, 2, 4, 8) if seed % 4 else (1, 3, 5))
    cfg = T.abi.default_config(k_value=0.5)
    ref = T.Oracle.run(snap, cfg)
    res = run_gpu(snap, cfg)
    assert_same(res, ref)
    assert stats_tuple(res.stats) == stats_tuple(ref.stats)
    assert res.stats.reserved[4] >= 1 and on_buckets(res.stats)
    monkeypatch.setenv("KAI_FILL_GENERAL", "1")
    gen = run_gpu(snap, cfg)
    assert gen.stats.reserved[4] >= 1 and not on_buckets(gen.stats)
    assert_same(gen, ref)


@pytest.mark.parametrize("seed", range(12))
def test_gpu_bucket_fill_whole_nodes_per_step(gpu, seed, monkeypatch):
    """k_fill_buckets places a gang of one class in steps of whole nodes (kai_fill_buckets.hpp): large gangs of small requests on 16-device nodes, 3- and 5-device requests,
    nearly full and empty clusters — against the oracle and against one placement per step (KAI_FILL_UNBATCHED); the CPU twin of this test also runs the general kernel"""
    rng = np.random.default_rng(5600 + seed)
    sizes, probs = ((1, 4, 8, 16, 64, 100), (.1, .2, .3, .2, .1, .1)) if seed % 2 else ((1, 2, 3, 24), (.3, .2, .2, .3))
    snap = T.pkg.synth.make_snapshot(int(rng.integers(3, 300)), int(rng.integers(50, 2500)), 5600 + seed, queue_levels=[(1,), (2, 2), (3, 4)][seed % 3], prefill=(0.0, 0.3, 0.6, 0.9)[seed % 4],
                                     gpu_mix=((16, .5), (8, .5)) if seed % 3 == 0 else ((8, .7), (4, .3)), gpus_per_pod=(1, 2, 4, 8) if seed % 4 else (1, 3, 5), gang_sizes=sizes, gang_p=probs,
                                     mem_per_gpu=8 * T.pkg.synth.GIB, cpu_per_gpu=2000.0, lexi_names=bool(seed % 5 == 0))
    cfg = T.abi.default_config(k_value=0.5)
    ref = T.Oracle.run(snap, cfg)
    res = run_gpu(snap, cfg)
    assert_same(res, ref)
    assert stats_tuple(res.stats) == stats_tuple(ref.stats)
    if not on_buckets(res.stats):
        pytest.skip("this cluster does not qualify for the bucket fill")
    monkeypatch.setenv("KAI_FILL_UNBATCHED", "1")
    one = run_gpu(snap, cfg)
    assert on_buckets(one.stats)
    assert_same(one, ref); assert stats_tuple(one.stats) == stats_tuple(ref.stats)


def on_counts(stats):
    """... as two wavefronts: the planned order over the levels' populations, the sets behind a command ring (kai_fill_counts.hpp; bit 61 of reserved[1])"""
    return bool((int(stats.reserved[1]) >> 61) & 1)


def on_levels(stats):
    """... with a wavefront per level behind the counting machine and a bookkeeper for the dead gangs (kai_fill_levels.hpp; bit 60 of reserved[1])"""
    return bool((int(stats.reserved[1]) >> 60) & 1)


@pytest.mark.parametrize("seed", range(20))
def test_gpu_counts_fill_against_the_one_wave_kernel_and_the_oracle(gpu, seed, monkeypatch):
    """k_fill_counts (kai_fill_counts.hpp) takes the clusters where no class carries a static bitmap of its own: gangs of one class decided from the levels' populations, gangs of
    several classes on a copy of the counts, the tasks' nodes written by the set worker behind the ring — against the oracle and the one-wave k_fill_buckets (KAI_FILL_ONE_WAVE)
    on the same snapshot (the CPU twin of this test also runs the general kernel and the native scalar shadow)."""
    rng = np.random.default_rng(5900 + seed)
    sizes, probs = [((1, 4, 8, 16, 64, 100), (.1, .2, .3, .2, .1, .1)), ((1, 2, 3, 24), (.3, .2, .2, .3)), ((1, 2, 700), (.6, .3, .1)), ((1,), (1.0,))][seed % 4]
    snap = T.pkg.synth.make_snapshot(int(rng.integers(1, 500)), int(rng.integers(1, 3000)), 5900 + seed, queue_levels=[(1,), (2, 2), (3, 4), (2, 2, 2)][seed % 4], prefill=(0.0, 0.3, 0.6, 0.95)[seed % 4],
                                     gpu_mix=((16, .5), (8, .5)) if seed % 3 == 0 else ((8, .7), (4, .3)), gpus_per_pod=(1, 2, 4, 8) if seed % 5 else (1, 3, 5), gang_sizes=sizes, gang_p=probs,
                                     mem_per_gpu=8 * T.pkg.synth.GIB, cpu_per_gpu=2000.0, zipf=bool(seed % 2), limits_frac=0.3 if seed % 3 == 1 else 0.0, lexi_names=bool(seed % 7 == 0))
    cfg = T.abi.default_config(k_value=0.5)
    ref = T.Oracle.run(snap, cfg)
    res = run_gpu(snap, cfg)
    assert_same(res, ref)
    assert stats_tuple(res.stats) == stats_tuple(ref.stats)
    if not on_buckets(res.stats):
        pytest.skip("this cluster does not qualify for the sets by free devices")
    assert on_counts(res.stats)
    assert on_levels(res.stats) == (seed % 3 != 0)  # (16-device nodes: more levels than kai_fill_levels.hpp has wavefronts for)
    if on_levels(res.stats):  # the kernel of kai_fill_counts.hpp (two set workers) on the same snapshot
        monkeypatch.setenv("KAI_FILL_TWO_WORKERS", "1")
        two = run_gpu(snap, cfg)
        assert on_counts(two.stats) and not on_levels(two.stats)
        assert_same(two, ref); assert stats_tuple(two.stats) == stats_tuple(ref.stats)
        monkeypatch.delenv("KAI_FILL_TWO_WORKERS")
    monkeypatch.setenv("KAI_FILL_ONE_WAVE", "1")
    one = run_gpu(snap, cfg)
    assert on_buckets(one.stats) and not on_counts(one.stats)
    assert_same(one, ref); assert stats_tuple(one.stats) == stats_tuple(ref.stats)


def on_device_loop(stats):
    """the round loop's state lived on the device (RoundCtl / k_round_next: rounds without the host; bit 59 of reserved[1])"""
    return bool((int(stats.reserved[1]) >> 59) & 1)


@pytest.mark.parametrize("seed", range(16))
def test_gpu_round_loop_on_the_device_against_the_loop_on_the_host(gpu, seed, monkeypatch):
    """Rounds without the host: the next round's kernels are on the stream before the last round's status has been seen (they read how far to plan and where their output starts
    from RoundCtl; the round enqueued in vain at the end finds `done` and leaves) — against the loop that drains the stream after every round (KAI_BATCH_HOST_LOOP=1) and the
    oracle: same operations, Statement numbers, counters, rounds.  Every fourth seed starts with plans of 8 jobs per leaf (many rounds, the depth moves both ways)."""
    import test_batch_path as B
    snap = B.regular_snapshot(100 + seed) if seed % 2 else T.pkg.synth.make_snapshot(40 + 30 * seed, 400 + 150 * seed, 7700 + seed, queue_levels=[(2, 2), (3, 4), (1,), (2, 2, 2)][seed % 4],
                                                                                     prefill=0.1 * (seed % 7), gpu_mix=((8, .6), (4, .4)), limits_frac=0.3 if seed % 3 == 0 else 0.0)
    cfg = T.abi.default_config(k_value=0.5, gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[(seed // 2) % 2])
    if seed % 4 == 3:
        monkeypatch.setenv("KAI_BATCH_H0", "8")
    ref = T.Oracle.run(snap, cfg)
    dev = run_gpu(snap, cfg)
    assert_same(dev, ref); assert stats_tuple(dev.stats) == stats_tuple(ref.stats)
    assert dev.stats.reserved[3] >= 0 and on_device_loop(dev.stats)
    monkeypatch.setenv("KAI_BATCH_HOST_LOOP", "1")
    host = run_gpu(snap, cfg)
    assert not on_device_loop(host.stats)
    assert_same(host, ref); assert stats_tuple(host.stats) == stats_tuple(ref.stats)
    assert int(host.stats.reserved[4]) == int(dev.stats.reserved[4]), "the two loops took different numbers of rounds"


@pytest.mark.parametrize("seed", range(8))
def test_gpu_plan_scan_in_segments_against_one_workgroup_per_node(gpu, seed, monkeypatch):
    """kai_plan_segments.hpp on the device (segments of 1 024 positions): forced for every height below the root (KAI_PLAN_SEG_MIN=1) against k_plan_scan (never) and the oracle;
    trees with limited inner queues (their gate ends a stream inside the plan) and plain ones; seeds 6, 7: streams of several thousand jobs per top-level queue (several segments)."""
    import test_batch_path as B
    if seed < 4:
        snap, cfg = B.inner_limits_snapshot(seed)
    elif seed < 6:
        snap, cfg = B.regular_snapshot(200 + seed), T.abi.default_config(k_value=0.5)
    else:
        snap = T.pkg.synth.make_snapshot(1200, 14000, 8800 + seed, queue_levels=(2, 3), prefill=0.3, gang_sizes=(1, 2), gang_p=(.8, .2), gpus_per_pod=(1, 2), zipf=True, limits_frac=0.3,
                                         inner_limits_frac=1.0 if seed == 7 else 0.0, nonpreempt_frac=0.1)
        cfg = T.abi.default_config(k_value=0.5)
    ref = T.Oracle.run(snap, cfg)
    monkeypatch.setenv("KAI_PLAN_SEG_MIN", "1")
    seg = run_gpu(snap, cfg)
    assert int(seg.stats.reserved[4]) >= 1, "the allocate action did not take the batch path"
    assert_same(seg, ref); assert stats_tuple(seg.stats) == stats_tuple(ref.stats)
    monkeypatch.setenv("KAI_PLAN_SEG_MIN", "1000000000")
    one = run_gpu(snap, cfg)
    assert_same(one, ref); assert stats_tuple(one.stats) == stats_tuple(ref.stats)
    assert int(one.stats.reserved[4]) == int(seg.stats.reserved[4]), "the two forms of the scan planned different rounds"


def test_gpu_bucket_fill_corners(gpu):
    """what k_bucket_build turns away (another resource may bind first, 32 devices per node) runs on the general kernel; static predicates per class, 16
    devices per node, a nearly full cluster run on the bucket kernel — all equal to the oracle"""
    S = T.pkg.synth
    cases = []
    cases.append((S.make_snapshot(60, 500, 5201, prefill=0.2, cpu_per_gpu=20000.0), False))
    cases.append((S.make_snapshot(60, 500, 5210, prefill=0.2, mem_per_gpu=96 * S.GIB), False))
    cases.append((S.make_snapshot(50, 600, 5401, gpu_mix=((32, .5), (8, .5)), mem_per_gpu=4 * S.GIB, cpu_per_gpu=1000.0, prefill=0.4), False))
    cases.append((S.make_snapshot(50, 600, 5400, gpu_mix=((16, .5), (8, .5)), gpus_per_pod=(1, 2, 4, 8, 16), mem_per_gpu=8 * S.GIB, cpu_per_gpu=2000.0, prefill=0.4), True))
    cases.append((S.make_snapshot(130, 900, 5403, prefill=0.97), True))
    for seed in (0, 2, 4):
        rng = np.random.default_rng(5300 + seed)
        snap = S.make_snapshot(int(rng.integers(20, 300)), int(rng.integers(100, 1200)), 5300 + seed, queue_levels=(2, 3), prefill=0.3)
        S.add_predicate_features(snap, 5300 + seed, nominated_frac=0.0, oversized_frac=0.0)
        a = snap.arrays
        a["node_class"] = rng.integers(0, 3, size=snap.n_nodes).astype(np.int32)
        a["pod_class"] = np.repeat(rng.integers(0, 2, size=snap.n_jobs), a["job_n_pods"]).astype(np.int32)
        a["class_fit"] = np.array([[1, 1, 0], [1, 0, 1]], np.uint8)
        snap.finalize()
        cases.append((snap, True))
    for snap, want in cases:
        cfg = T.abi.default_config(k_value=0.5)
        ref = T.Oracle.run(snap, cfg)
        res = run_gpu(snap, cfg)
        assert_same(res, ref)
        assert stats_tuple(res.stats) == stats_tuple(ref.stats)
        assert res.stats.reserved[4] >= 1 and on_buckets(res.stats) == want


@pytest.mark.parametrize("idx,scale", [(1, 1.0), (2, 0.1), (4, 0.02)])
def test_gpu_batch_and_sequential_engine_agree(gpu, idx, scale):
    snap, cfg, _ = T.pkg.synth.config(idx, scale)
    res = run_gpu(snap, cfg)
    assert res.stats.reserved[4] >= 1
    seq = T.abi.KaiConfig.from_buffer_copy(cfg); seq.engine_mode = 3
    res3 = run_gpu(snap, seq)
    assert res3.stats.reserved[4] == 0
    assert res.ops == res3.ops and (res.pod_node == res3.pod_node).all() and stats_tuple(res.stats) == stats_tuple(res3.stats)
    for k in res.shares_final:
        assert np.array_equal(res.shares_final[k], res3.shares_final[k])
    for k in res.nodes:
        assert np.array_equal(res.nodes[k], res3.nodes[k])


def test_gpu_three_level_index_against_the_oracle(gpu):
    """8 192 nodes: the class index has two super-blocks (NSB > 1), the level the small cases never reach; oracle = 40 s of CPU"""
    snap, cfg, _ = T.pkg.synth.config(4, 0.125)
    ref = T.Oracle.run(snap, cfg)
    for mode in (0, 3):
        c = T.abi.KaiConfig.from_buffer_copy(cfg); c.engine_mode = mode
        res = run_gpu(snap, c)
        assert_same(res, ref)
        assert stats_tuple(res.stats) == stats_tuple(ref.stats)
        assert (res.stats.reserved[4] >= 1) == (mode == 0)


def test_gpu_quarter_scale_modes_agree(gpu):
    """BASELINE config 5 at a quarter of its size (16 384 nodes x 250 000 pods): batch path vs sequential engine on the device"""
    snap, cfg, _ = T.pkg.synth.config(4, 0.25)
    res = run_gpu(snap, cfg)
    seq = T.abi.KaiConfig.from_buffer_copy(cfg); seq.engine_mode = 3
    res3 = run_gpu(snap, seq)
    assert res.stats.reserved[4] >= 1 and res3.stats.reserved[4] == 0
    assert res.ops == res3.ops and (res.pod_status == res3.pod_status).all() and (res.pod_node == res3.pod_node).all()
    assert stats_tuple(res.stats) == stats_tuple(res3.stats)
    for k in res.nodes:
        assert np.array_equal(res.nodes[k], res3.nodes[k])
    for k in res.shares_final:
        assert np.array_equal(res.shares_final[k], res3.shares_final[k])


def test_gpu_full_size_against_the_host_compiled_engine(gpu):
    """BASELINE config 5 at FULL size (65 536 nodes x 1 000 000 pending pods, the benched workload): every committed operation, pod state, node
    and queue share of the MI355X's batch path against the host-compiled sequential engine (tests/host_sim, 1.2 s of CPU; itself held to the
    oracle at every size the oracle finishes) and against the oracle's first decisions."""
    from test_engine_hostsim import HostSim
    snap, cfg, _ = T.pkg.synth.config(4, 1.0)
    res = run_gpu(snap, cfg)
    assert res.stats.reserved[4] >= 1
    seq = T.abi.KaiConfig.from_buffer_copy(cfg); seq.engine_mode = 3
    twin = HostSim.run(snap, seq)
    assert res.ops == twin.ops
    assert (res.pod_status == twin.pod_status).all() and (res.pod_node == twin.pod_node).all()
    assert stats_tuple(res.stats) == stats_tuple(twin.stats)
    for k in twin.nodes:
        assert np.array_equal(res.nodes[k], twin.nodes[k])
    for k in twin.shares_final:
        assert np.array_equal(res.shares_final[k], twin.shares_final[k])
    bounded = T.abi.KaiConfig.from_buffer_copy(cfg); bounded.reserved[0] = 3000  # the oracle stops after its first 3 000 decisions
    ref = T.Oracle.run(snap, bounded)
    assert len(ref.ops) > 500 and res.ops[:len(ref.ops)] == ref.ops


# ------------------------------------------------------------------------------------------------ shared GPUs (fractions of one device) on the MI355X
def assert_same_tol(res, ref, tol=1e-9):
    """with fractional quantities the queue sums are no longer sums of integers: the reference itself adds them in Go-map order, so shares are held to
    1e-9 (north_star: 1e-6); placements, pod states and node accounting stay exact"""
    assert res.ops == ref.ops and res.stmts == ref.stmts
    assert (res.pod_status == ref.pod_status).all() and (res.pod_node == ref.pod_node).all()
    for k in ref.shares_open:
        assert np.allclose(res.shares_open[k], ref.shares_open[k], rtol=0.0, atol=tol), k
        assert np.allclose(res.shares_final[k], ref.shares_final[k], rtol=0.0, atol=tol), k
    for k in ref.nodes:
        assert np.array_equal(res.nodes[k], ref.nodes[k]), k


def _frac_cases():
    from test_engine_hostsim import FRAC_GOLD, MEM_GOLD, MIG_GOLD, FRAC_VICTIM_GOLD
    out = [("allocate__allocateFractionalGpu", i, c, ("allocate",)) for i, c in FRAC_GOLD] + [("allocate__allocateGpuMemory", i, c, ("allocate",)) for i, c in MEM_GOLD] + [("allocate__allocateMIG", i, c, ("allocate",)) for i, c in MIG_GOLD]
    return out + [(n, i, c, a) for n, i, c, a in FRAC_VICTIM_GOLD]


@pytest.mark.parametrize("name,i,case,actions", _frac_cases(), ids=[f"{n}[{i}]" for n, i, _, _ in _frac_cases()])
def test_gpu_fractional_goldens(gpu, name, i, case, actions):
    """the reference's golden scenarios with shared devices, GPU-memory requests and MIG (allocateFractionalGpu / GpuMemory / MIG _test.go, the reclaim /
    preempt / consolidation GpuMemory and MIG tables and the fraction cases inside the other victim-action tables): expectations of the
    reference, and operations / states / node accounting / GPU groups of the oracle"""
    from test_engine_hostsim import _same_groups
    snap, cfg, meta = T.case_to_snapshot(case, fractions=True)
    ref = T.Oracle.run(snap, cfg, actions)
    res = run_gpu(snap, cfg, actions)
    assert_same_tol(res, ref)
    _same_groups(snap, res, ref)
    assert not T.check_expectations(snap, meta, res.pod_status, res.pod_node, res.nodes, res.gpu_groups)


@pytest.mark.parametrize("seed", range(30))
def test_gpu_fraction_fuzz(gpu, seed):
    from test_engine_hostsim import _same_groups, FRAC_ACTS
    snap = T.pkg.synth.make_crowded_snapshot(2 + seed % 9, 9300 + seed, fill=0.3 + 0.5 * (seed % 5) / 4, n_pending_jobs=6 + seed % 13, elastic_frac=0.2,
                                             hog_frac=0.5, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], cpu_only_frac=0.3 if seed % 4 == 0 else 0.0)
    T.pkg.synth.add_fractions(snap, seed, frac=0.6, portions=(0.25, 0.5, 0.75))
    cfg = T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], cpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[(seed // 2) % 2], k_value=(0.0, 0.5, 1.0)[seed % 3],
                               max_consolidation_preemptees=(-1, 16, 2)[seed % 3])
    if seed % 3 == 0: cfg.plugins = (cfg.plugins & ~T.abi.PLUGINS["gpupack"]) | T.abi.PLUGINS["gpuspread"]
    if seed % 7 == 0: cfg.plugins &= ~T.abi.PLUGINS["gpusharingorder"]
    for acts in (("allocate",), FRAC_ACTS[seed % len(FRAC_ACTS)]):
        ref = T.Oracle.run(snap, cfg, acts)
        res = run_gpu(snap, cfg, acts)
        assert_same_tol(res, ref)
        _same_groups(snap, res, ref)


@pytest.mark.parametrize("idx,scale", [(1, 1.0), (2, 0.1), (4, 0.01)])
def test_gpu_a_run_of_one_class_follows_its_node(gpu, idx, scale):
    """The staged job path of the sequential engine on the device (kai_engine.hpp allocate_job_fast, round 6): consecutive tasks of one scan class keep landing on the node the
    class's arg-max pointed to while that node's key does not drop, without a query of the index (its refresh is published once per run).  Same operations, states and shares
    as the oracle and as the path without the staged jobs (engine_mode 2: one query per decision); fewer index queries than that path."""
    snap, cfg, _ = T.pkg.synth.config(idx, scale)
    ref = T.Oracle.run(snap, cfg, ("allocate",))
    cfg.engine_mode = 3
    res = run_gpu(snap, cfg, ("allocate",))
    assert_same(res, ref)
    cfg.engine_mode = 2
    gen = run_gpu(snap, cfg, ("allocate",))
    assert_same(gen, ref)
    assert int(res.stats.reserved[0]) < int(gen.stats.reserved[0])  # index queries


@pytest.mark.parametrize("level", (0, 1, 2))
@pytest.mark.parametrize("seed", range(0, 24, 3))
def test_gpu_shared_gpus_keep_the_class_index(gpu, seed, level, monkeypatch):
    """Round 6: a session with shared GPUs keeps the arg-max index of every class that asks for no fraction (the gpusharingorder score as one more key bit, the node's summary
    bits kept by SgNode::refit), fraction pods are brute-force passes; KAI_SHARED_INDEX 0 / 1 / 2 = no index / index / index + staged job path (default).  Same clusters as
    tests/test_engine_hostsim.py::test_hostsim_shared_gpus_keep_the_class_index, on the device against the oracle; allocate, then a full cycle on the same snapshot."""
    from test_engine_hostsim import _same_groups, FRAC_ACTS
    monkeypatch.setenv("KAI_SHARED_INDEX", str(level))
    S = T.pkg.synth
    snap = S.make_snapshot(40 + 17 * (seed % 7), 700 + 90 * (seed % 5), 5100 + seed, queue_levels=((2, 3), (3,), (2, 2, 2))[seed % 3], prefill=(0.2, 0.5, 0.8)[seed % 3],
                           gpu_mix=((8, .6), (4, .2), (0, .2)), cpu_only_frac=0.25, limits_frac=0.2 if seed % 2 else 0.0)
    S.add_fractions(snap, seed, frac=(0.3, 0.7)[seed % 2], portions=(0.25, 0.5, 0.75), memory_requests=0.3 if seed % 4 == 3 else 0.0)
    cfg = T.abi.default_config(k_value=(0.0, 0.5, 1.0)[seed % 3], gpu_strategy=T.abi.SPREAD if seed % 6 == 5 else T.abi.BINPACK, max_consolidation_preemptees=16)
    if seed % 4 == 3: cfg.min_node_gpu_memory = 100
    if seed % 5 == 0: cfg.plugins = (cfg.plugins & ~T.abi.PLUGINS["gpupack"]) | T.abi.PLUGINS["gpuspread"]
    if seed % 7 == 3: cfg.plugins &= ~T.abi.PLUGINS["gpusharingorder"]
    for acts in (("allocate",), FRAC_ACTS[seed % len(FRAC_ACTS)]) if level == 2 else (("allocate",),):  # (the victim actions never read the index: one level carries the full cycle)
        ref = T.Oracle.run(snap, cfg, acts)
        res = run_gpu(snap, cfg, acts)
        assert_same_tol(res, ref)
        _same_groups(snap, res, ref)
        if level > 0 and acts == ("allocate",) and int(res.stats.decisions) > 200: assert int(res.stats.node_scans) < int(res.stats.decisions) // 2


@pytest.mark.parametrize("level", (2, 1))
def test_gpu_config3_with_fractions_hashes_to_the_oracles(gpu, level, monkeypatch):
    """BASELINE config 3 with 30 % of its one-GPU pods as fractions of a device (bench.py's other_shapes.c3_fractions_30) at full size: the operations must hash to the ORACLE's
    end-to-end run (profiles/full_size_pins.json C3fractions30) with the class index kept for the other classes — 2 202 passes over the nodes instead of one per decision."""
    import json, os
    monkeypatch.setenv("KAI_SHARED_INDEX", str(level))
    with open(os.path.join(T.ROOT, "profiles", "full_size_pins.json")) as f:
        pin = json.load(f)["C3fractions30"]
    snap, cfg, desc = T.pkg.synth.config(2, 1.0)
    T.pkg.synth.add_fractions(snap, 7, frac=0.3)
    assert (snap.n_nodes, snap.n_pods) == (pin["nodes"], pin["pods"])
    res = run_gpu(snap, cfg)
    assert len(res.ops) == pin["ops"] and T.ops_sha256(res.ops) == pin["ops_sha256"]
    assert int(res.stats.node_scans) < int(res.stats.decisions) // 10


@pytest.mark.parametrize("seed,ci", ((5592, 5), (8881176, 5), (31415658, 5)))
def test_gpu_victim_tasks_keep_their_eviction_order(gpu, seed, ci):
    """The two campaign cycles with fractions under allocate + consolidation + reclaim + preempt that differed from the oracle in rounds 1-2
    (tests/test_engine_hostsim.py::test_hostsim_victim_tasks_keep_their_eviction_order has the story)."""
    from test_engine_hostsim import _same_groups, fraction_campaign_case
    snap, cfg, acts = fraction_campaign_case(seed, ci)
    ref = T.Oracle.run(snap, cfg, acts)
    res = run_gpu(snap, cfg, acts)
    assert_same_tol(res, ref)
    _same_groups(snap, res, ref)


@pytest.mark.parametrize("wgs", (1, 2, 7, 64, 200))
def test_gpu_victim_search_on_several_workgroups(gpu, wgs, monkeypatch):
    """The victim actions on 1 / 2 / 7 / 64 / 200 workgroups of the MI355X (one replica of the session arrays each, the simulations of a partial job dealt out in waves,
    kai_engine_solver.inc solve_partial_multi): BASELINE config 4 at 2 % and two crowded clusters — the oracle's operations, Statement numbers, states and shares whatever
    the width; stats.reserved[1] says how many workgroups ran the action."""
    monkeypatch.setenv("KAI_VICTIM_WGS", str(wgs))
    snap, cfg, _ = T.pkg.synth.config(3, 0.02)
    acts = ("allocate", "consolidation", "reclaim")
    ref = T.Oracle.run(snap, cfg, acts)
    res = run_gpu(snap, cfg, acts)
    assert_same(res, ref)
    assert int(res.stats.reserved[1]) == wgs, int(res.stats.reserved[1])  # the last action (reclaim) ran on that many workgroups
    for seed in (3, 10):
        snap = T.pkg.synth.make_crowded_snapshot(6 + seed, 7700 + seed, fill=0.9, n_pending_jobs=6 + seed, elastic_frac=0.25, hog_frac=0.5, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3])
        cfg = T.abi.default_config(max_consolidation_preemptees=-1, k_value=0.5)
        acts = ("allocate", "consolidation", "reclaim", "preempt")
        assert_same(run_gpu(snap, cfg, acts), T.Oracle.run(snap, cfg, acts))


@pytest.mark.parametrize("seed", range(24))
def test_gpu_gpu_memory_fuzz(gpu, seed):
    """Requests for MiB of one device (ABI v5 pod_gpu_memory) beside fractions and whole GPUs on the device: the memory they take on a shared GPU, their
    accepted quota (ceil to 1/100 of a device), their weight while pending (memory / MinNodeGPUMemory), the memory term of consolidation's gate."""
    from test_engine_hostsim import _same_groups, FRAC_ACTS
    snap = T.pkg.synth.make_crowded_snapshot(2 + seed % 9, 9900 + seed, fill=0.3 + 0.5 * (seed % 5) / 4, n_pending_jobs=6 + seed % 13, elastic_frac=0.2,
                                             hog_frac=0.5, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], cpu_only_frac=0.3 if seed % 4 == 0 else 0.0)
    T.pkg.synth.add_fractions(snap, seed, frac=0.7, memory_requests=(0.5, 1.0)[seed % 2], gpu_memory=(100, 200, 16300)[seed % 3], portions=(0.25, 0.5, 0.75))
    cfg = T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], k_value=(0.0, 0.5, 1.0)[seed % 3])
    cfg.min_node_gpu_memory = (100, 200, 16300)[seed % 3] if seed % 5 else 100
    if seed % 3 == 0: cfg.plugins = (cfg.plugins & ~T.abi.PLUGINS["gpupack"]) | T.abi.PLUGINS["gpuspread"]
    for acts in (("allocate",), FRAC_ACTS[seed % len(FRAC_ACTS)]):
        ref = T.Oracle.run(snap, cfg, acts)
        res = run_gpu(snap, cfg, acts)
        assert_same_tol(res, ref)
        _same_groups(snap, res, ref)


@pytest.mark.parametrize("seed", range(16))
def test_gpu_mig_fuzz(gpu, seed):
    """MIG nodes and MIG requests on the device (ABI v5 res_mig_*), see test_hostsim_mig_fuzz."""
    from test_engine_hostsim import FRAC_ACTS
    snap = T.pkg.synth.make_crowded_snapshot(3 + seed % 9, 4400 + seed, fill=0.3 + 0.5 * (seed % 5) / 4, n_pending_jobs=6 + seed % 13, elastic_frac=0.2,
                                             hog_frac=0.5, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], cpu_only_frac=0.3 if seed % 4 == 0 else 0.0)
    T.pkg.synth.add_mig(snap, seed, node_frac=(0.3, 0.6, 1.0)[seed % 3], pod_frac=(0.5, 0.9)[seed % 2], legacy_frac=(0.0, 0.05, 0.2)[seed % 3])
    cfg = T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], k_value=(0.0, 0.5, 1.0)[seed % 3], max_consolidation_preemptees=(-1, 16, 2)[seed % 3])
    for acts in (("allocate",), FRAC_ACTS[seed % len(FRAC_ACTS)]):
        ref = T.Oracle.run(snap, cfg, acts)
        res = run_gpu(snap, cfg, acts)
        assert_same_tol(res, ref)


# ------------------------------------------------------------------------------------------------ node-axis sharding on the device
@pytest.mark.parametrize("world,offers", [(2, 0), (3, 16)])
def test_gpu_node_sharded_group_on_one_device(gpu, world, offers):
    """The device code of the node-sharded batch path (offers, virtual cluster, virtual fill, scatter-back: SURVEY 8e) with `world` handles — one
    per rank, each on its own thread — on THIS box's single GPU; the group's all-gather is done by the test with device-to-device copies, where
    the product uses torch.distributed over RCCL.  (The multi-process form of the same protocol runs over gloo in tests/test_dist_gloo.py.)
    Every rank must commit exactly the one-rank operations."""
    import ctypes as C
    import threading
    from test_batch_path import regular_snapshot
    hip = T.pkg.core._hip_runtime()
    cases = [T.pkg.synth.config(1, 0.3)[:2] + (("allocate",),), T.pkg.synth.config(2, 0.05)[:2] + (("allocate",),), T.pkg.synth.config(4, 0.01)[:2] + (("allocate",),),
             (regular_snapshot(5), T.abi.default_config(gpu_strategy=T.abi.SPREAD, k_value=0.5), ("allocate",)),
             # what the group does not shard runs replicated on every rank: BASELINE config 4 (topology gangs; allocate, consolidation, reclaim) and a crowded
             # cluster with an allocate AFTER the victim actions (sharded fill, replicated engine, ...)
             T.pkg.synth.config(3, 0.004)[:2] + (("allocate", "consolidation", "reclaim"),),
             (T.pkg.synth.make_crowded_snapshot(8, 1003, elastic_frac=0.0), T.abi.default_config(max_consolidation_preemptees=-1), ("allocate", "reclaim", "preempt", "allocate"))]
    for snap, cfg, actions in cases:
        ref = T.Oracle.run(snap, cfg, actions)
        barrier = threading.Barrier(world)
        sends, recvs = [None] * world, [None] * world
        results, errors = [None] * world, []

        def make_allgather(rank):
            def allgather(send, recv, nbytes):
                sends[rank], recvs[rank] = send, recv
                barrier.wait()
                for q in range(world):  # this rank's message into every rank's receive buffer
                    assert hip.hipMemcpy(C.c_void_p(recvs[q] + rank * nbytes), C.c_void_p(send), C.c_size_t(nbytes), 3) == 0
                import torch; torch.cuda.synchronize()
                barrier.wait()
                return 0
            return allgather

        def run(rank):
            try:
                with T.pkg.KaiCore(cfg, world=world, rank=rank, offers_per_class=offers, allgather=make_allgather(rank)) as core:
                    ssn = core.open_session(snap)
                    ops = [(int(o["kind"]), int(o["pod"]), int(o["node"]), int(o["job"])) for a in actions for o in ssn.execute(a)]
                    st, nd = ssn.pod_states()
                    results[rank] = (ops, st, nd, ssn.node_states(), ssn.queue_shares(), ssn.stats())
                    ssn.close()
            except Exception as e:  # noqa: BLE001 — reported below; the other threads are released
                errors.append((rank, repr(e))); barrier.abort()

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads: t.start()
        for t in threads: t.join(timeout=300)
        assert not errors, errors
        for rank in range(world):
            ops, st, nd, nodes, shares, stats = results[rank]
            assert ops == ref.ops
            assert (st == ref.pod_status).all() and (nd == ref.pod_node).all()
            for k in ref.nodes: assert np.array_equal(nodes[k], ref.nodes[k]), k
            for k in ref.shares_final: assert np.array_equal(shares[k], ref.shares_final[k]), k
            if actions == ("allocate",):
                assert stats.reserved[4] >= 1 and stats.reserved[0] >= 1  # batch rounds, exchanges


@pytest.mark.parametrize("world,wgs,cap", [(2, 32, 0), (3, 8, 7), (2, 2, 1024)])
def test_gpu_victim_waves_over_the_ranks_of_a_group_on_one_device(gpu, world, wgs, cap, monkeypatch):
    """The victim actions of a node-sharded group with their simulation waves dealt out over the ranks (kai_victim_shard.hpp): `world` handles, one per rank and thread, on THIS
    box's single GPU — each action's kernel runs on `wgs` workgroups per rank and rings its mailbox at every wave's end; the host thread inside kai_action_execute copies the
    wave out of HBM, all-gathers (here: a barrier and memmove between the threads, host memory; the product: kai_shard_attach_host's collective or the library's RCCL
    communicator), merges and answers.  Every rank must commit exactly the oracle's operations, and collectives must have happened.  (The multi-process form of the same
    protocol runs over gloo on the emulator: tests/test_dist_gloo.py.)"""
    import ctypes as C
    import threading
    monkeypatch.setenv("KAI_VICTIM_WGS", str(wgs))
    if cap: monkeypatch.setenv("KAI_VICTIM_XCAP", str(cap))
    cases = [T.pkg.synth.config(3, 0.01)[:2] + (("allocate", "consolidation", "reclaim"),),
             (T.pkg.synth.make_crowded_snapshot(8, 1003, elastic_frac=0.0), T.abi.default_config(max_consolidation_preemptees=-1), ("allocate", "reclaim", "preempt", "allocate")),
             (T.pkg.synth.make_crowded_snapshot(15, 7703, fill=0.9, n_pending_jobs=12, elastic_frac=0.25, hog_frac=0.5, queue_levels=(2, 2)), T.abi.default_config(max_consolidation_preemptees=16), ("consolidation", "reclaim", "preempt"))]
    for snap, cfg, actions in cases:
        ref = T.Oracle.run(snap, cfg, actions)
        barrier = threading.Barrier(world)
        sends, recvs = [None] * world, [None] * world
        results, errors = [None] * world, []

        def make_host_allgather(rank):
            def allgather(send, recv, nbytes):  # host memory of the library; must not touch the device (the action's kernel waits for this to return)
                sends[rank], recvs[rank] = send, recv
                barrier.wait(timeout=120)
                for q in range(world): C.memmove(recvs[q] + rank * nbytes, send, nbytes)
                barrier.wait(timeout=120)
                return 0
            return allgather

        def make_allgather(rank):  # the sharded fill's exchange (device memory), as in test_gpu_node_sharded_group_on_one_device
            hip = T.pkg.core._hip_runtime()
            def allgather(send, recv, nbytes):
                sends[rank], recvs[rank] = send, recv
                barrier.wait(timeout=120)
                for q in range(world): assert hip.hipMemcpy(C.c_void_p(recvs[q] + rank * nbytes), C.c_void_p(send), C.c_size_t(nbytes), 3) == 0
                import torch; torch.cuda.synchronize()
                barrier.wait(timeout=120)
                return 0
            return allgather

        def run(rank):
            try:
                with T.pkg.KaiCore(cfg, world=world, rank=rank, offers_per_class=16, allgather=make_allgather(rank), host_allgather=make_host_allgather(rank)) as core:
                  try:
                    ssn = core.open_session(snap)
                    ops, exchanges = [], 0
                    for a in actions:
                        ops += [(int(o["kind"]), int(o["pod"]), int(o["node"]), int(o["job"])) for o in ssn.execute(a)]
                        if a != "allocate": exchanges += int(ssn.stats().reserved[7])
                    st, nd = ssn.pod_states()
                    results[rank] = (ops, st, nd, ssn.node_states(), ssn.queue_shares(), exchanges)
                    ssn.close()
                  except BaseException as e:  # (said before the handle is destroyed: a failing rank used to take the process down there without a word)
                    import sys; print(f"rank {rank} failed inside the session: {e!r}", file=sys.stderr, flush=True); barrier.abort(); raise
            except BaseException as e:  # noqa: BLE001 — reported below; the other threads are released
                import sys; print(f"rank {rank}: {e!r}", file=sys.stderr, flush=True)
                errors.append((rank, repr(e))); barrier.abort()

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads: t.start()
        for t in threads: t.join(timeout=600)
        assert not errors, errors
        n_victim = sum(1 for a in actions if a != "allocate")
        for rank in range(world):
            ops, st, nd, nodes, shares, exchanges = results[rank]
            assert ops == ref.ops
            assert (st == ref.pod_status).all() and (nd == ref.pod_node).all()
            for k in ref.nodes: assert np.array_equal(nodes[k], ref.nodes[k]), k
            for k in ref.shares_final: assert np.array_equal(shares[k], ref.shares_final[k]), k
            assert exchanges == results[0][5] and exchanges >= n_victim  # the same collectives on every rank; at least every action's closing message
        assert results[0][5] > n_victim or not ref.ops  # waves were exchanged


def test_gpu_default_allgather_plumbing(gpu):
    """KaiCore's default exchange step — stage into torch tensors, torch.distributed.all_gather_into_tensor (backend nccl = RCCL), stage back — on
    this box's one GPU with a one-rank group: what a node-sharded group calls between kernels, minus the second GPU."""
    import torch
    import torch.distributed as dist
    os_env = __import__("os").environ
    os_env.setdefault("MASTER_ADDR", "127.0.0.1"); os_env.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        core = T.pkg.KaiCore(T.abi.default_config())
        core.world, core._user_allgather, core._stage = 1, None, None
        a = torch.arange(4096, dtype=torch.uint8, device="cuda"); b = torch.zeros(4096, dtype=torch.uint8, device="cuda")
        assert core._allgather(None, a.data_ptr(), b.data_ptr(), 4096) == 0
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        core.destroy()
    finally:
        dist.destroy_process_group()


def test_gpu_exchange_from_the_library_over_rccl(gpu):
    """kai_shard_attach_rccl: the library's own RCCL communicator (librccl resolved at run time, ncclGetUniqueId → ncclCommInitRank) and the group's exchange step as an
    ncclAllGather on the library's stream (kai_shard_allgather_probe = what the sharded fill calls between kernels) — on this box's one GPU as a group of one rank.
    The session behind such a handle runs as usual."""
    import ctypes as C
    import torch
    core = T.pkg.KaiCore(T.abi.default_config(), allgather="rccl")
    try:
        a = torch.arange(8192, dtype=torch.uint8, device="cuda") ; b = torch.zeros(8192, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        assert core.lib.kai_shard_allgather_probe(core.handle, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), 8192) == 0, core.lib.kai_last_error(core.handle)
        assert torch.equal(a, b)
        snap, cfg, _ = T.pkg.synth.config(0, 1.0)
        ssn = core.open_session(snap)
        ops = ssn.execute("allocate")
        assert len(ops) == len(T.Oracle.run(snap, cfg).ops)
    finally:
        core.destroy()


@pytest.mark.parametrize("kind,nodes,pods", [("fractions", 1500, 9000), ("gpu_memory", 700, 5000), ("mig", 1500, 9000)])
def test_gpu_shared_devices_and_mig_at_scale(gpu, kind, nodes, pods):
    """Shared devices, GPU-memory requests and MIG rows beyond the few-node fuzz cases: such snapshots scan every decision by brute force over all node blocks (no
    class index), so this is the scanner's block reduction with the shared-GPU fit, the gpusharingorder score and the MIG predicates on hundreds of blocks."""
    from test_engine_hostsim import _same_groups
    snap = T.pkg.synth.make_snapshot(nodes, pods, 7700 + nodes, queue_levels=(2, 3), prefill=0.4, gpu_mix=((8, .6), (4, .3), (0, .1)), gpus_per_pod=(1, 1, 2, 4), cpu_only_frac=0.1,
                                     limits_frac=0.2, queue_prios=(100, 200), oqws=(1.0, 2.0), nonpreempt_frac=0.1)
    if kind == "mig": T.pkg.synth.add_mig(snap, 5, node_frac=0.4, pod_frac=0.6, legacy_frac=0.02)
    else: T.pkg.synth.add_fractions(snap, 5, frac=0.7, portions=(0.25, 0.5, 0.75), memory_requests=0.6 if kind == "gpu_memory" else 0.0)
    cfg = T.abi.default_config(k_value=0.5)
    ref = T.Oracle.run(snap, cfg, ("allocate",), threads=8)
    res = run_gpu(snap, cfg, ("allocate",))
    assert len(ref.ops) > 500
    assert_same_tol(res, ref)
    if kind != "mig": _same_groups(snap, res, ref)


@pytest.mark.parametrize("kind,n", [("fractions", 100), ("mig", 100), ("gpu_memory", 60)])
def test_gpu_victim_actions_with_shared_devices_and_mig_on_crowded_clusters(gpu, kind, n):
    """A full cycle (allocate, consolidation, reclaim, preempt) on a crowded cluster of 60–100 nodes with fraction pods, GPU-memory requests or MIG: hundreds of
    operations, > 100 evictions; everything identical to the oracle (shares to 1e-9 where the quantities are not integers)."""
    from test_engine_hostsim import _same_groups
    snap = T.pkg.synth.make_crowded_snapshot(n, 4100 + n, fill=0.85, n_pending_jobs=60, elastic_frac=0.2, hog_frac=0.5, queue_levels=(2, 2, 2))
    if kind == "mig": T.pkg.synth.add_mig(snap, 9, node_frac=0.5, pod_frac=0.6, legacy_frac=0.02)
    else: T.pkg.synth.add_fractions(snap, 9, frac=0.6, portions=(0.25, 0.5, 0.75), memory_requests=0.5 if kind == "gpu_memory" else 0.0)
    cfg = T.abi.default_config(max_consolidation_preemptees=16, k_value=0.5)
    acts = ("allocate", "consolidation", "reclaim", "preempt")
    ref = T.Oracle.run(snap, cfg, acts)
    assert sum(1 for o in ref.ops if o[0] == 2) > 100
    res = run_gpu(snap, cfg, acts)
    assert_same_tol(res, ref)
    if kind != "mig": _same_groups(snap, res, ref)
