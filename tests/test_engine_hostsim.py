"""Debug aid: the engine's control flow (kai_engine.hpp) compiled for the host must match the oracle bit for bit.

This is NOT the parity claim (that is tests/test_gpu_parity.py on a real MI355X through the C ABI); it exists because
the development container has no GPU and the control flow is the part that needs a debugger.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kai_testlib as T


class HostSim(T.Oracle):
    _sim = None

    @classmethod
    def lib(cls):
        if cls._sim is None:
            so = os.environ.get("KAI_HOSTSIM_SO") or os.path.join(T.ROOT, "tests", "host_sim", "libhostsim.so")  # override: a -DKAI_SOLVER_TRACE build
            src = os.path.join(T.ROOT, "tests", "host_sim", "host_sim.cpp")
            import glob
            deps = [src] + glob.glob(os.path.join(T.ROOT, "tests", "host_sim", "*.hpp")) + glob.glob(os.path.join(T.ROOT, "kai-scheduler_amd", "csrc", "*.hpp")) + glob.glob(os.path.join(T.ROOT, "kai-scheduler_amd", "csrc", "*.inc")) + glob.glob(os.path.join(T.ROOT, "include", "*.h"))
            if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
                subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-pthread", "-o", so, src])
            raw = C.CDLL(so)
            raw.kai_hostsim_run.restype = C.c_int
            raw.kai_hostsim_last_exchanges.restype = C.c_int64
            cls._raw = raw

            class L:
                kai_oracle_run = raw.kai_hostsim_run
                kai_oracle_last_gpu_groups = raw.kai_hostsim_last_gpu_groups
            cls._sim = L
        return cls._sim


def assert_same(res, ref, share_tol=0.0):
    """share_tol: with fractional GPU quantities the queue sums are no longer sums of integers, so their last bits depend on the order of
    addition — the reference itself ranges Go maps there (proportion.go:347-401); the engine rolls pods up job by job, the oracle pod by pod.
    Placements stay exact; the shares are held to the task's 1e-6 (here 1e-9)."""
    assert res.ops == ref.ops
    if getattr(res, "stmts", None) is not None and getattr(ref, "stmts", None) is not None:
        assert res.stmts == ref.stmts  # Statement boundaries (kai_op.stmt)
    assert (res.pod_status == ref.pod_status).all() and (res.pod_node == ref.pod_node).all()
    same = (lambda a, b: np.array_equal(a, b)) if share_tol == 0.0 else (lambda a, b: np.allclose(a, b, rtol=0.0, atol=share_tol))
    for k in ref.shares_open:
        assert same(res.shares_open[k], ref.shares_open[k]), k
        assert same(res.shares_final[k], ref.shares_final[k]), k
    for k in ref.nodes:
        assert np.array_equal(res.nodes[k], ref.nodes[k]), k


GOLD_FILES = ("allocate__allocate", "allocate__allocateGang", "allocate__allocateElastic", "allocate__allocate_subgroups", "allocate__allocateTopology",
              "reclaim__reclaim", "reclaim__reclaimDepartments", "reclaim__reclaimGang", "reclaim__reclaim_elastic", "reclaim__reclaim_sub_group",
              "preempt__preempt", "preempt__preemptGang", "preempt__preempt_elastic", "preempt__preempt_subgroups",
              "consolidation__consolidation", "consolidation__consolidation_subgroups")
GOLD = [(n, i, c, T.load_golden(n)["actions"]) for n in GOLD_FILES
        for i, c in enumerate(T.load_golden(n)["cases"])]


@pytest.mark.parametrize("name,i,case,actions", GOLD, ids=[f"{n}[{i}]" for n, i, _, _ in GOLD])
def test_hostsim_golden(name, i, case, actions):
    try:
        snap, cfg, meta = T.case_to_snapshot(case)
    except T.Unsupported as e:
        pytest.skip(str(e))
    res = HostSim.run(snap, cfg, actions)
    assert not T.check_expectations(snap, meta, res.pod_status, res.pod_node, res.nodes)
    assert_same(res, T.Oracle.run(snap, cfg, actions))


@pytest.mark.parametrize("idx,scale", [(0, 1.0), (1, 0.2), (2, 0.03), (4, 0.004)])
def test_hostsim_synthetic(idx, scale):
    snap, cfg, _ = T.pkg.synth.config(idx, scale)
    assert_same(HostSim.run(snap, cfg), T.Oracle.run(snap, cfg))


@pytest.mark.parametrize("seed", range(6))
def test_hostsim_random_small(seed):
    rng = np.random.default_rng(seed)
    snap = T.pkg.synth.make_snapshot(int(rng.integers(1, 40)), int(rng.integers(0, 300)), 1000 + seed, queue_levels=(2, 3), prefill=float(rng.random()) * 0.8,
                                     gpu_mix=((8, .5), (4, .3), (0, .2)), cpu_only_frac=0.3, zipf=True, limits_frac=0.3, queue_prios=(100, 200),
                                     oqws=(1.0, 2.0), nonpreempt_frac=0.2, usage_max=0.2, lexi_names=bool(seed % 2))
    for strat in (T.abi.BINPACK, T.abi.SPREAD):
        cfg = T.abi.default_config(gpu_strategy=strat, cpu_strategy=strat, k_value=float(seed % 3) * 0.5)
        assert_same(HostSim.run(snap, cfg), T.Oracle.run(snap, cfg))


@pytest.mark.parametrize("seed", range(24))
def test_hostsim_victim_actions_crowded_cluster(seed):
    snap = T.pkg.synth.make_crowded_snapshot(6 + seed % 17, 1000 + seed, fill=0.85 + 0.1 * (seed % 2), queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3],
                                             cpu_only_frac=0.2 if seed % 5 == 0 else 0.0)
    cfg = T.abi.default_config(max_consolidation_preemptees=-1 if seed % 2 else 16)
    cfg.use_scheduling_signatures = seed % 2; cfg.allow_consolidating_reclaim = int(seed % 3 != 0)  # MinimalJobRepresentatives on for odd seeds
    evictions = 0
    for actions in (("reclaim",), ("preempt",), ("consolidation",), ("allocate", "consolidation", "reclaim", "preempt")):
        ref = T.Oracle.run(snap, cfg, actions)
        assert_same(HostSim.run(snap, cfg, actions), ref)
        evictions += sum(1 for o in ref.ops if o[0] == 2)
    assert evictions > 0 or seed % 17 < 2  # the generator must actually exercise the victim search


@pytest.mark.parametrize("scale", [0.003, 0.01])
def test_hostsim_config4_topology_consolidation_reclaim(scale):
    """BASELINE config 4 (scaled): zone/rack topology constraints on the pending gangs, a cluster 85 % full of preemptible Running jobs,
    one cycle = allocate, consolidation, reclaim."""
    snap, cfg, _ = T.pkg.synth.config(3, scale)
    acts = ("allocate", "consolidation", "reclaim")
    assert_same(HostSim.run(snap, cfg, acts), T.Oracle.run(snap, cfg, acts))


import test_oracle_golden as _G
INTEG_FILES = _G.INTEG_FILES  # incl. the fraction, GPU-memory and MIG tables (allocateFractionalGpu, allocateMIG, consolidationFractional, preempt/reclaim Fractional and MIG)
INTEG = [(n, i, c) for n in INTEG_FILES for i, c in enumerate(T.load_golden(n)["cases"])]


@pytest.mark.parametrize("name,i,case", INTEG, ids=[f"{n}[{i}]" for n, i, _ in INTEG])
def test_hostsim_integration_rounds(name, i, case):
    def run_both(snap, cfg, actions):
        res = HostSim.run(snap, cfg, actions)
        ref = T.Oracle.run(snap, cfg, actions)
        assert_same(res, ref, share_tol=1e-9)
        if "pod_gpu_portion" in snap.arrays: _same_groups(snap, res, ref)
        return res
    try:
        errs = T.run_integration(case, run_both, rounds_after=1, fractions=True)
    except T.Unsupported as e:
        pytest.skip(str(e))
    assert not errs, errs[:4]


@pytest.mark.parametrize("seed", range(16))
def test_hostsim_minruntime_protection(seed):
    """minruntime plugin (plugins/minruntime): start times up to 2 h before "now", min-runtimes of 0 / 30 / 60 min on half of the queues, both
    reclaim resolve methods — victim filters and the elastic-victim scenario validators must match the oracle, and protection must matter."""
    snap = T.pkg.synth.make_crowded_snapshot(6 + seed % 17, 1000 + seed, fill=0.9, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], minruntime=True)
    cfg = T.abi.default_config(max_consolidation_preemptees=-1); cfg.use_scheduling_signatures = seed % 2
    cfg.now_ns = T.pkg.synth.NOW_NS; cfg.default_preempt_min_runtime_ns = 0 if seed % 4 else 900 * 10**9
    cfg.default_reclaim_min_runtime_ns = 600 * 10**9 if seed % 3 == 0 else 0; cfg.reclaim_resolve_method = seed % 2
    for actions in (("reclaim",), ("preempt",), ("allocate", "consolidation", "reclaim", "preempt")):
        assert_same(HostSim.run(snap, cfg, actions), T.Oracle.run(snap, cfg, actions))


@pytest.mark.parametrize("seed", range(24))
def test_hostsim_replica_groups_with_required_rack(seed):
    """Two sub-groups of one job on the same required topology level (the replica fixtures of allocateTopology_test.go) in a crowded
    cluster: allocation through the sub-group DFS and the victim actions with the TopologyAwareIdleGpus scenario filter in play."""
    snap = T.pkg.synth.make_crowded_snapshot(6 + seed % 13, 3000 + seed, fill=0.8 + 0.1 * (seed % 2), queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3],
                                             two_podsets_frac=0.6, n_pending_jobs=14)
    T.pkg.synth.add_replica_topology(snap, seed, zones=2, nodes_per_rack=2 + seed % 2)
    cfg = T.abi.default_config(max_consolidation_preemptees=-1); cfg.use_scheduling_signatures = seed % 2
    for actions in (("allocate",), ("reclaim",), ("preempt",), ("consolidation",), ("allocate", "consolidation", "reclaim", "preempt")):
        assert_same(HostSim.run(snap, cfg, actions), T.Oracle.run(snap, cfg, actions))


@pytest.mark.parametrize("seed", [9, 31, 36, 43, 123, 234, 1623] + list(range(2000, 2060)))
def test_hostsim_broad_random_cycles(seed):
    """Seeds of the broad randomized campaign (tests/kai_testlib.py::broad_case), incl. the ones that exposed real divergences: staged job
    path after a victim action, candidate-node order under lexicographic node names, the session job's own tasks cache, leaf heaps
    whose keys change while a job waits.  Every case: operations, pod states, node accounting and queue shares identical to the oracle."""
    for snap, cfg, actions in T.broad_case(seed):
        assert_same(HostSim.run(snap, cfg, actions), T.Oracle.run(snap, cfg, actions))


@pytest.mark.parametrize("scale", [0.02, 0.05])
def test_hostsim_config5_in_the_mixed_shape(scale):
    """BASELINE config 5 as SURVEY 8d writes it down (bench.py --config C5 --mixed), scaled: zone / rack labels (8 x 64 racks: 521 domains, so subSetNodesFn's domain
    loops take their scan-lane forms), 5 % of the gangs with a required rack, 5 % elastic, half the cluster running, minruntime on."""
    snap, cfg, _ = T.pkg.synth.config(4, scale, mixed=True)
    assert_same(HostSim.run(snap, cfg, ("allocate",)), T.Oracle.run(snap, cfg, ("allocate",), threads=8 if snap.n_nodes >= 2048 else 1))


@pytest.mark.parametrize("scale,depth", [(0.02, 3), (0.05, 8)])
def test_hostsim_config4_with_queue_depth(scale, depth):
    """BASELINE config 4 with queueDepthPerAction for the victim actions (framework/session.go:398-404: jobs tried per queue and action; the reference's operator docs configure
    5 .. 15): allocate, consolidation, reclaim against the oracle.  The same shape at 30 % is pinned in profiles/full_size_pins.json (tools/pin_c4_depth.py)."""
    snap, cfg, _ = T.pkg.synth.config(3, scale)
    for a in ("consolidation", "reclaim", "preempt"):
        cfg.queue_depth[T.abi.ACTIONS[a]] = depth
    acts = ("allocate", "consolidation", "reclaim")
    ref = T.Oracle.run(snap, cfg, acts)
    assert any(o[0] == 2 for o in ref.ops)
    assert_same(HostSim.run(snap, cfg, acts), ref)


@pytest.fixture
def domain_loops_on_lanes():
    """subSetNodesFn's loops over the DOMAINS of a topology in the forms the scan lanes of the action kernel run (TopoScan ops 5 .. 14, Engine::topo_dom_body: roll-ups,
    ratios, every child's place among its siblings, paths as numbers, a chosen domain's place in the level order) for every topology, however small; the default
    sends trees of fewer than 16 domains through the control lane's loops."""
    HostSim.lib()
    HostSim._raw.kai_hostsim_set_dom_lanes_min(1)
    yield
    HostSim._raw.kai_hostsim_set_dom_lanes_min(16)


TOPO_GOLD = [g for g in GOLD if g[0] in ("allocate__allocateTopology", "allocate__allocate_subgroups", "consolidation__consolidation_subgroups", "reclaim__reclaim_sub_group")]


@pytest.mark.parametrize("name,i,case,actions", TOPO_GOLD, ids=[f"{n}[{i}]" for n, i, _, _ in TOPO_GOLD])
def test_hostsim_golden_topology_domain_loops_on_lanes(domain_loops_on_lanes, name, i, case, actions):
    """The reference's topology / sub-group tables (allocateTopology_test.go and the sub-group action tests) with the domain loops in their scan-lane forms."""
    try:
        snap, cfg, meta = T.case_to_snapshot(case)
    except T.Unsupported as e:
        pytest.skip(str(e))
    res = HostSim.run(snap, cfg, actions)
    assert not T.check_expectations(snap, meta, res.pod_status, res.pod_node, res.nodes)
    assert_same(res, T.Oracle.run(snap, cfg, actions))


@pytest.mark.parametrize("seed", list(range(24)))
def test_hostsim_replica_groups_domain_loops_on_lanes(domain_loops_on_lanes, seed):
    snap = T.pkg.synth.make_crowded_snapshot(6 + seed % 13, 3000 + seed, fill=0.8 + 0.1 * (seed % 2), queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3],
                                             two_podsets_frac=0.6, n_pending_jobs=14)
    T.pkg.synth.add_replica_topology(snap, seed, zones=2, nodes_per_rack=2 + seed % 2)
    cfg = T.abi.default_config(max_consolidation_preemptees=-1); cfg.use_scheduling_signatures = seed % 2
    for actions in (("allocate",), ("allocate", "consolidation", "reclaim", "preempt")):
        assert_same(HostSim.run(snap, cfg, actions), T.Oracle.run(snap, cfg, actions))


@pytest.mark.parametrize("seed", list(range(2100, 2140)))
def test_hostsim_broad_random_cycles_domain_loops_on_lanes(domain_loops_on_lanes, seed):
    for snap, cfg, actions in T.broad_case(seed):
        assert_same(HostSim.run(snap, cfg, actions), T.Oracle.run(snap, cfg, actions))


# ------------------------------------------------------------------------------------------------ shared GPUs (host twin only so far)
def _same_groups(snap, res, ref):
    """GPU groups up to the naming of the groups created by the run (the reference draws UUIDs): groups of the snapshot by id, new ones by who shares them."""
    new = T.NEW_GPU_GROUP
    a, b = res.gpu_groups, ref.gpu_groups
    assert np.array_equal(np.where(a < new, a, new), np.where(b < new, b, new))
    fwd, bwd = {}, {}
    for p in range(snap.n_pods):
        if a[p] >= new:
            ka, kb = (int(res.pod_node[p]), int(a[p])), (int(ref.pod_node[p]), int(b[p]))
            assert fwd.setdefault(ka, kb) == kb and bwd.setdefault(kb, ka) == ka


FRAC_GOLD = [(i, c) for i, c in enumerate(T.load_golden("allocate__allocateFractionalGpu")["cases"])]
MEM_GOLD = [(i, c) for i, c in enumerate(T.load_golden("allocate__allocateGpuMemory")["cases"])]  # allocateGpuMemory_test.go: requests for MiB of one device
MIG_GOLD = [(i, c) for i, c in enumerate(T.load_golden("allocate__allocateMIG")["cases"])]        # allocateMIG_test.go: MIG instances, legacy MIG tasks


@pytest.mark.parametrize("i,case", FRAC_GOLD + MEM_GOLD + MIG_GOLD, ids=[f"allocateFractionalGpu[{i}]" for i, _ in FRAC_GOLD] + [f"allocateGpuMemory[{i}]" for i, _ in MEM_GOLD] + [f"allocateMIG[{i}]" for i, _ in MIG_GOLD])
def test_hostsim_fractional_goldens(i, case):
    """The engine's control flow with the shared-GPU code compiled in (KAI_SHARED_GPUS, host twin) against the oracle and the reference's
    expectations on allocateFractionalGpu_test.go."""
    snap, cfg, meta = T.case_to_snapshot(case, fractions=True)
    ref = T.Oracle.run(snap, cfg, ("allocate",))
    res = HostSim.run(snap, cfg, ("allocate",))
    assert_same(res, ref, share_tol=1e-9)
    _same_groups(snap, res, ref)
    assert not T.check_expectations(snap, meta, res.pod_status, res.pod_node, res.nodes, res.gpu_groups)


@pytest.mark.parametrize("seed", range(60))
def test_hostsim_fraction_fuzz(seed):
    snap = T.pkg.synth.make_crowded_snapshot(2 + seed % 9, 9300 + seed, fill=0.3 + 0.5 * (seed % 5) / 4, n_pending_jobs=6 + seed % 13, elastic_frac=0.2,
                                             hog_frac=0.5, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], cpu_only_frac=0.3 if seed % 4 == 0 else 0.0)
    T.pkg.synth.add_fractions(snap, seed, frac=0.6)
    cfg = T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], cpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[(seed // 2) % 2], k_value=(0.0, 0.5, 1.0)[seed % 3])
    if seed % 3 == 0: cfg.plugins = (cfg.plugins & ~T.abi.PLUGINS["gpupack"]) | T.abi.PLUGINS["gpuspread"]
    if seed % 7 == 0: cfg.plugins &= ~T.abi.PLUGINS["gpusharingorder"]
    ref = T.Oracle.run(snap, cfg, ("allocate",))
    res = HostSim.run(snap, cfg, ("allocate",))
    assert_same(res, ref, share_tol=1e-9)
    _same_groups(snap, res, ref)


def test_hostsim_callers_high_node_flag_bits_are_ignored():
    """Bits 28-31 of kai_snapshot_soa.node_flags belong to the library (legacy-MIG mark, the two summary bits of a node's shared GPUs): whatever the caller leaves there is
    masked at session open (kai_host_prep.hpp) — a shared-GPU session with garbage in them places exactly as without."""
    S = T.pkg.synth
    def build():
        snap = S.make_snapshot(60, 800, 7311, queue_levels=(2, 3), prefill=0.5, gpu_mix=((8, .6), (4, .2), (0, .2)), cpu_only_frac=0.25)
        S.add_fractions(snap, 3, frac=0.5, portions=(0.25, 0.5, 0.75))
        return snap
    cfg = T.abi.default_config(k_value=0.5)
    clean = HostSim.run(build(), cfg, ("allocate",))
    dirty_snap = build()
    dirty_snap.arrays["node_flags"] = (dirty_snap.arrays["node_flags"].astype(np.uint32) | np.uint32(0xF0000000)).astype(dirty_snap.arrays["node_flags"].dtype)
    dirty_snap.finalize()
    dirty = HostSim.run(dirty_snap, cfg, ("allocate",))
    assert clean.ops == dirty.ops and (clean.pod_node == dirty.pod_node).all()
    assert_same(dirty, T.Oracle.run(build(), cfg, ("allocate",)), share_tol=1e-9)


@pytest.mark.parametrize("idx,scale", [(1, 1.0), (2, 0.1), (4, 0.01)])
def test_hostsim_a_run_of_one_class_follows_its_node(idx, scale):
    """The staged job path of the sequential engine (kai_engine.hpp allocate_job_fast, round 6): consecutive tasks of one scan class keep landing on the node the class's arg-max
    pointed to while that node's key does not drop — without a query of the index, whose refresh is published once per run.  Same operations, states and shares as the oracle and
    as the path without the staged jobs (engine_mode 2: one query per decision); fewer index queries than that path."""
    snap, cfg, _ = T.pkg.synth.config(idx, scale)
    ref = T.Oracle.run(snap, cfg, ("allocate",))
    cfg.engine_mode = 3
    res = HostSim.run(snap, cfg, ("allocate",))
    assert_same(res, ref)
    cfg.engine_mode = 2
    gen = HostSim.run(snap, cfg, ("allocate",))
    assert_same(gen, ref)
    assert int(res.stats.reserved[0]) < int(gen.stats.reserved[0])  # index queries


@pytest.mark.parametrize("level", (0, 1, 2))
@pytest.mark.parametrize("seed", range(24))
def test_hostsim_shared_gpus_keep_the_class_index(seed, level, monkeypatch):
    """Round 6: in a session with shared GPUs the pods that ask for a fraction (or MiB) of a device are in no scan class — brute-force passes over the nodes' GPU groups —
    while every other class keeps its arg-max index, the gpusharingorder score as one more key bit above the others (kai_engine.hpp key_shared_layout, kai_host_prep.hpp
    build_classes).  KAI_SHARED_INDEX = 0: no index (the form before round 6), 1: index, general job path, 2 (default): index + the staged job path for gangs without a
    fraction pod.  Mid-size clusters with CPU-only classes, whole-GPU classes of several sizes and fractions, every level against the oracle; with the index most decisions
    are index queries."""
    monkeypatch.setenv("KAI_SHARED_INDEX", str(level))
    S = T.pkg.synth
    snap = S.make_snapshot(40 + 17 * (seed % 7), 700 + 90 * (seed % 5), 5100 + seed, queue_levels=((2, 3), (3,), (2, 2, 2))[seed % 3], prefill=(0.2, 0.5, 0.8)[seed % 3],
                           gpu_mix=((8, .6), (4, .2), (0, .2)), cpu_only_frac=0.25, limits_frac=0.2 if seed % 2 else 0.0)
    S.add_fractions(snap, seed, frac=(0.3, 0.7)[seed % 2], portions=(0.25, 0.5, 0.75), memory_requests=0.3 if seed % 4 == 3 else 0.0)
    cfg = T.abi.default_config(k_value=(0.0, 0.5, 1.0)[seed % 3], gpu_strategy=T.abi.SPREAD if seed % 6 == 5 else T.abi.BINPACK)
    if seed % 4 == 3: cfg.min_node_gpu_memory = 100
    if seed % 5 == 0: cfg.plugins = (cfg.plugins & ~T.abi.PLUGINS["gpupack"]) | T.abi.PLUGINS["gpuspread"]
    if seed % 7 == 3: cfg.plugins &= ~T.abi.PLUGINS["gpusharingorder"]
    ref = T.Oracle.run(snap, cfg, ("allocate",))
    res = HostSim.run(snap, cfg, ("allocate",))
    assert_same(res, ref, share_tol=1e-9)
    _same_groups(snap, res, ref)
    if level > 0 and int(res.stats.decisions) > 200: assert int(res.stats.node_scans) < int(res.stats.decisions) // 2  # (every pass over the nodes counts, also the ones a kept answer stands for)


@pytest.mark.parametrize("seed", range(60))
def test_hostsim_gpu_memory_fuzz(seed, monkeypatch):
    """Requests for MiB of one device beside fractions (ABI v5 pod_gpu_memory): what they take on a shared device, their accepted quota
    (ceil to 1/100 of the device), their weight while pending (memory / MinNodeGPUMemory) — every action, engine twin against the oracle.
    Even seeds use portions whose quantities are exact in binary; odd seeds use arbitrary ones (0.2, 0.3: 0.3 of a device counts as 0.31) with the
    harness adding the queue sums in the oracle's order, so that the last bit of a sum of non-integers is not what is compared."""
    if seed % 2: monkeypatch.setenv("KAI_HOSTSIM_POD_ORDER_SUMS", "1")
    snap = T.pkg.synth.make_crowded_snapshot(2 + seed % 9, 9900 + seed, fill=0.3 + 0.5 * (seed % 5) / 4, n_pending_jobs=6 + seed % 13, elastic_frac=0.2,
                                             hog_frac=0.5, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], cpu_only_frac=0.3 if seed % 4 == 0 else 0.0)
    T.pkg.synth.add_fractions(snap, seed, frac=0.7, memory_requests=0.5, gpu_memory=(100, 200, 16300)[seed % 3], portions=(0.2, 0.25, 0.3, 0.5, 0.75) if seed % 2 else (0.25, 0.5, 0.75))
    cfg = T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], k_value=(0.0, 0.5, 1.0)[seed % 3])
    cfg.min_node_gpu_memory = (100, 200, 16300)[seed % 3] if seed % 5 else 100
    if seed % 3 == 0: cfg.plugins = (cfg.plugins & ~T.abi.PLUGINS["gpupack"]) | T.abi.PLUGINS["gpuspread"]
    actions = FRAC_ACTS[seed % len(FRAC_ACTS)] if seed % 2 else ("allocate",)
    ref = T.Oracle.run(snap, cfg, actions)
    res = HostSim.run(snap, cfg, actions)
    assert_same(res, ref, share_tol=1e-9)
    _same_groups(snap, res, ref)


@pytest.mark.parametrize("seed", range(40))
def test_hostsim_mig_fuzz(seed):
    """MIG nodes (MigStrategy mixed) and MIG requests (ABI v5 res_mig_*): instances as resource rows, GPU quota by weight while GPUs() stays 0, idle
    instances in the nodes' GPU sums, the predicates of node_info.go:315-359 with legacy MIG tasks — every action, engine twin against the oracle."""
    snap = T.pkg.synth.make_crowded_snapshot(3 + seed % 9, 4400 + seed, fill=0.3 + 0.5 * (seed % 5) / 4, n_pending_jobs=6 + seed % 13, elastic_frac=0.2,
                                             hog_frac=0.5, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3], cpu_only_frac=0.3 if seed % 4 == 0 else 0.0)
    T.pkg.synth.add_mig(snap, seed, node_frac=(0.3, 0.6, 1.0)[seed % 3], pod_frac=(0.5, 0.9)[seed % 2], legacy_frac=(0.0, 0.05, 0.2)[seed % 3])
    cfg = T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], k_value=(0.0, 0.5, 1.0)[seed % 3], max_consolidation_preemptees=(-1, 16, 2)[seed % 3])
    actions = FRAC_ACTS[seed % len(FRAC_ACTS)] if seed % 2 else ("allocate",)
    ref = T.Oracle.run(snap, cfg, actions)
    res = HostSim.run(snap, cfg, actions)
    assert_same(res, ref)


def _fraction_victim_goldens():
    """Golden cases outside allocateFractionalGpu_test.go that hold fraction pods (reclaim / preempt / consolidation / integration tables)."""
    import test_oracle_golden as G
    out = []
    for name, i, case, actions in G.ALL:
        if name in ("allocate__allocateFractionalGpu", "allocate__allocateGpuMemory", "allocate__allocateMIG"): continue  # FRAC_GOLD, MEM_GOLD, MIG_GOLD
        try:
            T.case_to_snapshot(case)
        except T.Unsupported:
            try: T.case_to_snapshot(case, fractions=True)
            except T.Unsupported: continue
            out.append((name, i, case, actions))  # shared devices, GPU-memory requests or MIG: reclaim / preempt / consolidation (GpuMemory, MIG) tables
    return out


FRAC_VICTIM_GOLD = _fraction_victim_goldens()


@pytest.mark.parametrize("name,i,case,actions", FRAC_VICTIM_GOLD, ids=[f"{n}[{i}]" for n, i, _, _ in FRAC_VICTIM_GOLD])
def test_hostsim_fractional_victim_goldens(name, i, case, actions):
    """Victim actions over shared GPUs: eviction of fraction pods, their re-placement on another GPU group of the same node
    (ConsolidateSharedPodInfoToDifferentGPU), statement undo with the previous groups — engine twin against the oracle and the reference's tables."""
    snap, cfg, meta = T.case_to_snapshot(case, fractions=True)
    ref = T.Oracle.run(snap, cfg, actions)
    res = HostSim.run(snap, cfg, actions)
    assert_same(res, ref, share_tol=1e-9)
    _same_groups(snap, res, ref)
    assert not T.check_expectations(snap, meta, res.pod_status, res.pod_node, res.nodes, res.gpu_groups)


FRAC_ACTS = (("reclaim",), ("preempt",), ("consolidation",), ("allocate", "consolidation", "reclaim", "preempt"), ("allocate", "reclaim"), ("allocate", "preempt"), ("reclaim", "preempt"))


@pytest.mark.parametrize("seed", range(70))
def test_hostsim_fraction_victim_fuzz(seed):
    """Crowded clusters with fraction pods under every victim action.  The portions are multiples of 1/4: quota sums stay exact in float64, so the
    exact comparisons of the proportion plugin (resource_quantities.go:59-97) cannot flip on the order of addition (the reference ranges Go maps
    there, so with portions like 0.2 its own outcome at an exact tie is not defined).  Node accounting is compared bit for bit — including what a
    rolled-back simulation leaves behind on a node whose victim was re-placed on another GPU group (the first copy stays booked in the reference)."""
    snap = T.pkg.synth.make_crowded_snapshot(2 + seed % 7, 9500 + seed, fill=0.6 + 0.35 * (seed % 5) / 4, n_pending_jobs=4 + seed % 11, elastic_frac=0.2 * (seed % 2),
                                             hog_frac=0.5, queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3])
    T.pkg.synth.add_fractions(snap, seed, frac=(0.3, 0.6, 0.9)[seed % 3], portions=((0.25, 0.25, 0.5, 0.75), (0.25, 0.5, 0.5, 0.75), (0.5,))[seed % 3])
    cfg = T.abi.default_config(max_consolidation_preemptees=(-1, 16, 2)[seed % 3], gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], k_value=(0.0, 0.5, 1.0)[seed % 3])
    cfg.use_scheduling_signatures = seed % 2
    if seed % 3 == 0: cfg.plugins = (cfg.plugins & ~T.abi.PLUGINS["gpupack"]) | T.abi.PLUGINS["gpuspread"]
    for acts in (FRAC_ACTS[seed % len(FRAC_ACTS)], FRAC_ACTS[3]):
        ref = T.Oracle.run(snap, cfg, acts)
        res = HostSim.run(snap, cfg, acts)
        assert_same(res, ref, share_tol=1e-9)
        _same_groups(snap, res, ref)


@pytest.mark.parametrize("seed", range(3000, 3040))
def test_hostsim_broad_random_cycles_with_fractions(seed):
    """The broad campaign (topology, sub-groups, minruntime, signatures, every action order) with a share of the one-GPU pods turned into fraction
    pods on shared GPUs — tools/host_campaign.py with CAMPAIGN_FRACTIONS=1 ran 9 000 such cycles identical to the oracle."""
    for ci in range(len(T.broad_case(seed))):
        snap, cfg, actions = T.broad_case(seed)[ci]  # the cases of one seed share snapshot objects: take a fresh one before changing it
        T.pkg.synth.add_fractions(snap, seed * 7 + ci, frac=(0.3, 0.6, 0.9)[(seed + ci) % 3], portions=((0.25, 0.5, 0.75), (0.5,), (0.25, 0.25, 0.5))[seed % 3])
        ref, res = T.Oracle.run(snap, cfg, actions), HostSim.run(snap, cfg, actions)
        assert_same(res, ref, share_tol=1e-9)
        _same_groups(snap, res, ref)


FRACTION_VICTIM_ORDER_SEEDS = ((5592, 5), (8881176, 5),  # tools/host_campaign.py CAMPAIGN_FRACTIONS=1: the two cycles of rounds 1-2 that differed from the oracle
                               (31415658, 5))            # round 3: a node's GPU-group table (then 16 slots) overflowed on entries that rolled-back simulations leave booked


def fraction_campaign_case(seed, ci):
    snap, cfg, actions = T.broad_case(seed)[ci]
    cfg.engine_mode = seed % 3 if ci % 2 else 0
    T.pkg.synth.add_fractions(snap, seed * 7 + ci, frac=(0.3, 0.6, 0.9)[(seed + ci) % 3], portions=((0.25, 0.5, 0.75), (0.5,), (0.25, 0.25, 0.5))[seed % 3])
    return snap, cfg, actions


@pytest.mark.parametrize("seed,ci", FRACTION_VICTIM_ORDER_SEEDS)
def test_hostsim_victim_tasks_keep_their_eviction_order(seed, ci):
    """VictimInfo.Tasks and potentialVictimsTasks are SLICES in the order GetTasksToEvict handed the tasks over (base_scenario.go:109-137): proportion's
    splitVictimTasks (proportion.go:187-220) takes the first minAvailable of them as the core tasks, so with pods of different sizes (fractions) the
    reclaim validator's amounts depend on that order.  The engine kept every task group in the canonical pod order only (what ranging the clone's pod
    MAP stands for) and so validated another split: seed 8881176 took another solution, seed 5592 the same operations in another order."""
    snap, cfg, actions = fraction_campaign_case(seed, ci)
    ref, res = T.Oracle.run(snap, cfg, actions), HostSim.run(snap, cfg, actions)
    assert_same(res, ref, share_tol=1e-9)
    _same_groups(snap, res, ref)


@pytest.fixture
def engines():
    """Victim actions of the host-compiled engine on several engines (threads over replicas of the context, kai_engine_solver.inc solve_partial_multi); back to one afterwards.
    The harness itself checks on every run that all engines committed the same operations and ended in the same state."""
    HostSim.lib()
    yield HostSim._raw.kai_hostsim_set_multi
    HostSim._raw.kai_hostsim_set_multi(1)


@pytest.mark.parametrize("seed", range(5200, 5260))
def test_hostsim_victim_search_on_several_engines_broad(engines, seed):
    """The simulations of a partial job dealt out in waves over 2 .. 9 engines: operations, Statement numbers, states, node accounting and shares of the oracle —
    the broad campaign's cases (topology, sub-groups, elastic gangs, minruntime, signatures, every action order)."""
    engines(2 + seed % 8)
    for snap, cfg, actions in T.broad_case(seed):
        ref, res = T.Oracle.run(snap, cfg, actions), HostSim.run(snap, cfg, actions)
        assert_same(res, ref)


@pytest.mark.parametrize("seed", range(24))
def test_hostsim_victim_search_on_several_engines_crowded(engines, seed):
    """Crowded clusters (long victim queues: many scenarios per partial job, rejections by the reclaim validator that leave nodes feasible) on 3 .. 16 engines, and the
    same cycle on one engine: identical operations and the identical statistics (scenarios, simulations, filtered scenarios count what the reference's order reaches)."""
    snap = T.pkg.synth.make_crowded_snapshot(6 + seed % 20, 7700 + seed, fill=0.7 + 0.25 * (seed % 4) / 3, n_pending_jobs=6 + seed % 17, elastic_frac=0.25 * (seed % 3), hog_frac=0.5,
                                             queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3])
    cfg = T.abi.default_config(max_consolidation_preemptees=(-1, 16, 2)[seed % 3], gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], k_value=(0.0, 0.5, 1.0)[seed % 3])
    cfg.use_scheduling_signatures = seed % 2
    acts = FRAC_ACTS[seed % len(FRAC_ACTS)]
    ref = T.Oracle.run(snap, cfg, acts)
    engines(1); one = HostSim.run(snap, cfg, acts)
    engines(3 + seed % 14); res = HostSim.run(snap, cfg, acts)
    assert_same(res, ref); assert_same(one, ref)
    assert (int(res.stats.decisions), int(res.stats.jobs_attempted), int(res.stats.jobs_committed)) == (int(one.stats.decisions), int(one.stats.jobs_attempted), int(one.stats.jobs_committed))


def test_hostsim_config4_on_several_engines(engines):
    """BASELINE config 4 (zone / rack topology, consolidation + reclaim) at 2 %: one engine, eight engines, the oracle."""
    snap, cfg, _ = T.pkg.synth.config(3, 0.02)
    acts = ("allocate", "consolidation", "reclaim")
    ref = T.Oracle.run(snap, cfg, acts)
    engines(8); res = HostSim.run(snap, cfg, acts)
    assert_same(res, ref)
    st = (C.c_int64 * 4)(); HostSim._raw.kai_hostsim_multi_stats(st)
    assert st[0] > 0 and st[2] <= st[1], list(st)  # waves ran; simulations counted <= simulations run


def test_hostsim_config3_full_size_hashes_to_the_oracles_pin():
    """profiles/full_size_pins.json (tools/pin_full_sizes.py: the oracle's full-size runs): BASELINE config 3 at full size on the host-compiled engine gives the pinned
    operation and state hashes (the GPU twin of this test also covers config 5)."""
    import json
    with open(os.path.join(T.ROOT, "profiles", "full_size_pins.json")) as f:
        pin = json.load(f)["C3"]
    snap, cfg, desc = T.pkg.synth.config(2, 1.0)
    assert (desc, snap.n_nodes, snap.n_pods) == (pin["workload"], pin["nodes"], pin["pods"])
    cfg.engine_mode = 3
    res = HostSim.run(snap, cfg)
    assert T.ops_sha256(res.ops) == pin["ops_sha256"] and T.state_sha256(res) == pin["state_sha256"]


def _victim_stats(raw_fn):
    out = (C.c_int64 * 3)()
    raw_fn(out)
    return tuple(int(x) for x in out)


@pytest.mark.parametrize("nodes", [10, 30, 60])
def test_hostsim_reclaim_large_jobs_walks_the_victims_log(nodes):
    """The reference's BenchmarkReclaimLargeJobs shape (integration_tests/reclaim/reclaim_benchmark_test.go:62-160, restated by tools/ref_benchmarks.py): one pending gang of
    nodes/10 x 8-GPU tasks against 8 running one-GPU jobs per node — per partial job hundreds of scenarios that end at the AccumulatedIdleGpus filter, recorded victims in front
    of them.  The engine pops the victims queue once per pending job (kai_engine_solver.inc vl_*), walks the log per partial job, keeps the filter as a running sum and builds the
    task groups only for the scenario that gets through: operations, scenarios simulated, simulations run and scenarios dropped must be the oracle's."""
    import sys
    sys.path.insert(0, os.path.join(T.ROOT, "tools"))
    import ref_benchmarks as RB
    snap, cfg, _ = T.case_to_snapshot(RB.reclaim_large(nodes), ("reclaim",))
    ref = T.Oracle.run(snap, cfg, ("reclaim",)); o_stats = _victim_stats(T.Oracle.lib().kai_oracle_last_victim_stats)
    res = HostSim.run(snap, cfg, ("reclaim",)); e_stats = _victim_stats(HostSim._raw.kai_hostsim_last_victim_stats)
    assert_same(res, ref)
    assert e_stats[:2] == o_stats[:2] and e_stats[2] >= o_stats[2]  # (the engine also books a simulation its pre-check turns away as a dropped scenario)
    assert o_stats[2] > 5 * nodes


@pytest.mark.parametrize("seed", range(40))
def test_hostsim_victims_log_with_recorded_victims_of_elastic_jobs(seed):
    """Crowded clusters of elastic running jobs (more pods than minAvailable) under reclaim and preempt: GetTasksToEvict hands an elastic job over one task at a time and pushes it
    back, so a partial job's recorded victims meet log entries whose job still has tasks in the queue — the walk has to tell an entry it may pass over from one that changes the
    pop sequence (vl_diverge: the queue rebuilt, the entries so far replayed as the real builder)."""
    snap = T.pkg.synth.make_crowded_snapshot(6 + seed % 5, 7000 + seed)
    cfg = T.abi.default_config(max_consolidation_preemptees=-1)
    acts = ("allocate", "reclaim", "preempt") if seed % 2 else ("allocate", "consolidation", "reclaim")
    assert_same(HostSim.run(snap, cfg, acts), T.Oracle.run(snap, cfg, acts))


def test_hostprep_refuses_malformed_snapshots_like_a_sequential_pass():
    """kai_session_open's host preparation (kai_host_prep.hpp) checks every index of the snapshot on the host's cores, chunk by chunk; the failure it reports must be the one
    a sequential pass stops at: the checks in their order (pod-sets, jobs, pods, …), and within one loop the lowest index.  A snapshot large enough for several chunks."""
    import ctypes as C
    HostSim.lib(); raw = HostSim._raw
    raw.kai_hostsim_prep_error.restype = C.c_int

    def verdict(snap, cfg):
        st = snap.as_struct(); buf = C.create_string_buffer(256)
        rc = raw.kai_hostsim_prep_error(C.byref(cfg), C.byref(st), buf, 256)
        return rc, buf.value.decode()

    def fresh():
        snap, cfg, _ = T.pkg.synth.config(1, 8.0)   # 8 000 nodes x ~88 000 pods
        for k in list(snap.arrays):                 # own copies: the cases below write into them
            snap.arrays[k] = snap.arrays[k].copy()
        return snap, cfg
    snap, cfg = fresh()
    P, J, S, Q = snap.n_pods, snap.n_jobs, snap.n_podsets, snap.n_queues
    assert P >= 65536 and verdict(snap, cfg) == (0, "")
    cases = [
        (lambda a: a["pod_job"].__setitem__(P - 3, J), "pod_job out of range"),
        (lambda a: (a["pod_job"].__setitem__(P - 3, J), a["podset_job"].__setitem__(S - 1, -1)), "podset_job out of range"),          # the pod-set loop runs first
        (lambda a: (a["pod_job"].__setitem__(5, -7), a["pod_podset"].__setitem__(P - 1, S + 9)), "pod_job out of range"),            # lowest pod wins inside the pod loop
        (lambda a: (a["pod_podset"].__setitem__(P - 1, S + 9), a["job_n_podsets"].__setitem__(J - 1, -1)), "job pod-set range out of bounds"),
        (lambda a: a["job_queue"].__setitem__(J - 2, Q), "bad job_queue"),
        (lambda a: a["job_queue"].__setitem__(J // 2, -5), "bad job_queue"),
        (lambda a: a["job_first_pod"].__setitem__(J - 1, P), "job pod range out of bounds"),
        (lambda a: a["job_first_pod"].__setitem__(J - 1, int(a["job_first_pod"][J - 2])), "job pod ranges overlap"),
        (lambda a: a["node_name_rank"].__setitem__(7, int(a["node_name_rank"][8])), "node_name_rank must be a permutation of 0..N-1"),
        (lambda a: a["queue_parent"].__setitem__(0, 0), "bad queue_parent"),
    ]
    for corrupt, want in cases:
        snap, cfg = fresh()
        if want == "job pod ranges overlap" and int(snap.arrays["job_n_pods"][J - 1]) == 0:
            snap.arrays["job_n_pods"][J - 1] = 1
        corrupt(snap.arrays)
        rc, msg = verdict(snap, cfg)
        assert rc != 0 and msg == want, (want, rc, msg)
