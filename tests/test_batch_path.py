"""The batch path of the allocate action (kai-scheduler_amd/csrc/kai_batch.hpp: plan / fill / apply) against the oracle.

CPU part: the kernels' bodies run on the lock-step SIMT emulator of kai_simt.hpp inside tests/host_sim (debug aid, see its header);
`stats.reserved[4]` = allocate actions that took the batch path, `reserved[5]` = plan/fill rounds.  The `-m gpu` twin of these tests is in
tests/test_gpu_parity.py (same snapshots through the C ABI on the MI355X)."""
import numpy as np
import pytest

import kai_testlib as T
from test_engine_hostsim import HostSim, assert_same

abi = T.abi
synth = T.pkg.synth


def stats_tuple(s):
    return (s.decisions, s.jobs_attempted, s.jobs_committed, s.rollbacks)


def regular_snapshot(seed):
    """Clusters whose queued jobs are all 'regular' (plain gangs, one pod-set): what the batch path takes.  Everything else varies:
    queue tree shape (1-4 levels, ragged fan-out), quotas (zipf), limits, queue priorities, over-quota weights, historical usage,
    non-preemptible jobs, CPU-only jobs and nodes, node mix, pre-filled nodes, lexicographic node names."""
    rng = np.random.default_rng(9000 + seed)
    levels = [(1,), (3,), (2, 2), (3, 4), (2, 2, 2), (1, 5), (4, 1, 3), (2, 3, 2, 2)][seed % 8]
    n_nodes = int(rng.integers(1, 150)); n_pods = int(rng.integers(1, 1200))
    snap = synth.make_snapshot(n_nodes, n_pods, 9000 + seed, queue_levels=levels, prefill=float(rng.random()) * 0.9,
                               gpu_mix=((8, .5), (4, .3), (0, .2)) if seed % 3 else ((8, 1.0),), cpu_only_frac=0.3 if seed % 2 else 0.0,
                               zipf=bool(seed % 2), limits_frac=0.4 if seed % 3 == 0 else 0.0, queue_prios=(100, 200) if seed % 4 < 2 else (100,),
                               oqws=(1.0, 2.0, 4.0), nonpreempt_frac=0.25 if seed % 5 < 3 else 0.0, usage_max=0.3 if seed % 2 else 0.0,
                               lexi_names=bool(seed % 7 == 0), single_pod_jobs=bool(seed % 11 == 0))
    return snap


def run_both(snap, cfg, need_batch=True):
    ref = T.Oracle.run(snap, cfg)
    res = HostSim.run(snap, cfg)
    assert_same(res, ref)
    assert stats_tuple(res.stats) == stats_tuple(ref.stats)
    if need_batch:
        assert res.stats.reserved[4] == 1, "the allocate action did not take the batch path"
    seq = abi.KaiConfig.from_buffer_copy(cfg); seq.engine_mode = 3  # the sequential engine on the same snapshot
    res3 = HostSim.run(snap, seq)
    assert res3.stats.reserved[4] == 0
    assert_same(res3, ref)
    return res


@pytest.mark.parametrize("idx,scale", [(0, 1.0), (1, 0.2), (2, 0.03), (4, 0.006)])
def test_batch_baseline_configs(idx, scale):
    snap, cfg, _ = synth.config(idx, scale)
    res = run_both(snap, cfg)
    assert res.stats.reserved[5] >= 1


@pytest.mark.parametrize("seed", range(48))
def test_batch_random_regular(seed):
    snap = regular_snapshot(seed)
    strat = (abi.BINPACK, abi.SPREAD)[seed % 2]
    cfg = abi.default_config(gpu_strategy=strat, cpu_strategy=(abi.BINPACK, abi.SPREAD)[(seed // 2) % 2], k_value=float(seed % 3) * 0.5)
    run_both(snap, cfg)


def test_batch_declines_irregular_jobs():
    """elastic gangs / two pod-sets are pushed back or walk the sub-group tree: the sequential engine takes the action"""
    snap = synth.make_snapshot(30, 300, 77, queue_levels=(2, 2), elastic_frac=0.4, multi_podset_frac=0.3)
    res = HostSim.run(snap, abi.default_config())
    assert res.stats.reserved[4] == 0
    assert_same(res, T.Oracle.run(snap, abi.default_config()))


def test_batch_after_victim_actions_is_sequential():
    """once something is releasing in the session the batch path (Idle == Idle+Releasing) is off"""
    snap = synth.make_crowded_snapshot(8, 1003, elastic_frac=0.0)
    cfg = abi.default_config(max_consolidation_preemptees=-1)
    acts = ("reclaim", "allocate")
    ref = T.Oracle.run(snap, cfg, acts); res = HostSim.run(snap, cfg, acts)
    assert_same(res, ref)


def test_batch_block_level_in_hbm(monkeypatch):
    """the fill kernel's variant for clusters whose block level does not fit the LDS (KAI_BATCH_L1_HBM forces it)"""
    monkeypatch.setenv("KAI_BATCH_L1_HBM", "1")
    for seed in (3, 8, 13):
        snap = regular_snapshot(seed)
        run_both(snap, abi.default_config(k_value=0.5))


@pytest.mark.parametrize("seed", range(12))
def test_batch_plugin_subsets(seed):
    """without some of the node-order / predicate plugins the fill kernel runs its general (not constant-folded) class key"""
    snap = regular_snapshot(100 + seed)
    cfg = abi.default_config(gpu_strategy=(abi.BINPACK, abi.SPREAD)[seed % 2], k_value=0.5)
    drop = (abi.PLUGIN_RESOURCETYPE, abi.PLUGIN_NODEAVAILABILITY, abi.PLUGIN_NODEPLACEMENT, abi.PLUGIN_RESOURCETYPE | abi.PLUGIN_NODEAVAILABILITY)[seed % 4]
    cfg.plugins &= ~drop
    run_both(snap, cfg)


@pytest.mark.parametrize("order", [1, 2])
def test_batch_under_other_wave_schedules(order):
    """The emulator lets the waves of a workgroup take turns in reverse or drift apart at random (kai_simt.hpp KW_EMU_ORDER): results that depend
    on it are races between waves (plan, apply and exchange kernels run several waves per workgroup).  The setting is read once per process."""
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import kai_testlib as T\nfrom test_engine_hostsim import HostSim\nfrom test_batch_path import regular_snapshot\n"
            "for seed in (0, 3, 7, 11, 29, 41):\n"
            "    snap = regular_snapshot(seed); cfg = T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed %% 2], k_value=0.5)\n"
            "    ref = T.Oracle.run(snap, cfg); res = HostSim.run(snap, cfg)\n"
            "    assert res.ops == ref.ops and res.stats.reserved[4] == 1, seed\n"
            "snap, cfg, _ = T.pkg.synth.config(2, 0.03)\n"
            "assert HostSim.run(snap, cfg).ops == T.Oracle.run(snap, cfg).ops\n") % os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, KW_EMU_ORDER=str(order), KW_EMU_SEED="5")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


# ---------------------------------------------------------------------------------------------- the bucket fill (kai_fill_buckets.hpp)
# stats.reserved[6] of the host simulation = allocate actions whose fill ran on the bucket kernel (sets of nodes by free devices in LDS)
def _buckets(res):
    return int(res.stats.reserved[6])


@pytest.mark.parametrize("seed", range(16))
def test_bucket_fill_takes_binpacked_gpu_classes(seed, monkeypatch):
    """bin-packed GPU classes on nodes where only the devices can bind: the bucket kernel and the general kernel against the oracle, and against each other"""
    rng = np.random.default_rng(5100 + seed)
    snap = synth.make_snapshot(int(rng.integers(1, 400)), int(rng.integers(1, 1500)), 5100 + seed, queue_levels=[(1,), (2, 2), (3, 4), (2, 2, 2)][seed % 4],
                               prefill=float(rng.random()) * 0.9, gpu_mix=((8, .6), (4, .4)) if seed % 2 else ((8, 1.0),), zipf=bool(seed % 2),
                               limits_frac=0.3 if seed % 3 == 0 else 0.0, lexi_names=bool(seed % 5 == 0), gpus_per_pod=(1, 2, 4, 8) if seed % 4 else (1, 3, 5))
    cfg = abi.default_config(k_value=0.5)
    res = run_both(snap, cfg)
    assert _buckets(res) == 1
    monkeypatch.setenv("KAI_FILL_GENERAL", "1")
    gen = HostSim.run(snap, cfg)
    assert _buckets(gen) == 0 and gen.stats.reserved[4] == 1
    assert_same(gen, res)
    assert stats_tuple(gen.stats) == stats_tuple(res.stats)


@pytest.mark.parametrize("seed", range(12))
def test_bucket_fill_whole_nodes_per_step(seed, monkeypatch):
    """A gang of one class is placed in steps of whole nodes (the class's best node takes g / q tasks, the next nodes of its level follow in one ds_xor): large gangs of
    small requests on 16-device nodes, 3- and 5-device requests (a node is left at g mod q and becomes another class's best), nearly full and nearly empty clusters —
    against the oracle, against one placement per step (KAI_FILL_UNBATCHED) and against the general kernel"""
    rng = np.random.default_rng(5600 + seed)
    sizes, probs = ((1, 4, 8, 16, 64, 100), (.1, .2, .3, .2, .1, .1)) if seed % 2 else ((1, 2, 3, 24), (.3, .2, .2, .3))
    snap = synth.make_snapshot(int(rng.integers(3, 300)), int(rng.integers(50, 2500)), 5600 + seed, queue_levels=[(1,), (2, 2), (3, 4)][seed % 3], prefill=(0.0, 0.3, 0.6, 0.9)[seed % 4],
                               gpu_mix=((16, .5), (8, .5)) if seed % 3 == 0 else ((8, .7), (4, .3)), gpus_per_pod=(1, 2, 4, 8) if seed % 4 else (1, 3, 5), gang_sizes=sizes, gang_p=probs,
                               mem_per_gpu=8 * synth.GIB, cpu_per_gpu=2000.0, lexi_names=bool(seed % 5 == 0))
    cfg = abi.default_config(k_value=0.5)
    res = run_both(snap, cfg)
    if _buckets(res) != 1:
        pytest.skip("this cluster does not qualify for the bucket fill (a resource other than the devices may bind first)")
    monkeypatch.setenv("KAI_FILL_UNBATCHED", "1")
    one = HostSim.run(snap, cfg)
    assert _buckets(one) == 1
    assert_same(one, res); assert stats_tuple(one.stats) == stats_tuple(res.stats)
    monkeypatch.delenv("KAI_FILL_UNBATCHED"); monkeypatch.setenv("KAI_FILL_GENERAL", "1")
    gen = HostSim.run(snap, cfg)
    assert _buckets(gen) == 0
    assert_same(gen, res); assert stats_tuple(gen.stats) == stats_tuple(res.stats)


def _counts(res):
    """allocate actions whose fill ran as two wavefronts (kai_fill_counts.hpp: the planned order over the levels' populations, the sets behind a command ring)"""
    return int(res.stats.reserved[7]) & 0xffffffff


def _levels(res):
    """... with a wavefront per level behind the counting machine (kai_fill_levels.hpp)"""
    return int(res.stats.reserved[7]) >> 32


@pytest.mark.parametrize("seed", range(20))
def test_counts_fill_against_the_one_wave_kernel_the_general_kernel_and_the_oracle(seed, monkeypatch):
    """The fill split in two (kai_fill_counts.hpp) takes the plain clusters — no class with a static bitmap of its own: gangs of one class decided from the levels' populations,
    gangs of several classes simulated on a copy of the counts, the tasks' nodes resolved by the set worker behind the ring.  Against the oracle, the one-wave bucket kernel
    (KAI_FILL_ONE_WAVE) and the general kernel; the native scalar shadow of tests/host_sim checks every launch's outputs on the way (KAI_HOSTSIM_NATIVE_FILL)."""
    rng = np.random.default_rng(5900 + seed)
    sizes, probs = [((1, 4, 8, 16, 64, 100), (.1, .2, .3, .2, .1, .1)), ((1, 2, 3, 24), (.3, .2, .2, .3)), ((1, 2, 700), (.6, .3, .1)), ((1,), (1.0,))][seed % 4]
    snap = synth.make_snapshot(int(rng.integers(1, 500)), int(rng.integers(1, 3000)), 5900 + seed, queue_levels=[(1,), (2, 2), (3, 4), (2, 2, 2)][seed % 4], prefill=(0.0, 0.3, 0.6, 0.95)[seed % 4],
                               gpu_mix=((16, .5), (8, .5)) if seed % 3 == 0 else ((8, .7), (4, .3)), gpus_per_pod=(1, 2, 4, 8) if seed % 5 else (1, 3, 5), gang_sizes=sizes, gang_p=probs,
                               mem_per_gpu=8 * synth.GIB, cpu_per_gpu=2000.0, zipf=bool(seed % 2), limits_frac=0.3 if seed % 3 == 1 else 0.0, lexi_names=bool(seed % 7 == 0))
    cfg = abi.default_config(k_value=0.5)
    monkeypatch.setenv("KAI_HOSTSIM_NATIVE_FILL", "1")
    res = run_both(snap, cfg)
    if _buckets(res) != 1:
        pytest.skip("this cluster does not qualify for the sets by free devices")
    assert _counts(res) == 1
    import ctypes as C
    ms, a, b, d = C.c_double(), C.c_int64(), C.c_int64(), C.c_int64()
    HostSim._raw.kai_hostsim_native_fill(C.byref(ms), C.byref(a), C.byref(b), C.byref(d))
    assert a.value >= 1 and d.value == 0, "the native shadow of a launch ended with other outputs than the emulated kernel"
    assert _levels(res) == (0 if seed % 3 == 0 else 1)  # (16-device nodes: more levels than kai_fill_levels.hpp has wavefronts for)
    if _levels(res):  # the kernel of kai_fill_counts.hpp (two set workers) on the same snapshot
        monkeypatch.setenv("KAI_FILL_TWO_WORKERS", "1")
        two = HostSim.run(snap, cfg)
        HostSim._raw.kai_hostsim_native_fill(C.byref(ms), C.byref(a), C.byref(b), C.byref(d))
        assert _counts(two) == 1 and _levels(two) == 0 and a.value >= 1 and d.value == 0
        assert_same(two, res); assert stats_tuple(two.stats) == stats_tuple(res.stats)
        monkeypatch.delenv("KAI_FILL_TWO_WORKERS")
    monkeypatch.setenv("KAI_FILL_ONE_WAVE", "1")
    one = HostSim.run(snap, cfg)
    assert _buckets(one) == 1 and _counts(one) == 0
    assert_same(one, res); assert stats_tuple(one.stats) == stats_tuple(res.stats)
    monkeypatch.delenv("KAI_FILL_ONE_WAVE"); monkeypatch.setenv("KAI_FILL_GENERAL", "1")
    gen = HostSim.run(snap, cfg)
    assert _buckets(gen) == 0
    assert_same(gen, res); assert stats_tuple(gen.stats) == stats_tuple(res.stats)


def test_fill_levels_small_arithmetic():
    """kai_fill_levels.hpp: the scalar division by multiplication (every a <= 1024, b <= 8), the packed quotient tables, the (source, target) ring numbering, the 8-byte command's fields"""
    HostSim.lib()
    assert HostSim._raw.kai_hostsim_fill_levels_selfcheck() == 0


@pytest.mark.parametrize("two_workers", [0, 1])
@pytest.mark.parametrize("order", [1, 2])
def test_counts_fill_does_not_depend_on_how_the_two_wavefronts_interleave(order, two_workers):
    """The counting machine and the set workers only meet at the command ring (and the workers of kai_fill_levels.hpp at their hand-over rings): the emulator runs the waves of
    the workgroup in reverse order and with random passes sat out (KW_EMU_ORDER), the results stay the oracle's."""
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import kai_testlib as T\nfrom test_engine_hostsim import HostSim\nfrom test_batch_path import assert_same\n"
            "for seed in (1, 2, 3):\n"
            "    snap = T.pkg.synth.make_snapshot(150, 1500, 6100 + seed, queue_levels=(2, 3), prefill=0.4, gang_sizes=(1, 4, 40), gang_p=(.5, .3, .2), mem_per_gpu=8 * T.pkg.synth.GIB, cpu_per_gpu=2000.0)\n"
            "    cfg = T.abi.default_config(k_value=0.5)\n"
            "    res = HostSim.run(snap, cfg)\n"
            "    assert res.stats.reserved[7] == (1 if %d else 1 | 1 << 32), seed\n"
            "    assert_same(res, T.Oracle.run(snap, cfg))\n") % (T.ROOT, os.path.join(T.ROOT, "tests"), two_workers)
    env = dict(os.environ, KW_EMU_ORDER=str(order), KW_EMU_SEED="11")
    if two_workers: env["KAI_FILL_TWO_WORKERS"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]


def test_bucket_fill_declines_when_another_resource_may_bind():
    """20 000 mCPU per device on nodes of 64 000 - 192 000 mCPU: the CPU runs out before the devices do on most nodes, k_bucket_build's proof fails and
    the general kernel takes the fill — same results"""
    for seed in (1, 2, 3):
        snap = synth.make_snapshot(60, 500, 5200 + seed, prefill=0.2, cpu_per_gpu=20000.0)
        res = run_both(snap, abi.default_config(k_value=0.5))
        assert _buckets(res) == 0
    # memory: 96 GiB per device on nodes of 256 - 1024 GiB
    snap = synth.make_snapshot(60, 500, 5210, prefill=0.2, mem_per_gpu=96 * synth.GIB)
    assert _buckets(run_both(snap, abi.default_config())) == 0
    # pod slots: 3 slots on a node of 8 devices
    snap = synth.make_snapshot(40, 400, 5211, prefill=0.0)
    snap.arrays["node_allocatable"][abi.RES_PODS, ::3] = 3; snap.finalize()
    assert _buckets(run_both(snap, abi.default_config())) == 0


@pytest.mark.parametrize("seed", range(6))
def test_bucket_fill_with_static_predicates(seed):
    """pod classes x node classes (the compiled NodeAffinity / TaintToleration table), nodes that are not ready, worker labels under restrictSchedulingNodes: every
    class looks its nodes up through its own bitmap"""
    rng = np.random.default_rng(5300 + seed)
    snap = synth.make_snapshot(int(rng.integers(20, 300)), int(rng.integers(100, 1200)), 5300 + seed, queue_levels=(2, 3), prefill=0.3)
    synth.add_predicate_features(snap, 5300 + seed, nominated_frac=0.0, oversized_frac=0.02 if seed % 2 else 0.0)
    a = snap.arrays
    a["node_class"] = rng.integers(0, 3, size=snap.n_nodes).astype(np.int32)
    a["pod_class"] = np.repeat(rng.integers(0, 2, size=snap.n_jobs), a["job_n_pods"]).astype(np.int32)
    a["class_fit"] = np.array([[1, 1, 0], [1, 0, 1]], np.uint8)
    snap.finalize()
    cfg = abi.default_config(k_value=0.5)
    cfg.restrict_node_scheduling = seed % 2
    res = run_both(snap, cfg)
    assert _buckets(res) == (0 if seed % 2 else 1)  # (an oversized request asks for more CPU than any node has: the proof cannot hold for its class)


def test_bucket_fill_levels():
    """16 devices per node are the most the sets hold; 32 go to the general kernel; requests of 16 devices; one node; no pending pod"""
    snap = synth.make_snapshot(50, 600, 5400, gpu_mix=((16, .5), (8, .5)), gpus_per_pod=(1, 2, 4, 8, 16), mem_per_gpu=8 * synth.GIB, cpu_per_gpu=2000.0, prefill=0.4)
    assert _buckets(run_both(snap, abi.default_config())) == 1
    snap = synth.make_snapshot(50, 600, 5401, gpu_mix=((32, .5), (8, .5)), gpus_per_pod=(1, 2, 4, 8), mem_per_gpu=4 * synth.GIB, cpu_per_gpu=1000.0, prefill=0.4)
    assert _buckets(run_both(snap, abi.default_config())) == 0
    snap = synth.make_snapshot(1, 40, 5402, prefill=0.0)
    assert _buckets(run_both(snap, abi.default_config())) == 1
    snap = synth.make_snapshot(130, 900, 5403, prefill=0.97)  # nearly full: most classes are dead from the start, gangs roll back
    assert _buckets(run_both(snap, abi.default_config())) == 1


@pytest.mark.parametrize("seed", range(16))
def test_round_loop_on_the_device_against_the_loop_on_the_host(seed, monkeypatch):
    """Rounds without the host (RoundCtl, k_round_next): the loop's state on the device, rounds enqueued ahead and the one too many at the end — against the loop that reads the
    fill's status on the host after every round (KAI_BATCH_HOST_LOOP=1): same operations, same counters, same number of rounds; both against the oracle."""
    snap = regular_snapshot(100 + seed) if seed % 2 else synth.make_snapshot(40 + 30 * seed, 400 + 150 * seed, 7700 + seed, queue_levels=[(2, 2), (3, 4), (1,), (2, 2, 2)][seed % 4],
                                                                               prefill=0.1 * (seed % 7), gpu_mix=((8, .6), (4, .4)), limits_frac=0.3 if seed % 3 == 0 else 0.0)
    cfg = abi.default_config(k_value=0.5, gpu_strategy=(abi.BINPACK, abi.SPREAD)[(seed // 2) % 2])
    if seed % 4 == 3:
        monkeypatch.setenv("KAI_BATCH_H0", "8")  # short first plans: many rounds, the plan's depth moves both ways
    dev = run_both(snap, cfg)
    monkeypatch.setenv("KAI_BATCH_HOST_LOOP", "1")
    host = HostSim.run(snap, cfg)
    assert host.stats.reserved[4] == 1 and dev.stats.reserved[4] == 1
    assert_same(host, dev)
    assert stats_tuple(host.stats) == stats_tuple(dev.stats)
    assert host.stats.reserved[5] == dev.stats.reserved[5], "the two loops took different numbers of rounds"


def inner_limits_snapshot(seed):
    """queue trees whose INNER queues carry GPU limits: a parent turns jobs away that its leaf would still take — the gate of an inner node ends its stream inside the plan
    (k_plan_scan pass 1 / k_seg_gate), which the leaf-only limits of the other generators never reach"""
    rng = np.random.default_rng(6600 + seed)
    snap = synth.make_snapshot(int(rng.integers(20, 250)), int(rng.integers(100, 1500)), 6600 + seed, queue_levels=[(2, 3), (3, 4), (2, 2, 2), (1, 5), (4,)][seed % 5], prefill=float(rng.random()) * 0.6,
                               gpu_mix=[((8, 1.0),), ((8, .6), (4, .4))][seed % 2], zipf=bool(seed % 2), limits_frac=0.3 if seed % 3 == 0 else 0.0, inner_limits_frac=(0.5, 1.0)[seed % 2],
                               gang_sizes=(1, 2, 4), gang_p=(.6, .3, .1), gpus_per_pod=(1, 2) if seed % 4 else (1, 2, 4, 8), nonpreempt_frac=0.2 * (seed % 2), queue_prios=(100, 200) if seed % 2 else (100,))
    cfg = abi.default_config(k_value=(0.0, 0.5, 1.0)[seed % 3], gpu_strategy=abi.BINPACK if seed % 4 else abi.SPREAD)
    return snap, cfg


@pytest.mark.parametrize("seed", range(10))
def test_plan_scan_in_segments_against_one_workgroup_per_node_and_the_oracle(seed, monkeypatch):
    """kai_plan_segments.hpp: a node's stream cut into segments on different workgroups (sums of the segments before, the first job turned away as a minimum over the segments,
    the running maximum of the segments before) against k_plan_scan (one workgroup per node) and the oracle — on trees whose inner queues turn jobs away (odd seeds: every inner
    queue limited) and on the plain generator (leaf limits only).  The emulator's segments hold 128 positions, so streams of a few hundred jobs take several."""
    snap, cfg = inner_limits_snapshot(seed) if seed < 6 else (regular_snapshot(200 + seed), abi.default_config(k_value=0.5))
    monkeypatch.setenv("KAI_PLAN_SEG_MIN", "1")           # every height below the root in segments
    seg = run_both(snap, cfg)
    monkeypatch.setenv("KAI_PLAN_SEG_MIN", "1000000000")  # never
    one = HostSim.run(snap, cfg)
    assert one.stats.reserved[4] == 1
    assert_same(one, seg)
    assert stats_tuple(one.stats) == stats_tuple(seg.stats)
    assert one.stats.reserved[5] == seg.stats.reserved[5], "the two forms of the scan planned different rounds"
