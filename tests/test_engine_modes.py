"""The engine's three node-search modes must agree bit for bit, including the decision counters:
0 = class index + staged job path, 1 = brute-force scans only, 2 = class index with the general job path.
(host simulation of the control flow here; tests/test_gpu_parity.py repeats it on the MI355X)"""
import numpy as np
import pytest

import kai_testlib as T
from test_engine_hostsim import HostSim, assert_same


def stats_tuple(s):
    return (s.decisions, s.jobs_attempted, s.jobs_committed, s.rollbacks)


def run_modes(run, snap, cfg):
    ref = T.Oracle.run(snap, cfg)
    for mode in (0, 1, 2):
        c = T.abi.KaiConfig.from_buffer_copy(cfg); c.engine_mode = mode
        res = run(snap, c)
        assert_same(res, ref)
        assert stats_tuple(res.stats) == stats_tuple(ref.stats), (mode, stats_tuple(res.stats), stats_tuple(ref.stats))


@pytest.mark.parametrize("idx,scale", [(0, 1.0), (1, 0.3), (2, 0.03), (4, 0.01)])
def test_modes_agree_synthetic(idx, scale):
    snap, cfg, _ = T.pkg.synth.config(idx, scale)
    run_modes(HostSim.run, snap, cfg)


@pytest.mark.parametrize("seed", range(10))
def test_modes_agree_random(seed):
    rng = np.random.default_rng(100 + seed)
    snap = T.pkg.synth.make_snapshot(int(rng.integers(1, 200)), int(rng.integers(0, 1500)), 2000 + seed, queue_levels=(2, 3), prefill=float(rng.random()) * 0.8,
                                     gpu_mix=((8, .5), (4, .3), (0, .2)), cpu_only_frac=0.3, zipf=True, limits_frac=0.3, queue_prios=(100, 200),
                                     oqws=(1.0, 2.0), nonpreempt_frac=0.2, usage_max=0.2, lexi_names=bool(seed % 2))
    for strat in (T.abi.BINPACK, T.abi.SPREAD):
        run_modes(HostSim.run, snap, T.abi.default_config(gpu_strategy=strat, cpu_strategy=strat, k_value=float(seed % 3) * 0.5))


def test_finite_queue_depth():
    snap, cfg, _ = T.pkg.synth.config(1, 0.2)
    for depth in (1, 3, 50):
        c = T.abi.KaiConfig.from_buffer_copy(cfg); c.queue_depth[0] = depth
        run_modes(HostSim.run, snap, c)


@pytest.mark.parametrize("seed", range(8))
def test_modes_agree_elastic_and_subgroups(seed):
    """elastic jobs (grow one pod per pop, re-pushed with a changed order key), two-pod-set gangs and task-order labels"""
    rng = np.random.default_rng(300 + seed)
    snap = T.pkg.synth.make_snapshot(int(rng.integers(4, 120)), int(rng.integers(20, 900)), 3000 + seed, queue_levels=(2, 2), prefill=float(rng.random()) * 0.6,
                                     gpu_mix=((8, .6), (4, .4)), zipf=bool(seed % 2), limits_frac=0.2, queue_prios=(100, 200), oqws=(1.0, 2.0),
                                     nonpreempt_frac=0.1, elastic_frac=0.4, multi_podset_frac=0.3, task_prio_frac=0.3, lexi_names=bool(seed % 3 == 0))
    run_modes(HostSim.run, snap, T.abi.default_config(k_value=float(seed % 2)))
