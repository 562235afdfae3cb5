"""Node-axis sharding over several ranks on CPU: world-size-2 and -3 gloo groups run the REAL exchange protocol of the batch
path — every rank owns a contiguous name-rank range of the nodes, offers its K best nodes per scan class, the offers are all-gathered
(torch.distributed, gloo here / RCCL on the GPU box) into a virtual cluster on which every rank runs the same fill until a class runs out of
offers that beat the held-back floors — with the kernels on the lock-step emulator (tests/host_sim).  The committed operations, pod states,
node accounting and queue shares of EVERY rank must equal the single-rank run and the oracle bit for bit."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _cases():
    sys.path.insert(0, HERE)
    import kai_testlib as T
    from test_batch_path import regular_snapshot
    out = []  # (snapshot, config, actions of the cycle, does the allocate action take the (sharded) batch path)
    for idx, scale in ((1, 0.1), (2, 0.02), (4, 0.004)):
        snap, cfg, _ = T.pkg.synth.config(idx, scale)
        out.append((snap, cfg, ("allocate",), True))
    for seed in (1, 2, 5, 9, 12):
        out.append((regular_snapshot(seed), T.abi.default_config(gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], k_value=0.5), ("allocate",), True))
    # actions the group does not shard run replicated on every rank: BASELINE config 4 (zone / rack topology gangs; allocate, consolidation, reclaim),
    # elastic gangs and two pod-sets, a crowded cluster under all four actions with an allocate after the victim actions (sharded fill, then replicated engine, ...)
    snap, cfg, _ = T.pkg.synth.config(3, 0.004)
    out.append((snap, cfg, ("allocate", "consolidation", "reclaim"), False))
    out.append((T.pkg.synth.make_snapshot(30, 300, 77, queue_levels=(2, 2), elastic_frac=0.4, multi_podset_frac=0.3), T.abi.default_config(), ("allocate",), False))
    out.append((T.pkg.synth.make_crowded_snapshot(8, 1003, elastic_frac=0.0), T.abi.default_config(max_consolidation_preemptees=-1), ("allocate", "reclaim", "preempt", "allocate"), None))
    return out


def _worker(rank, world, port, k_offers, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import kai_testlib as T
    from test_engine_hostsim import HostSim
    d = T.pkg.dist
    d.init("gloo")
    HostSim.lib()
    raw = HostSim._raw

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
    def allgather(user, send, recv, nbytes):  # the group's exchange step: the library's buffers, the caller's collective
        s = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,)))
        r = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,)))
        dist.all_gather_into_tensor(r, s)
        return 0

    rows = []
    for snap, cfg, actions, _ in _cases():
        raw.kai_hostsim_set_shard(rank, world, k_offers, allgather, None)
        res = HostSim.run(snap, cfg, actions)
        rows.append((res.ops, res.stmts, res.pod_status.tolist(), res.pod_node.tolist(), {k: v.tolist() for k, v in res.nodes.items()},
                     {k: v.tolist() for k, v in res.shares_final.items()}, int(res.stats.reserved[4]), int(raw.kai_hostsim_last_exchanges()),
                     (int(res.stats.decisions), int(res.stats.jobs_attempted), int(res.stats.jobs_committed), int(res.stats.rollbacks))))
    raw.kai_hostsim_set_shard(0, 1, 0, None, None)
    out.put((rank, rows))
    d.finish()


@pytest.mark.parametrize("world,k_offers", [(2, 0), (2, 16), (3, 16)])
def test_node_sharded_group_equals_one_rank(world, k_offers):
    sys.path.insert(0, HERE)
    import kai_testlib as T
    from test_engine_hostsim import HostSim
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k_offers, out)) for r in range(world)]
    for p in procs: p.start()
    got = dict(out.get(timeout=600) for _ in range(world))
    for p in procs: p.join(timeout=60); assert p.exitcode == 0
    for ci, (snap, cfg, actions, want_batch) in enumerate(_cases()):
        ref = T.Oracle.run(snap, cfg, actions)
        one = HostSim.run(snap, cfg, actions)
        assert one.ops == ref.ops
        for rank in range(world):
            ops, stmts, st, nd, nodes, shares, batch, exchanges, stats = got[rank][ci]
            if want_batch is True:
                assert batch == 1, "the sharded group did not take the batch path"
                assert exchanges >= 1 or len(ref.ops) == 0
            elif want_batch is False:
                assert batch == 0  # replicated: the same engine on every rank, no exchange
            assert [tuple(o) for o in ops] == ref.ops and stmts == ref.stmts
            assert st == ref.pod_status.tolist() and nd == ref.pod_node.tolist()
            for k in ref.nodes: assert nodes[k] == ref.nodes[k].tolist(), k
            for k in ref.shares_final: assert shares[k] == ref.shares_final[k].tolist(), k
            if len(actions) == 1:  # (statistics are those of the cycle's last action)
                assert stats == (ref.stats.decisions, ref.stats.jobs_attempted, ref.stats.jobs_committed, ref.stats.rollbacks)


def test_single_process_is_a_noop():
    sys.path.insert(0, HERE)
    import kai_testlib as T
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"): os.environ.pop(k, None)
    d = T.pkg.dist
    assert d.env_world() == (0, 0, 1)
    d.barrier(); assert d.max_over_ranks(3.5) == 3.5 and d.sum_over_ranks(2.0) == 2.0


# ------------------------------------------------------------------------------------------------ the victim search's waves over the ranks of a group
def _victim_cases():
    sys.path.insert(0, HERE)
    import kai_testlib as T
    out = []  # (snapshot, config, actions)
    snap, cfg, _ = T.pkg.synth.config(3, 0.01)  # BASELINE config 4: zone / rack topology gangs, consolidation + reclaim
    out.append((snap, cfg, ("allocate", "consolidation", "reclaim")))
    for seed in range(6):  # crowded clusters: long victims queues, rejections by the reclaim validator that leave nodes feasible, elastic gangs
        s = T.pkg.synth.make_crowded_snapshot(6 + 3 * seed, 7700 + seed, fill=0.7 + 0.25 * (seed % 4) / 3, n_pending_jobs=6 + 2 * seed, elastic_frac=0.25 * (seed % 3), hog_frac=0.5,
                                              queue_levels=((2, 2), (3,), (2, 2, 2))[seed % 3])
        c = T.abi.default_config(max_consolidation_preemptees=(-1, 16, 2)[seed % 3], gpu_strategy=(T.abi.BINPACK, T.abi.SPREAD)[seed % 2], k_value=(0.0, 0.5, 1.0)[seed % 3])
        c.use_scheduling_signatures = seed % 2
        out.append((s, c, (("allocate", "reclaim", "preempt", "allocate"), ("reclaim",), ("consolidation", "preempt"), ("allocate", "consolidation", "reclaim", "preempt"))[seed % 4]))
    return out


def _victim_worker(rank, world, port, engines, cap, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import kai_testlib as T
    from test_engine_hostsim import HostSim
    d = T.pkg.dist
    d.init("gloo")
    HostSim.lib()
    raw = HostSim._raw
    raw.kai_hostsim_victim_exchanges.restype = C.c_int64

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
    def allgather(user, send, recv, nbytes):
        s = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,)))
        r = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,)))
        dist.all_gather_into_tensor(r, s)
        return 0

    rows = []
    raw.kai_hostsim_set_multi(engines)
    for snap, cfg, actions in _victim_cases():
        raw.kai_hostsim_set_shard(rank, world, 16, allgather, None)
        raw.kai_hostsim_set_victim_shard(1, cap, allgather, None)
        res = HostSim.run(snap, cfg, actions)
        st = (C.c_int64 * 4)(); raw.kai_hostsim_multi_stats(st)
        rows.append((res.ops, res.stmts, res.pod_status.tolist(), res.pod_node.tolist(), {k: v.tolist() for k, v in res.nodes.items()},
                     {k: v.tolist() for k, v in res.shares_final.items()}, int(raw.kai_hostsim_victim_exchanges()), list(st),
                     (int(res.stats.decisions), int(res.stats.jobs_attempted), int(res.stats.jobs_committed), int(res.stats.reserved[2]), int(res.stats.reserved[3]))))
    raw.kai_hostsim_set_victim_shard(0, 0, None, None)
    raw.kai_hostsim_set_shard(0, 1, 0, None, None)
    raw.kai_hostsim_set_multi(1)
    out.put((rank, rows))
    d.finish()


@pytest.mark.parametrize("world,engines,cap", [(2, 1, 0), (2, 3, 0), (3, 2, 5), (2, 4, 1024)])
def test_victim_waves_over_the_ranks_of_a_group(world, engines, cap):
    """SURVEY 8e, the victim actions: every rank holds the whole session, simulation i of a wave belongs to rank i mod world, the ranks all-gather the wave's outcomes
    (gloo here; RCCL or the caller's collective on the GPU box) and count the wave the same way.  Every rank must end with the oracle's operations, Statement numbers,
    pod states, node accounting, queue shares AND the one-rank statistics (scenarios / simulations count what the reference's order reaches, not what was run)."""
    sys.path.insert(0, HERE)
    import kai_testlib as T
    from test_engine_hostsim import HostSim
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_victim_worker, args=(r, world, port, engines, cap, out)) for r in range(world)]
    for p in procs: p.start()
    got = dict(out.get(timeout=300) for _ in range(world))
    for p in procs: p.join(timeout=60); assert p.exitcode == 0
    HostSim.lib(); HostSim._raw.kai_hostsim_set_multi(1)
    total_exchanges = 0
    for ci, (snap, cfg, actions) in enumerate(_victim_cases()):
        ref = T.Oracle.run(snap, cfg, actions)
        one = HostSim.run(snap, cfg, actions)
        assert one.ops == ref.ops
        for rank in range(world):
            ops, stmts, st, nd, nodes, shares, exchanges, mw, stats = got[rank][ci]
            assert [tuple(o) for o in ops] == ref.ops and stmts == ref.stmts
            assert st == ref.pod_status.tolist() and nd == ref.pod_node.tolist()
            for k in ref.nodes: assert nodes[k] == ref.nodes[k].tolist(), k
            for k in ref.shares_final: assert shares[k] == ref.shares_final[k].tolist(), k
            assert stats == (int(one.stats.decisions), int(one.stats.jobs_attempted), int(one.stats.jobs_committed), int(one.stats.reserved[2]), int(one.stats.reserved[3]))  # (decisions, jobs, scenarios, simulations)
            n_victim_actions = sum(1 for a in actions if a != "allocate")
            assert exchanges >= n_victim_actions  # at least the closing message of every victim action
            assert exchanges == got[0][ci][6]     # the same number of collectives on every rank
            total_exchanges += exchanges
    assert total_exchanges > world * sum(sum(1 for a in acts if a != "allocate") for _, _, acts in _victim_cases())  # waves were exchanged, not only closing messages


@pytest.mark.parametrize("ranks", [1, 2, 3, 4, 8, 16])
def test_wave_exchange_protocol(ranks):
    """kai_victim_shard.hpp on its own (pack / merge / closing message), R ranks in one process: 50 random waves per seed — every rank ends with the identical merged wave,
    equal to what one rank that ran every simulation would hold (hit = the lowest simulation that did not simply fail, every counted simulation's status and counters) —
    and the fault path: a rank that leaves the protocol during the others' wave takes them out (fault flag in the merged wave, no further collective, the same number
    of collectives on every rank)."""
    sys.path.insert(0, HERE)
    from test_engine_hostsim import HostSim
    HostSim.lib()
    fn = HostSim._raw.kai_hostsim_xw_selftest
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
    for cap in (0, 1, 5, 64, 128, 1024, 5000):
        for seed in range(1, 9):
            assert fn(ranks, cap, 0, 0, seed) == 0, (ranks, cap, seed)
        for gone in range(min(ranks, 4)):
            assert fn(ranks, cap, 1, gone, 7) == 0, (ranks, cap, gone)
