"""Randomized campaign of the victim actions' simulation waves over the ranks of a group (kai_victim_shard.hpp) on the CPU: `world` gloo processes, the engines on the
emulator's host twin (tests/host_sim), seeds lo..hi of tests/kai_testlib.py::broad_case (crowded clusters, topology, sub-groups, elastic gangs, minruntime, every action
order).  Every rank's operations, Statement numbers, pod states, node accounting and queue shares must equal the oracle's; every rank must issue the same number of collectives.
usage: dist_victim_campaign.py <seed lo> <seed hi> [world=2] [engines per rank=2] [wave length=0 (default)]      CAMPAIGN_SECONDS bounds the run."""
import ctypes as C, os, socket, sys, time
HERE = os.path.dirname(os.path.abspath(__file__)); TESTS = os.path.join(HERE, "..", "tests")
sys.path.insert(0, TESTS)
import numpy as np
import torch.multiprocessing as mp


def worker(rank, world, port, lo, hi, engines, cap, budget, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch, torch.distributed as dist
    import kai_testlib as T
    from test_engine_hostsim import HostSim
    T.pkg.dist.init("gloo"); HostSim.lib(); raw = HostSim._raw; raw.kai_hostsim_victim_exchanges.restype = C.c_int64

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
    def allgather(user, send, recv, nbytes):
        s = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,)))
        r = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,)))
        dist.all_gather_into_tensor(r, s); return 0

    raw.kai_hostsim_set_multi(engines)
    bad = tot = ex = 0; t0 = time.time()
    for seed in range(lo, hi):
        for ci, (snap, cfg, acts) in enumerate(T.broad_case(seed)):
            if all(a == "allocate" for a in acts): continue
            o = T.Oracle.run(snap, cfg, acts)
            raw.kai_hostsim_set_shard(rank, world, 16, allgather, None); raw.kai_hostsim_set_victim_shard(1, cap, allgather, None)
            g = HostSim.run(snap, cfg, acts); tot += 1
            n_ex = int(raw.kai_hostsim_victim_exchanges()); ex += n_ex
            t = torch.tensor([n_ex], dtype=torch.int64); lst = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]; dist.all_gather(lst, t)
            ok = o.ops == g.ops and g.stmts == o.stmts and (o.pod_status == g.pod_status).all() and (o.pod_node == g.pod_node).all() and all(np.array_equal(o.nodes[k], g.nodes[k]) for k in o.nodes) \
                and all(np.array_equal(o.shares_final[k], g.shares_final[k]) for k in o.shares_final) and len({int(x) for x in lst}) == 1
            if not ok:
                bad += 1; print("MISMATCH rank", rank, "seed", seed, "case", ci, acts, flush=True)
        stop = torch.tensor([1 if time.time() - t0 > budget else 0]); dist.all_reduce(stop, op=dist.ReduceOp.MAX)  # the ranks stop together
        if int(stop): break
    raw.kai_hostsim_set_victim_shard(0, 0, None, None); raw.kai_hostsim_set_shard(0, 1, 0, None, None)
    out.put((rank, tot, bad, ex, seed)); T.pkg.dist.finish()


if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2]); world = int(sys.argv[3]) if len(sys.argv) > 3 else 2; engines = int(sys.argv[4]) if len(sys.argv) > 4 else 2; cap = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); out = ctx.Queue(); budget = float(os.environ.get("CAMPAIGN_SECONDS", "150"))
    procs = [ctx.Process(target=worker, args=(r, world, port, lo, hi, engines, cap, budget, out)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(out.get(timeout=budget + 600) for _ in range(world))
    for p in procs: p.join(timeout=60)
    print(f"victim waves over {world} ranks x {engines} engines (wave length {cap or 'default'}): cycles {res[0][1]}, mismatches {sum(r[2] for r in res)}, collectives per rank {res[0][3]}, last seed {res[0][4]}")
