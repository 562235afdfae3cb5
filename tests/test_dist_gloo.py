"""N > 1 plumbing on CPU: two processes over gloo exercise what bench.py does across GPUs — every rank builds and
schedules its own shard (different contents, same shape), a barrier brackets the timed region, the elapsed time is the
max over ranks and the work is summed.  No data-path collective exists (one scheduling shard per GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, HERE)
    import kai_testlib as T
    d = T.pkg.dist
    r, lr, w = d.init("gloo")
    assert (r, lr, w) == (rank, rank, world)
    snap, cfg, _ = T.pkg.synth.config(1, 0.05, seed_offset=d.shard_seed(0, rank))
    d.barrier()
    res = T.Oracle.run(snap, cfg)  # the shard's cycle (CPU oracle here; the HIP path on the GPU box)
    d.barrier()
    elapsed = d.max_over_ranks(1.0 + rank)          # rank 1 is "slower"
    total = d.sum_over_ranks(float(res.stats.decisions))
    out.put((rank, elapsed, total, int(res.stats.decisions), int(snap.arrays["pod_req"].sum())))
    d.finish()


def test_two_rank_shards_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs: p.start()
    rows = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs: p.join(timeout=60); assert p.exitcode == 0
    (r0, e0, t0, d0, c0), (r1, e1, t1, d1, c1) = rows
    assert e0 == e1 == 2.0                     # max over ranks
    assert t0 == t1 == float(d0 + d1)          # whole-job work = sum of the shards
    assert c0 != c1                            # the shards differ


def test_single_process_is_a_noop():
    sys.path.insert(0, HERE)
    import kai_testlib as T
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"): os.environ.pop(k, None)
    d = T.pkg.dist
    assert d.env_world() == (0, 0, 1)
    d.barrier(); assert d.max_over_ranks(3.5) == 3.5 and d.sum_over_ranks(2.0) == 2.0
