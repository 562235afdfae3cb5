"""Multi-GPU plumbing of the scheduling-cycle core: one process per GPU.

Two ways to use the GPUs of one node.  (1) Node-sharded group: `KaiCore(world=G, rank=g)` shards the NODE axis of one session; the
group's exchange step (offers + floors, an all-gather of a few KB per rank) goes through `torch.distributed` — backend "nccl" = RCCL over xGMI — from a
callback in core.py.  (2) Scheduling shards, the way the reference scales out (conf/scheduler_conf.go:95-112 node-pool label filter; pkg/operator
SchedulingShard): one independent session per GPU, no data-path collective.  This module holds what both need around the timed region: the process group, the
barrier and the max / sum over ranks that bench.py's contract asks for.
"""
from __future__ import annotations

import os


def env_world():
    """(rank, local_rank, world_size) as torch.distributed.run exports them; (0, 0, 1) when launched plainly."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_seed(base_seed: int, rank: int) -> int:
    """Every rank schedules its own shard: same shape, different contents (weak scaling)."""
    return base_seed + 1000003 * rank


def init(backend: str, device=None):
    """Join the process group (nccl = RCCL on the GPU box, gloo in the CPU tests).  No-op for a single process."""
    import torch.distributed as dist

    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


_HOST_GROUP = None


def host_group():
    """A gloo group over all ranks for collectives on HOST memory (the victim actions' wave exchange, core.py host_allgather=True): the default group when that is
    gloo already, else a second group beside the nccl one.  Collective: every rank calls it."""
    import torch.distributed as dist

    global _HOST_GROUP
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_backend() == "gloo":
        return None  # (None = the default group)
    if _HOST_GROUP is None:
        _HOST_GROUP = dist.new_group(backend="gloo")
    return _HOST_GROUP


def barrier(sync=None):
    """Device sync + rank barrier + device sync (the bracket bench.py's contract prescribes)."""
    import torch.distributed as dist

    if sync:
        sync()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if sync:
        sync()


def max_over_ranks(value: float, device="cpu") -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def finish():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
