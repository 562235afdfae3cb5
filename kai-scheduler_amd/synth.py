"""Synthetic cluster snapshots of BASELINE.json's configs.

All generators are deterministic in `seed` (base 0x4B4149 + config index), produce integer-valued quantities
(< 2**53) and distinct creation timestamps so that the reference's order is well defined.
Node names are `node-%06d` (string order == numeric order) unless `lexi_names` asks for `node-%d`.
"""
from __future__ import annotations

import numpy as np

from . import abi

GIB = float(1 << 30)
NOW_NS = 1_780_000_000 * 1_000_000_000  # "now" of the synthetic cycles (minruntime plugin)
SEED0 = 0x4B4149


def _queue_tree(levels, rng, total_gpu, zipf=False, limits_frac=0.0, prios=(100,), oqws=(1.0,), usage_max=0.0):
    """levels=[a,b,..] → a top queues, each with b children, … ; returns dict of arrays + leaf list."""
    parent, depth_nodes = [], [[-1]]
    names = []
    cur = [-1]
    for li, width in enumerate(levels):
        nxt = []
        for p in cur:
            for k in range(width):
                idx = len(parent)
                parent.append(p)
                names.append(("q" if li == len(levels) - 1 else f"l{li}") + f"-{idx:05d}")
                nxt.append(idx)
        cur = nxt
    Q = len(parent)
    parent = np.array(parent, np.int32)
    leaves = np.array(cur, np.int32)
    deserved = np.full((3, Q), -1.0)
    limit = np.full((3, Q), -1.0)
    oqw = np.ones((3, Q))
    if zipf:
        w = 1.0 / np.arange(1, len(leaves) + 1) ** 1.1
        w = w[rng.permutation(len(leaves))]
        share = np.floor(0.8 * total_gpu * w / w.sum())
    else:
        share = np.full(len(leaves), np.floor(total_gpu / max(len(leaves), 1)))
    deserved[abi.Q_GPU, leaves] = share
    # inner queues deserve the sum of their children (bottom-up)
    for q in range(Q - 1, -1, -1):
        if parent[q] >= 0:
            if deserved[abi.Q_GPU, parent[q]] < 0:
                deserved[abi.Q_GPU, parent[q]] = 0
            deserved[abi.Q_GPU, parent[q]] += deserved[abi.Q_GPU, q]
    if limits_frac > 0:
        lim = rng.random(len(leaves)) < limits_frac
        limit[abi.Q_GPU, leaves[lim]] = 2 * deserved[abi.Q_GPU, leaves[lim]] + 8
    oqw[abi.Q_GPU, :] = rng.choice(np.array(oqws, float), size=Q)
    usage = rng.random((3, Q)) * usage_max if usage_max > 0 else np.zeros((3, Q))
    prio = rng.choice(np.array(prios, np.int32), size=Q).astype(np.int32)
    created = (np.arange(Q, dtype=np.int64) + 1) * 60_000_000_000
    return dict(parent=parent, names=names, leaves=leaves, deserved=deserved, limit=limit, oqw=oqw, usage=usage, prio=prio, created=created)


def make_snapshot(n_nodes, n_pending_pods, seed, *, queue_levels=(2, 2), prefill=0.3, gpu_mix=((8, 1.0),), cpu_only_frac=0.0,
                  gang_sizes=(1, 2, 4, 8), gang_p=(0.4, 0.2, 0.2, 0.2), gpus_per_pod=(1, 2, 4, 8), zipf=False, limits_frac=0.0,
                  queue_prios=(100,), oqws=(1.0,), nonpreempt_frac=0.0, usage_max=0.0, lexi_names=False, single_pod_jobs=False,
                  uniform_nodes=False, cpu_per_gpu=4000.0, mem_per_gpu=32 * GIB, elastic_frac=0.0, multi_podset_frac=0.0, task_prio_frac=0.0,
                  inner_limits_frac=0.0) -> abi.Snapshot:
    """inner_limits_frac: that share of the INNER queues gets a GPU limit of 0.3 .. 1.2 times what their subtree deserves (drawn from a generator of its own: the snapshots of
    every other parameter set stay what they were) — a parent queue that turns jobs away while its leaves would still take them (capacity gate up the chain, proportion.go)."""
    rng = np.random.default_rng(seed)
    R = 4
    N = n_nodes
    # ---- nodes
    kinds = np.array([g for g, _ in gpu_mix]); probs = np.array([p for _, p in gpu_mix], float); probs /= probs.sum()
    gpus = kinds[rng.choice(len(kinds), size=N, p=probs)].astype(float) if N else np.zeros(0)
    alloc = np.zeros((R, N))
    if uniform_nodes:  # test-fixture defaults (test_utils/nodes_fake/nodes.go:31-36)
        alloc[abi.RES_CPU] = 20000.0; alloc[abi.RES_MEM] = 20e9
    else:
        alloc[abi.RES_CPU] = rng.integers(64, 193, size=N) * 1000.0
        alloc[abi.RES_MEM] = rng.integers(256, 1025, size=N) * GIB
    alloc[abi.RES_GPU] = gpus
    alloc[abi.RES_PODS] = 110
    node_names = [(f"node-{i}" if lexi_names else f"node-{i:06d}") for i in range(N)]
    total_gpu = float(gpus.sum())
    qt = _queue_tree(list(queue_levels), rng, total_gpu, zipf=zipf, limits_frac=limits_frac, prios=queue_prios, oqws=oqws, usage_max=usage_max)
    leaves = qt["leaves"]
    if inner_limits_frac > 0:
        rng2 = np.random.default_rng(seed + 7717)
        inner = np.setdiff1d(np.arange(len(qt["parent"])), leaves)
        pick = inner[rng2.random(len(inner)) < inner_limits_frac]
        qt["limit"][abi.Q_GPU, pick] = np.floor(qt["deserved"][abi.Q_GPU, pick] * (0.3 + 0.9 * rng2.random(len(pick)))) + 1

    # ---- pending gangs
    mean_size = 1.0 if single_pod_jobs else max(float(np.dot(gang_sizes, gang_p)), 1e-9)
    est = int(n_pending_pods / mean_size * 1.5) + 64
    sizes = np.ones(est, np.int64) if single_pod_jobs else rng.choice(np.array(gang_sizes), size=est, p=np.array(gang_p))
    csum = np.cumsum(sizes)
    nj = int(np.searchsorted(csum, n_pending_pods, side="left")) + 1 if n_pending_pods > 0 else 0
    sizes = sizes[:nj]
    if nj:
        sizes[-1] -= csum[nj - 1] - n_pending_pods  # trim the last gang to hit the pod count exactly
        if sizes[-1] <= 0:
            sizes = sizes[:-1]; nj -= 1
    is_cpu = rng.random(nj) < cpu_only_frac
    g_pod = np.where(is_cpu, 0, rng.choice(np.array(gpus_per_pod), size=nj)).astype(float)
    if single_pod_jobs:
        g_pod = np.where(is_cpu, 0, 1.0)
    cpu_pod = np.where(is_cpu, rng.integers(1, 17, size=nj) * 1000.0, cpu_per_gpu * g_pod)
    mem_pod = np.where(is_cpu, 4 * GIB, mem_per_gpu * g_pod)
    # ---- running filler: one running single-pod job per partly used node
    used = rng.binomial(gpus.astype(np.int64), prefill) if N else np.zeros(0, np.int64)
    run_nodes = np.nonzero(used > 0)[0]
    nr = len(run_nodes)
    J = nj + nr
    job_sizes = np.concatenate([sizes, np.ones(nr, np.int64)]).astype(np.int32)
    first_pod = np.concatenate([[0], np.cumsum(job_sizes)[:-1]]).astype(np.int32) if J else np.zeros(0, np.int32)
    P = int(job_sizes.sum())
    pod_job = np.repeat(np.arange(J, dtype=np.int32), job_sizes)
    pod_req = np.zeros((R, P))
    pod_req[abi.RES_CPU] = np.repeat(np.concatenate([cpu_pod, 4000.0 * used[run_nodes]]), job_sizes)
    pod_req[abi.RES_MEM] = np.repeat(np.concatenate([mem_pod, np.minimum(32 * GIB * used[run_nodes], alloc[abi.RES_MEM, run_nodes] - GIB)]), job_sizes)
    pod_req[abi.RES_GPU] = np.repeat(np.concatenate([g_pod, used[run_nodes].astype(float)]), job_sizes)
    pod_req[abi.RES_PODS] = 1.0
    # keep the filler inside its node's cpu
    fill_cpu = np.minimum(4000.0 * used[run_nodes], alloc[abi.RES_CPU, run_nodes] - 1000.0)
    pod_req[abi.RES_CPU, P - nr:] = fill_cpu
    pod_status = np.full(P, abi.POD_STATUS["Pending"], np.int32)
    pod_node = np.full(P, -1, np.int32)
    pod_status[P - nr:] = abi.POD_STATUS["Running"]
    pod_node[P - nr:] = run_nodes
    nonpre = rng.random(J) < nonpreempt_frac
    job_prio = np.where(nonpre, rng.choice(np.array([100, 125]), size=J), rng.choice(np.array([50, 75]), size=J) if nonpreempt_frac > 0 else 50).astype(np.int32)
    job_queue = leaves[rng.integers(0, len(leaves), size=J)].astype(np.int32) if J else np.zeros(0, np.int32)
    created = (rng.permutation(J).astype(np.int64) + 1) * 60_000_000_000  # distinct, not aligned with index order

    snap = abi.Snapshot(n_res=R)
    a = snap.arrays
    a["node_allocatable"] = alloc
    a["node_flags"] = np.zeros(N, np.uint32)
    a["node_gpu_count"] = gpus.astype(np.int32)
    a["node_name_rank"] = abi.rank_strings(node_names) if lexi_names else np.arange(N, dtype=np.uint32)
    # ---- pod-sets: one ("default") per job, or — for a fraction of the pending gangs — two named sub-groups; a fraction of the
    # single-pod-set gangs is elastic (minAvailable < pods: the job grows one pod per pop, allocate.go:69-72)
    n_ps = np.ones(J, np.int32)
    draw = lambda frac: (rng.random(J) < frac) if frac > 0 else np.zeros(J, bool)  # no RNG draw when the feature is off: the BASELINE configs stay as they were
    multi = draw(multi_podset_frac) & (job_sizes >= 2) & (np.arange(J) < nj)
    n_ps[multi] = 2
    first_ps = np.concatenate([[0], np.cumsum(n_ps)[:-1]]).astype(np.int32) if J else np.zeros(0, np.int32)
    S = int(n_ps.sum())
    podset_job = np.repeat(np.arange(J, dtype=np.int32), n_ps)
    podset_min = np.zeros(S, np.int32); podset_rank = np.zeros(S, np.uint32)
    pod_podset = np.zeros(P, np.int32)
    elastic = draw(elastic_frac) & (job_sizes >= 2) & ~multi & (np.arange(J) < nj)
    for j in range(J):
        b, n, s0 = int(first_pod[j]), int(job_sizes[j]), int(first_ps[j])
        if n_ps[j] == 1:
            podset_min[s0] = max(1, n // 2) if elastic[j] else n
            pod_podset[b:b + n] = s0
        else:
            h = n // 2
            podset_min[s0], podset_min[s0 + 1] = h, n - h
            podset_rank[s0], podset_rank[s0 + 1] = (1, 0) if rng.random() < 0.5 else (0, 1)  # name order independent of index order
            pod_podset[b:b + h] = s0; pod_podset[b + h:b + n] = s0 + 1
    a["pod_req"] = pod_req; a["pod_job"] = pod_job; a["pod_podset"] = pod_podset
    a["pod_status"] = pod_status; a["pod_node"] = pod_node
    a["pod_uid_rank"] = np.arange(P, dtype=np.uint32)  # pod UIDs are zero-padded by construction
    if task_prio_frac > 0:  # task-order label on some pods (plugins/taskorder/task_order.go:28-63)
        has = rng.random(P) < task_prio_frac
        a["pod_flags"] = np.where(has, abi.POD_HAS_TASK_PRIORITY, 0).astype(np.uint32)
        a["pod_task_priority"] = rng.integers(0, 4, size=P).astype(np.int32)
    a["podset_job"] = podset_job; a["podset_min_available"] = podset_min; a["podset_name_rank"] = podset_rank
    a["job_queue"] = job_queue; a["job_priority"] = job_prio; a["job_preemptible"] = (job_prio < 100).astype(np.int32)
    a["job_created_ns"] = created; a["job_uid_rank"] = np.arange(J, dtype=np.uint32)
    a["job_signature"] = job_prio.astype(np.int64)  # the synthetic pods carry no selector / affinity / toleration: the priority class is what distinguishes signatures
    a["job_first_pod"] = first_pod; a["job_n_pods"] = job_sizes; a["job_first_podset"] = first_ps; a["job_n_podsets"] = n_ps
    a["queue_parent"] = qt["parent"]; a["queue_priority"] = qt["prio"]; a["queue_created_ns"] = qt["created"]
    a["queue_uid_rank"] = np.arange(len(qt["parent"]), dtype=np.uint32)
    a["queue_deserved"] = qt["deserved"]; a["queue_limit"] = qt["limit"]; a["queue_oqw"] = qt["oqw"]; a["queue_usage"] = qt["usage"]
    snap.node_names = node_names
    snap.queue_names = qt["names"]
    snap.finalize()
    return snap


def make_crowded_snapshot(n_nodes, seed, *, fill=0.9, n_pending_jobs=12, queue_levels=(2, 2), hog_frac=0.6, elastic_frac=0.3,
                          nonpreempt_frac=0.1, cpu_only_frac=0.0, minruntime=False, two_podsets_frac=0.0) -> abi.Snapshot:
    """A nearly full cluster for the victim actions (reclaim / preempt / consolidation): ~`fill` of the GPUs are held by
    Running gangs (1-4 pods x 1-4 GPUs, some elastic = more pods than minAvailable), `hog_frac` of them in the first leaf queue so that
    it sits over its fair share; pending gangs wait in every queue with mixed priorities (train 50 / build 100 non-preemptible)."""
    rng = np.random.default_rng(seed)
    R, N = 4, n_nodes
    alloc = np.zeros((R, N)); alloc[abi.RES_CPU] = 64000.0; alloc[abi.RES_MEM] = 512 * GIB; alloc[abi.RES_GPU] = 8.0; alloc[abi.RES_PODS] = 110
    free = np.full(N, 8.0)
    qt = _queue_tree(list(queue_levels), rng, 8.0 * N)
    leaves = qt["leaves"]
    jobs = []  # (queue, prio, preemptible, min_available, [(gpus, status, node)])
    target = fill * 8.0 * N
    while 8.0 * N - free.sum() < target:
        size = int(rng.integers(1, 5)); g = float(rng.choice([1, 1, 2, 4]))
        pods = []
        for _ in range(size):
            cand = np.nonzero(free >= g)[0]
            if len(cand) == 0:
                break
            n = int(rng.choice(cand)); free[n] -= g; pods.append((g, "Running", n))
        if not pods:
            break
        q = int(leaves[0]) if rng.random() < hog_frac else int(rng.choice(leaves))
        nonpre = rng.random() < nonpreempt_frac
        mn = len(pods) if (len(pods) < 2 or rng.random() >= elastic_frac) else max(1, len(pods) // 2)
        jobs.append((q, int(rng.choice([100, 125])) if nonpre else int(rng.choice([50, 60, 75])), 0 if nonpre else 1, mn, pods))
    for _ in range(n_pending_jobs):
        size = int(rng.integers(1, 4)); g = 0.0 if rng.random() < cpu_only_frac else float(rng.choice([1, 2, 4, 8]))
        nonpre = rng.random() < nonpreempt_frac
        jobs.append((int(rng.choice(leaves)), int(rng.choice([100, 125])) if nonpre else int(rng.choice([50, 60, 75])), 0 if nonpre else 1, size,
                     [(g, "Pending", -1)] * size))
    order = rng.permutation(len(jobs)); jobs = [jobs[i] for i in order]
    J = len(jobs); sizes = np.array([len(j[4]) for j in jobs], np.int32); P = int(sizes.sum())
    first_pod = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    pod_req = np.zeros((R, P)); pod_status = np.zeros(P, np.int32); pod_node = np.full(P, -1, np.int32)
    p = 0
    for j in jobs:
        for g, st, n in j[4]:
            pod_req[abi.RES_GPU, p] = g; pod_req[abi.RES_CPU, p] = 4000.0 * g if g > 0 else 2000.0; pod_req[abi.RES_MEM, p] = 16 * GIB * max(g, 0.25); pod_req[abi.RES_PODS, p] = 1.0
            pod_status[p] = abi.POD_STATUS[st]; pod_node[p] = n; p += 1
    snap = abi.Snapshot(n_res=R); a = snap.arrays
    a["node_allocatable"] = alloc; a["node_flags"] = np.zeros(N, np.uint32); a["node_gpu_count"] = np.full(N, 8, np.int32); a["node_name_rank"] = np.arange(N, dtype=np.uint32)
    # pod-sets: one per job, or (a fraction of the pending gangs of >= 2 pods) two named ones of half the pods each
    two = np.array([two_podsets_frac > 0 and len(j[4]) >= 2 and j[4][0][1] == "Pending" and rng.random() < two_podsets_frac for j in jobs], bool)
    n_ps = np.where(two, 2, 1).astype(np.int32); first_ps = np.concatenate([[0], np.cumsum(n_ps)[:-1]]).astype(np.int32)
    S = int(n_ps.sum()); podset_job = np.repeat(np.arange(J, dtype=np.int32), n_ps); podset_min = np.zeros(S, np.int32); podset_rank = np.zeros(S, np.uint32)
    pod_podset = np.zeros(P, np.int32)
    for ji, j in enumerate(jobs):
        b, n, s0 = int(first_pod[ji]), len(j[4]), int(first_ps[ji])
        if two[ji]:
            h = n // 2; podset_min[s0], podset_min[s0 + 1] = h, n - h
            podset_rank[s0], podset_rank[s0 + 1] = (1, 0) if rng.random() < 0.5 else (0, 1)
            pod_podset[b:b + h] = s0; pod_podset[b + h:b + n] = s0 + 1
        else:
            podset_min[s0] = j[3]; pod_podset[b:b + n] = s0
    a["pod_req"] = pod_req; a["pod_job"] = np.repeat(np.arange(J, dtype=np.int32), sizes); a["pod_podset"] = pod_podset
    a["pod_status"] = pod_status; a["pod_node"] = pod_node; a["pod_uid_rank"] = np.arange(P, dtype=np.uint32)
    a["podset_job"] = podset_job; a["podset_min_available"] = podset_min; a["podset_name_rank"] = podset_rank
    a["job_queue"] = np.array([j[0] for j in jobs], np.int32); a["job_priority"] = np.array([j[1] for j in jobs], np.int32)
    a["job_preemptible"] = np.array([j[2] for j in jobs], np.int32)
    a["job_signature"] = a["job_priority"].astype(np.int64)
    a["job_created_ns"] = (rng.permutation(J).astype(np.int64) + 1) * 60_000_000_000; a["job_uid_rank"] = np.arange(J, dtype=np.uint32)
    a["job_first_pod"] = first_pod; a["job_n_pods"] = sizes; a["job_first_podset"] = first_ps; a["job_n_podsets"] = n_ps
    a["queue_parent"] = qt["parent"]; a["queue_priority"] = qt["prio"]; a["queue_created_ns"] = qt["created"]; a["queue_uid_rank"] = np.arange(len(qt["parent"]), dtype=np.uint32)
    a["queue_deserved"] = qt["deserved"]; a["queue_limit"] = qt["limit"]; a["queue_oqw"] = qt["oqw"]; a["queue_usage"] = qt["usage"]
    if minruntime:  # minruntime plugin inputs: start times up to 2 h before NOW_NS, min-runtimes of 0 / 30 / 60 min on some queues
        Qn = len(qt["parent"]); running = np.array([any(st == "Running" for _, st, _ in j[4]) for j in jobs])
        a["job_last_start_ns"] = np.where(running, NOW_NS - rng.integers(0, 7200, size=J).astype(np.int64) * 1_000_000_000, 0).astype(np.int64)
        pick = lambda: np.where(rng.random(Qn) < 0.5, -1, rng.choice(np.array([0, 1800, 3600]), size=Qn) * 1_000_000_000).astype(np.int64)
        a["queue_preempt_min_runtime_ns"] = pick(); a["queue_reclaim_min_runtime_ns"] = pick()
    snap.node_names = [f"node-{i:06d}" for i in range(N)]; snap.queue_names = qt["names"]
    snap.finalize()
    return snap


def add_topology(snap: abi.Snapshot, seed: int, zones: int = 8, racks_per_zone: int = 40, req_rack_frac: float = 0.5, pref_rack_frac: float = 0.2) -> abi.Snapshot:
    """Label the nodes `zone / rack` (one Topology CR with levels [zone, rack], plugins/topology/topology_plugin.go:57-110) and give a
    fraction of the pending gangs a topology constraint on their root sub-group set: requiredLevel = rack, or preferredLevel = rack with
    requiredLevel = zone (BASELINE config 4)."""
    rng = np.random.default_rng(seed ^ 0x70B0)
    a = snap.arrays
    N, J, S = snap.n_nodes, snap.n_jobs, snap.n_podsets
    racks = max(1, min(zones * racks_per_zone, N))
    zones = max(1, min(zones, racks))
    rack_of = (np.arange(N, dtype=np.int64) * racks // max(N, 1)).astype(np.int32)      # contiguous name ranges per rack
    zone_of = (rack_of.astype(np.int64) * zones // racks).astype(np.int32)
    # domains: zones first (level row 0), then racks (level row 1, parent = its zone); ids "z%03d" / "z%03d.r%04d" rank like their indices
    a["topo_level_off"] = np.array([0, 2], np.int32)
    a["node_domain"] = np.stack([zone_of, zones + rack_of]).astype(np.int32)
    rack_zone = np.zeros(racks, np.int32); rack_zone[rack_of] = zone_of
    a["domain_level"] = np.concatenate([np.zeros(zones, np.int32), np.ones(racks, np.int32)])
    a["domain_parent"] = np.concatenate([np.full(zones, -1, np.int32), rack_zone])
    a["domain_id_rank"] = np.concatenate([np.arange(zones), np.arange(racks)]).astype(np.uint32)
    # one root sub-group set per job (jobs_fake/jobs.go:116-123), every pod-set directly under it
    pending = np.zeros(J, bool); np.logical_or.at(pending, a["pod_job"], a["pod_status"] == abi.POD_STATUS["Pending"])
    u = rng.random(J)
    req_rack = pending & (a["job_n_pods"] >= 2) & (u < req_rack_frac)
    pref_rack = pending & (a["job_n_pods"] >= 2) & ~req_rack & (u < req_rack_frac + pref_rack_frac)
    a["group_job"] = np.arange(J, dtype=np.int32); a["group_parent"] = np.full(J, -1, np.int32); a["group_name_rank"] = np.zeros(J, np.uint32)
    a["group_topology"] = np.where(req_rack | pref_rack, 0, -1).astype(np.int32)
    a["group_required_level"] = np.where(req_rack, 1, np.where(pref_rack, 0, -1)).astype(np.int32)
    a["group_preferred_level"] = np.where(pref_rack, 1, -1).astype(np.int32)
    a["job_root_group"] = np.arange(J, dtype=np.int32)
    a["podset_group"] = a["podset_job"].astype(np.int32)
    a["podset_topology"] = np.full(S, -1, np.int32); a["podset_required_level"] = np.full(S, -1, np.int32); a["podset_preferred_level"] = np.full(S, -1, np.int32)
    return snap.finalize()


def add_replica_topology(snap: abi.Snapshot, seed: int, zones: int = 2, nodes_per_rack: int = 3) -> abi.Snapshot:
    """zone / rack labels plus, for every two-pod-set job, the sub-group tree of the reference's replica fixtures
    (actions/allocate/allocateTopology_test.go): a root SubGroupSet (preferred level zone for half of them) with two child SubGroupSets,
    one pod-set each, BOTH with requiredLevel = rack — two sub-groups on one constraint level, the case in which the
    TopologyAwareIdleGpus scenario filter matters.  Some single-pod-set gangs get requiredLevel = rack on their root."""
    rng = np.random.default_rng(seed ^ 0x7E91)
    a = snap.arrays
    N, J, S = snap.n_nodes, snap.n_jobs, snap.n_podsets
    racks = max(1, (N + nodes_per_rack - 1) // nodes_per_rack); zones = max(1, min(zones, racks))
    rack_of = (np.arange(N) // nodes_per_rack).astype(np.int32); zone_of = (rack_of.astype(np.int64) * zones // racks).astype(np.int32)
    a["topo_level_off"] = np.array([0, 2], np.int32); a["node_domain"] = np.stack([zone_of, zones + rack_of]).astype(np.int32)
    rack_zone = np.zeros(racks, np.int32); rack_zone[rack_of] = zone_of
    a["domain_level"] = np.concatenate([np.zeros(zones, np.int32), np.ones(racks, np.int32)]); a["domain_parent"] = np.concatenate([np.full(zones, -1, np.int32), rack_zone])
    a["domain_id_rank"] = np.concatenate([np.arange(zones), np.arange(racks)]).astype(np.uint32)
    gj, gp, gr, gt, gq, gf, root = [], [], [], [], [], [], []
    ps_group = np.zeros(S, np.int32)
    pending = np.zeros(J, bool); np.logical_or.at(pending, a["pod_job"], a["pod_status"] == abi.POD_STATUS["Pending"])
    for j in range(J):
        r = len(gj); root.append(r)
        s0, n = int(a["job_first_podset"][j]), int(a["job_n_podsets"][j])
        if n == 2:
            pref = rng.random() < 0.5
            gj.append(j); gp.append(-1); gr.append(0); gt.append(0 if pref else -1); gq.append(-1); gf.append(0 if pref else -1)
            for k in range(2):
                gj.append(j); gp.append(r); gr.append(k); gt.append(0); gq.append(1); gf.append(-1)
                ps_group[s0 + k] = r + 1 + k
        else:
            req = pending[j] and a["job_n_pods"][j] >= 2 and rng.random() < 0.4
            gj.append(j); gp.append(-1); gr.append(0); gt.append(0 if req else -1); gq.append(1 if req else -1); gf.append(-1)
            ps_group[s0] = r
    a["group_job"] = np.array(gj, np.int32); a["group_parent"] = np.array(gp, np.int32); a["group_name_rank"] = np.array(gr, np.uint32)
    a["group_topology"] = np.array(gt, np.int32); a["group_required_level"] = np.array(gq, np.int32); a["group_preferred_level"] = np.array(gf, np.int32)
    a["job_root_group"] = np.array(root, np.int32); a["podset_group"] = ps_group
    a["podset_topology"] = np.full(S, -1, np.int32); a["podset_required_level"] = np.full(S, -1, np.int32); a["podset_preferred_level"] = np.full(S, -1, np.int32)
    return snap.finalize()


def config(idx: int, scale: float = 1.0, seed_offset: int = 0, mixed: bool = False) -> tuple[abi.Snapshot, abi.KaiConfig, str]:
    """BASELINE.json configs[idx] (0-based) → (snapshot, config, description).  `scale` shrinks node and pod counts together.
    mixed (config 5 only): the shape SURVEY.md section 8d writes down for it — nodes labelled 8 zones x 64 racks, 5 % of the gangs with a required rack, 5 % elastic
    gangs (minMember below their pod count), the cluster half full of running pods, the minruntime plugin on: jobs the batch path of the allocate action does not take."""
    seed = SEED0 + idx + seed_offset
    n = lambda x: max(1, int(round(x * scale)))
    if idx == 4 and mixed:
        s = make_snapshot(n(65536), n(1000000), seed, queue_levels=(8, 16, 16), prefill=0.5, zipf=True, limits_frac=0.2, queue_prios=(100, 200),
                          oqws=(1.0, 2.0, 4.0), nonpreempt_frac=0.1, usage_max=0.3, elastic_frac=0.05)
        add_topology(s, seed, zones=min(8, max(1, n(8))), racks_per_zone=max(1, min(64, n(65536) // (8 * 4))), req_rack_frac=0.05, pref_rack_frac=0.0)
        cfg = abi.default_config(k_value=0.5)
        cfg.plugins |= abi.PLUGINS.get("minruntime", 0)
        return s, cfg, f"C5-mixed {n(65536)}n x {n(1000000)}p zone/rack labels, 5% topology gangs, 5% elastic gangs, minruntime"
    if idx == 0:   # C1: 16 nodes / 64 single-pod jobs, single queue, bin-pack (plumbing)
        s = make_snapshot(16, 64, seed, queue_levels=(1, 1), prefill=0.0, single_pod_jobs=True, uniform_nodes=True, cpu_per_gpu=1000.0, mem_per_gpu=1e9)
        return s, abi.default_config(), "C1 16n x 64p single-queue bin-pack"
    if idx == 1:   # C2: 1k nodes x 10k pods, gangs + bin-pack
        s = make_snapshot(n(1000), n(10000), seed, queue_levels=(2, 2), prefill=0.3)
        return s, abi.default_config(), f"C2 {n(1000)}n x {n(10000)}p gangs + bin-pack"
    if idx == 2:   # C3: 10k x 100k, 2-level queues + proportion/DRF (pure DRF: kValue 0)
        s = make_snapshot(n(10000), n(100000), seed, queue_levels=(10, 10), prefill=0.3, gpu_mix=((8, .7), (4, .2), (0, .1)), cpu_only_frac=0.2,
                          zipf=True, limits_frac=0.2, queue_prios=(100, 200), oqws=(1.0, 2.0, 4.0), nonpreempt_frac=0.1)
        return s, abi.default_config(k_value=0.0), f"C3 {n(10000)}n x {n(100000)}p 2-level queues + DRF"
    if idx == 4:   # C5: 64k x 1M, 3-level queues, time-based fair-share (kValue 0.5, historical usage)
        s = make_snapshot(n(65536), n(1000000), seed, queue_levels=(8, 16, 16), prefill=0.3, zipf=True, limits_frac=0.2, queue_prios=(100, 200),
                          oqws=(1.0, 2.0, 4.0), nonpreempt_frac=0.1, usage_max=0.3)
        return s, abi.default_config(k_value=0.5), f"C5 {n(65536)}n x {n(1000000)}p full chain + time-based fair-share"
    if idx == 3:   # C4: 10k x 100k, zone/rack topology, cluster 85 % full of preemptible Running jobs; actions allocate, consolidation, reclaim
        s = make_snapshot(n(10000), n(100000), seed, queue_levels=(10, 10), prefill=0.85, zipf=True, limits_frac=0.2, queue_prios=(100, 200),
                          oqws=(1.0, 2.0, 4.0), nonpreempt_frac=0.1)
        add_topology(s, seed, zones=min(8, max(1, n(8))), racks_per_zone=max(1, min(40, n(10000) // (8 * 4))))
        return s, abi.default_config(k_value=0.0, max_consolidation_preemptees=16), f"C4 {n(10000)}n x {n(100000)}p zone/rack topology + consolidation + reclaim"
    raise ValueError(f"no BASELINE config {idx}")


def add_fractions(snap: abi.Snapshot, seed: int, frac: float = 0.4, portions=(0.2, 0.25, 0.5, 0.75), gpu_memory: int = 100, memory_requests: float = 0.0) -> abi.Snapshot:
    """Turn a share of the one-GPU pods into requests for a fraction of one device (ABI v4: pod_gpu_portion / pod_gpu_group / node_gpu_memory).
    Placed ones are packed first-fit into shared-GPU groups of their node (numeric group names 0, 1, …; every group takes one of the GPUs the
    original pods held, so the node accounting of the reference — api/node_info/gpu_sharing_node_info.go — stays within the node's devices);
    pending ones ask for their portion.  memory_requests: that share of them asks for MiB of one device instead (annotation gpu-memory, ABI v5
    pod_gpu_memory: the GPU column and the portion are 0, the request is portion x gpu_memory MiB)."""
    rng = np.random.default_rng(seed ^ 0xF2AC)
    a = snap.arrays
    P, N = snap.n_pods, snap.n_nodes
    S = abi.POD_STATUS
    active = S["Running"] | S["Bound"] | S["Binding"] | S["Allocated"]
    portion = np.zeros(P, np.float64); group = np.full(P, -1, np.int32); gmem = np.zeros(P, np.int64)
    fill = [[] for _ in range(N)]  # per node: used portion (in 1/100) of every group
    for p in range(P):
        if a["pod_req"][abi.RES_GPU, p] != 1.0 or rng.random() >= frac:
            continue
        st = int(a["pod_status"][p])
        if st != S["Pending"] and not (st & active):
            continue
        v = float(rng.choice(portions)); portion[p] = v; a["pod_req"][abi.RES_GPU, p] = np.floor(v * 100.0 + 0.5) / 100.0  # GPUs(): 1/100 fixed point
        if st & active:
            n = int(a["pod_node"][p]); need = int(round(v * 100))
            for g, used in enumerate(fill[n]):
                if used + need <= 100:
                    fill[n][g] += need; group[p] = g; break
            else:
                fill[n].append(need); group[p] = len(fill[n]) - 1
        if memory_requests > 0 and rng.random() < memory_requests:
            gmem[p] = int(round(v * gpu_memory)); portion[p] = 0.0; a["pod_req"][abi.RES_GPU, p] = 0.0
    a["pod_gpu_portion"] = portion; a["pod_gpu_group"] = group; a["node_gpu_memory"] = np.full(N, gpu_memory, np.int64)
    if memory_requests > 0: a["pod_gpu_memory"] = gmem
    return snap.finalize()


def add_mig(snap: abi.Snapshot, seed: int, node_frac: float = 0.3, pod_frac: float = 0.5, legacy_frac: float = 0.05) -> abi.Snapshot:
    """Turn a share of the GPU nodes into MIG nodes with MigStrategy "mixed" (ABI v5 res_mig_*): two more resource rows, nvidia.com/mig-1g.10gb and
    nvidia.com/mig-2g.20gb, hold their instances (as many 1g instances as the node had GPUs plus half as many 2g ones), their whole-GPU count becomes 0
    (nodes_fake/nodes.go:71-73), and the whole-GPU pods that run there ask for 1g instances instead (same GPU quota: weight 1).  A share of the pending
    GPU pods ask for MIG instances as well (2g ones for even counts); a few pods are legacy MIG tasks (never schedulable; the node of a running one takes
    no MIG request, api/node_info/node_info.go:315-359)."""
    rng = np.random.default_rng(seed ^ 0x316)
    a = snap.arrays
    P, N, R0 = snap.n_pods, snap.n_nodes, snap.n_res
    if R0 + 2 > 8:
        raise ValueError("no free resource rows for MIG profiles")
    S = abi.POD_STATUS
    active = S["Running"] | S["Bound"] | S["Binding"] | S["Allocated"] | S["Pipelined"] | S["Releasing"]
    alloc = np.vstack([a["node_allocatable"], np.zeros((2, N))]); req = np.vstack([a["pod_req"], np.zeros((2, P))])
    r1, r2 = R0, R0 + 1
    mig_node = np.zeros(N, bool)
    for n in range(N):
        g = alloc[abi.RES_GPU, n]
        if g > 0 and g == int(g) and rng.random() < node_frac:
            mig_node[n] = True
            alloc[r1, n] = g; alloc[r2, n] = int(g) // 2; alloc[abi.RES_GPU, n] = 0
            a["node_flags"][n] |= abi.NODE_MIG_ENABLED | abi.NODE_MIG_MIXED
    flags = a["pod_flags"].copy() if "pod_flags" in a else np.zeros(P, np.uint32)
    for p in range(P):
        g = req[abi.RES_GPU, p]
        if not (g > 0 and g == int(g)):
            continue
        st = int(a["pod_status"][p]); n = int(a["pod_node"][p])
        if st & active and n >= 0:
            if mig_node[n]:
                req[r1, p] = g; req[abi.RES_GPU, p] = 0
                if rng.random() < legacy_frac: flags[p] |= abi.POD_LEGACY_MIG
        elif st == S["Pending"] and rng.random() < pod_frac:
            if int(g) % 2 == 0 and rng.random() < 0.5: req[r2, p] = int(g) // 2
            else: req[r1, p] = g
            req[abi.RES_GPU, p] = 0
            if rng.random() < legacy_frac: flags[p] |= abi.POD_LEGACY_MIG
    a["node_allocatable"] = alloc; a["pod_req"] = req; a["pod_flags"] = flags
    mg = np.zeros(R0 + 2, np.int32); mm = np.zeros(R0 + 2, np.int64); mg[r1], mm[r1], mg[r2], mm[r2] = 1, 10, 2, 20
    a["res_mig_gpus"] = mg; a["res_mig_memory"] = mm
    snap.n_res = R0 + 2
    return snap.finalize()


def add_predicate_features(snap: abi.Snapshot, seed: int, *, nominated_frac=0.15, not_ready_frac=0.1, worker_label_frac=0.8, foreign_frac=0.2,
                           oversized_frac=0.05) -> abi.Snapshot:
    """Exercise the predicate / node-order corners the BASELINE configs leave untouched:
      * status.nominatedNodeName on some pending pods (plugins/nominatednode/nominatednode.go:29-41: +1e6 for that node),
      * nodes that fail CheckNodeConditionPredicate (scheduler_util/scheduler_utils.go:12-40 → KAI_NODE_NOT_READY),
      * GPU / CPU worker labels on most nodes, for restrictSchedulingNodes (plugins/predicates/predicates.go:243-259),
      * Running pods of another scheduler (subtracted from the proportion totals, plugins/proportion/proportion.go:276-285),
      * pending pods that ask for more than any node has (what MaxNodeResourcesPredicate.PreFilter turns away up front,
        k8s_internal/predicates/maxNodeResources.go:59-96; the engine reaches the same verdict node by node)."""
    rng = np.random.default_rng(seed ^ 0x9E37)
    a = snap.arrays
    N, P = snap.n_nodes, snap.n_pods
    S = abi.POD_STATUS
    flags = a["node_flags"].copy()
    flags[rng.random(N) < not_ready_frac] |= abi.NODE_NOT_READY
    gpu_nodes = a["node_allocatable"][abi.RES_GPU] > 0
    lab = rng.random(N) < worker_label_frac
    flags[lab & gpu_nodes] |= abi.NODE_GPU_WORKER
    flags[lab & ~gpu_nodes] |= abi.NODE_CPU_WORKER
    flags[rng.random(N) < 0.1] |= abi.NODE_CPU_WORKER  # some GPU nodes also take CPU-only work
    a["node_flags"] = flags
    nom = np.full(P, -1, np.int32)
    pending = a["pod_status"] == S["Pending"]
    pick = pending & (rng.random(P) < nominated_frac)
    if N:
        nom[pick] = rng.integers(0, N, size=int(pick.sum()))
    a["pod_nominated_node"] = nom
    pf = a["pod_flags"].copy() if "pod_flags" in a else np.zeros(P, np.uint32)
    running = a["pod_status"] == S["Running"]
    pf[running & (rng.random(P) < foreign_frac)] |= abi.POD_FOREIGN_SCHEDULER
    a["pod_flags"] = pf
    big = pending & (rng.random(P) < oversized_frac)
    a["pod_req"][abi.RES_CPU, big] = a["node_allocatable"][abi.RES_CPU].max(initial=0.0) + 1000.0
    return snap.finalize()
