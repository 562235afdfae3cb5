#!/usr/bin/env python3
"""bench.py — pod-placement decisions/s of the scheduling-cycle core on synthetic cluster snapshots.

    python bench.py --gpus 1 --steps K --warmup W [--config C5 --scale 1.0]

A "step" is one scheduling cycle of the hot path over one snapshot that is already resident in HBM:
kai_session_reset (the OnSessionOpen math: node accounting, proportion totals / usage / fair-share) followed by
the allocate Action.  The unit of work is one placement decision = one allocateTask (scores every node of the node
set, picks the best fitting one, ends in allocate / pipeline / fail — SURVEY.md section 8d).

Prints ONE JSON line with the contract's fields plus
  roofline     — the dominant kernel against the resource it saturates: on the batch path the fill kernel against instruction issue of its working wavefronts (with the decision
                 throughput against HBM — algorithmic bytes = decisions x (N x 128 B + 80 B), SURVEY 8d — beside it as `algorithmic_vs_streaming`); on the sequential engine k_action
                 against HBM
  cpu_baseline — the CPU oracle (a faithful single-thread restatement of the reference path, kind "port") timed on a
                 bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
B_NODE, B_POD_OUT = 128, 80  # algorithmic bytes per node of the node set / per decision

CONFIGS = {"C1": 0, "C2": 1, "C3": 2, "C4": 3, "C5": 4}


def pmc_traffic(workload, kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this command (tools/runs/gpu_prof.sh + tools/pmc_traffic.py write
    profiles/pmc_traffic.json; counters need their own passes, so this is not measured inside the timed run), or None when no pass is on file."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)[workload][kernel]["bytes_per_launch"]
    except (OSError, KeyError, ValueError, TypeError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.environ.get("KAI_BENCH_CONFIG", "C5"), choices=sorted(CONFIGS))
    ap.add_argument("--scale", type=float, default=None, help="shrinks nodes and pods together (default 1.0; C4: 0.01 — measured on one MI355X: 10 %% of config 4 in 34 s, 30 %% with --queue-depth 8 in 55 s, the FULL size with --queue-depth 8 in about 273 s, operations equal to the oracle's: a hash test of the -m gpu suite, DESIGN.md section 6)")
    ap.add_argument("--actions", default="", help="comma-separated actions of one cycle (default: allocate; C4: allocate,consolidation,reclaim)")
    ap.add_argument("--mixed", action="store_true", help="config C5 in the shape SURVEY 8d gives it: zone/rack labels, 5 %% topology gangs, 5 %% elastic gangs, minruntime — jobs the batch path leaves to the sequential engine")
    ap.add_argument("--fractions", type=float, default=0.0, help="that share of the one-GPU pods asks for a fraction of one device (shared GPUs: every decision is a brute-force scan of the nodes' GPU groups — the streaming kernel of SURVEY 8d)")
    ap.add_argument("--queue-depth", type=int, default=0, help="queueDepthPerAction for reclaim / preempt / consolidation (jobs tried per queue and action; 0 = the default: unlimited).  The reference's operator docs configure 5 .. 15 (docs/operator/README.md:64-66, scheduling-shards.md:51-54)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="decisions the CPU oracle is timed on (-1 = auto, 0 = skip)")
    args = ap.parse_args()

    # `--gpus N` launched plainly (no WORLD_SIZE): one rank per GPU is the contract, so the script starts itself under torch.distributed.run — a line that says
    # n_gpus N always comes from N ranks; a mismatch between --gpus and the launcher's world size is refused instead of reported
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus > 1 and env_world == 0:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    if env_world and env_world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {env_world} rank(s): refusing to report a line for another world size")

    import numpy as np
    import torch

    import __graft_entry__ as G
    pkg = G._load_pkg()
    rank, local_rank, world = pkg.dist.env_world()
    assert torch.cuda.is_available(), "bench.py needs a MI355X: the scheduling-cycle core has no CPU path"
    # KAI_BENCH_BACKEND=gloo KAI_BENCH_ONE_DEVICE=1: rehearsal of the multi-process path on a box with ONE GPU (every rank on device 0, the exchange staged through
    # host memory over gloo) — RCCL refuses two ranks on one device.  The driver's runs use neither.
    backend = os.environ.get("KAI_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("KAI_BENCH_ONE_DEVICE") == "1" else local_rank
    red_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        torch.cuda.set_device(dev_index)
        pkg.dist.init(backend, device=torch.device("cuda", dev_index))
    pkg.load_library()

    if args.scale is None:
        args.scale = float(os.environ.get("KAI_BENCH_SCALE", "0.01" if args.config == "C4" else "1.0"))
    idx = CONFIGS[args.config]
    actions = tuple(a for a in (args.actions or ("allocate,consolidation,reclaim" if args.config == "C4" else "allocate")).split(",") if a)
    t0 = time.time()
    # N > 1, default: REPLICAS — the GPUs as independent scheduling shards of the same shape (how KAI itself scales out: a scheduler instance per node pool,
    # conf/scheduler_conf.go:95-112), weak scaling, no data-path collective: for the allocate action this is the multi-GPU mode that pays ("replicas only").
    # KAI_BENCH_MULTI=shard: the ranks shard the NODE axis of ONE snapshot instead (per exchange every rank offers its best nodes per scan class, one all-gather over
    # RCCL / xGMI, the same virtual fill on every rank — built and protocol-tested, but the fill is one dependency chain and an exchange only adds to it: strong scaling below 1),
    # with the replicas as a second leg beside it.
    sharded = world > 1 and os.environ.get("KAI_BENCH_MULTI", "replicas") == "shard"
    snap, cfg, desc = pkg.synth.config(idx, args.scale, seed_offset=0 if (sharded or world == 1) else pkg.dist.shard_seed(0, rank), mixed=args.mixed)
    if args.queue_depth > 0:
        for a in ("consolidation", "reclaim", "preempt"):
            cfg.queue_depth[pkg.abi.ACTIONS[a]] = args.queue_depth
        desc += f", queueDepthPerAction {args.queue_depth} for the victim actions"
    if args.fractions > 0:
        pkg.synth.add_fractions(snap, 7, frac=args.fractions)
        desc += f" + {round(args.fractions * 100)} % of the one-GPU pods as fractions of one device"
    gen_s = time.time() - t0
    N = snap.n_nodes
    if os.environ.get("KAI_BENCH_ENGINE_MODE"):
        cfg.engine_mode = int(os.environ["KAI_BENCH_ENGINE_MODE"])  # 3 = sequential engine only (A/B against the batch path)

    def barrier():
        pkg.dist.barrier(torch.cuda.synchronize)

    # KAI_BENCH_RCCL=1: the node-sharded group's exchange from the library's own RCCL communicator (kai_shard_attach_rccl: ncclAllGather on its stream) instead of the
    # caller-supplied collective (the Python mirror's torch.distributed.all_gather_into_tensor with a host round trip per exchange)
    # victim actions of the cycle on a group: their simulation waves dealt out over the ranks (kai_victim_shard.hpp) — the exchange on host memory over a gloo group
    # beside the nccl one, or the library's own communicator with KAI_BENCH_RCCL=1
    rccl = os.environ.get("KAI_BENCH_RCCL") == "1"
    host_ag = True if (sharded and not rccl and any(a != "allocate" for a in actions)) else None
    if host_ag:
        pkg.KaiCore.host_group = pkg.dist.host_group()
    core = pkg.KaiCore(cfg, gpu_ids=(dev_index,), world=world, rank=rank, allgather="rccl" if rccl else None, host_allgather=host_ag) if sharded else pkg.KaiCore(cfg, gpu_ids=(dev_index,))
    t0 = time.time()
    ssn = core.open_session(snap)  # host → HBM once; the timed steps replay from the resident copy
    upload_s = time.time() - t0

    kernel_ms, open_ms, decisions, placed = [], [], 0, 0
    last_stats = [None]  # statistics of the cycle's last action
    first_ops = []

    trace_steps = os.environ.get("KAI_BENCH_TRACE") == "1"  # host clocks of a step's parts on stderr (diagnostic)

    def step(record):
        nonlocal decisions, placed
        t_s0 = time.perf_counter()
        ssn.reset()
        t_s1 = time.perf_counter()
        o_ms = ssn.stats().upload_ms
        n_ops, n_dec, k_ms_sum, st = 0, 0, 0.0, None
        ops_step = []
        view = len(actions) == 1  # one action per step: its operations stay where the C ABI wrote them (the session's ops_out buffer) instead of being copied once more
        for a in actions:  # one scheduling cycle: the configured actions in order on the same session
            t_a0 = time.perf_counter()
            o = ssn.execute(a, copy=not view); n_ops += len(o)
            t_a1 = time.perf_counter()
            ops_step.append(o)  # kept as returned (numpy over the C ABI's kai_op array): converting is left for after the timed region
            s_a = ssn.stats(); n_dec += int(s_a.decisions); k_ms_sum += s_a.kernel_ms
            st = s_a if st is None else st  # the engine counters reported below are the allocate action's
            last_stats[0] = s_a
            if trace_steps:
                print(f"bench step: {a}: execute {1e3 * (t_a1 - t_a0):.2f} ms (device {s_a.kernel_ms:.2f}), statistics {1e3 * (time.perf_counter() - t_a1):.3f} ms", file=sys.stderr)
        if trace_steps:
            print(f"bench step: reset {1e3 * (t_s1 - t_s0):.2f} ms (device {o_ms:.2f}), actions {1e3 * (time.perf_counter() - t_s1):.2f} ms", file=sys.stderr)
        if record:
            kernel_ms.append(k_ms_sum); open_ms.append(o_ms)
            decisions = n_dec; placed = n_ops
            first_ops[:] = ops_step
        return st

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    step_wall = []
    for _ in range(args.steps):
        t_w = time.perf_counter()
        st = step(True)
        step_wall.append((time.perf_counter() - t_w) * 1e3)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = pkg.dist.max_over_ranks(elapsed, device=red_dev)
    ssn.close(); core.destroy()

    first_ops = [(int(x["kind"]), int(x["pod"]), int(x["node"]), int(x["job"])) for o in first_ops for x in o]  # the last step's committed operations
    if sharded:  # one job: every rank committed the same operations
        total_decisions, total_placed = decisions * args.steps, placed * args.steps
    else:        # replicas: one scheduling shard per rank
        total_decisions = pkg.dist.sum_over_ranks(decisions * args.steps, device=red_dev)
        total_placed = pkg.dist.sum_over_ranks(placed * args.steps, device=red_dev)
    value = total_placed / elapsed  # BASELINE's metric: pod placements per second (committed allocate / pipeline operations; evictions of the victim actions are operations too)
    k_ms = float(np.mean(kernel_ms))
    b_dec = N * B_NODE + B_POD_OUT
    batch = int(st.reserved[4]) > 0
    lat = sorted((a + b) for a, b in zip(kernel_ms, open_ms))
    drained = int(st.reserved[3])
    engine = {"index_queries": int(st.reserved[0]), "brute_force_scans": int(st.node_scans), "drained_jobs": int(st.reserved[2]), "drained_decisions": drained,
              "jobs_attempted": int(st.jobs_attempted), "jobs_committed": int(st.jobs_committed)}
    if batch:
        plan_ms, fill_ms, apply_ms = (int(st.reserved[7]) >> 42) / 1e3, ((int(st.reserved[7]) >> 21) & 0x1fffff) / 1e3, (int(st.reserved[7]) & 0x1fffff) / 1e3
        rounds = int(st.reserved[4]); fill_dec = decisions - drained
        buckets = bool((int(st.reserved[1]) >> 62) & 1)  # the fill ran on k_fill_buckets (kai_fill_buckets.hpp): sets of nodes by free devices, all in LDS
        counts = bool((int(st.reserved[1]) >> 61) & 1)   # ... as two wavefronts side by side (kai_fill_counts.hpp): the planned order over the levels' populations, the sets behind a command ring
        levels = bool((int(st.reserved[1]) >> 60) & 1)   # ... with a wavefront per level behind a counting machine that only decides, and a bookkeeper for the dead gangs (kai_fill_levels.hpp)
        fill_kernel = "k_fill_levels" if levels else "k_fill_counts" if counts else "k_fill_buckets" if buckets else "k_fill"
        if sharded:
            engine["exchanges_per_step"] = int(st.reserved[0])
        engine.update({"path": "batch (plan / fill / apply rounds)", "rounds": rounds, "mispredicted_jobs": int(st.reserved[6]), "fill_wave_cycles": int(st.reserved[5]),
                       "fill_kernel": fill_kernel, "round_loop": "on the device (RoundCtl / k_round_next: the host enqueues ahead and never drains the stream between rounds)" if (int(st.reserved[1]) >> 59) & 1 else "on the host (one drain of the stream per round)", "fill_block_loads": int(st.reserved[1]) & ((1 << 48) - 1), "plan_ms": plan_ms, "fill_ms": fill_ms, "apply_ms": apply_ms,
                       "fill_cycles_per_decision": int(st.reserved[5]) / max(fill_dec, 1)})
        # the dominant kernel: k_fill, `rounds` launches per step, timed with HIP events on its stream around every launch (kai_core.hip DevLauncher)
        alg_bytes_launch = fill_dec * b_dec / max(rounds, 1); avg_launch_ms = fill_ms / max(rounds, 1)
        achieved = alg_bytes_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        traffic = pmc_traffic(desc, fill_kernel)
        cyc_dec = int(st.reserved[5]) / max(fill_dec, 1)
        if levels:
            limiter = ("instruction issue of ONE wavefront, the counting machine of k_fill_levels: it walks the planned order over the levels' populations (a handful of integers in registers) and does nothing but "
                       "decide and emit 8-byte commands; a wavefront per level executes them on the LDS-resident sets, a bookkeeper wavefront books the dead gangs' decisions; a conditional branch costs a lone "
                       "wavefront ~20 cycles, a scalar instruction ~5 (tools/micro/issue_rate.hip) — not HBM and not LDS")
            bound_actual = "single-wave issue (the counting machine; the level workers and the bookkeeper run beside it)"
        elif counts:
            limiter = ("instruction issue of ONE wavefront, the counting machine of k_fill_counts: it walks the planned order over the levels' populations (a handful of integers in registers), ~1 000 cycles "
                       "per gang of dependent, mostly scalar instructions (profiles/r05j: section clocks); two more wavefronts execute its commands on the LDS-resident sets and write the tasks' nodes and "
                       "keep up with it (10 - 35 % idle) — not HBM and not LDS")
            bound_actual = "single-wave issue (the counting machine; the set workers run beside it)"
        elif buckets:
            limiter = ("instruction issue of ONE wavefront (k_fill_buckets: 1 workgroup, wavefront 0 walks the planned order over bitmaps of the nodes by free devices: ~134 instructions per decision, "
                       "most of them scalar, 1.4 LDS instructions — profiles/r04q_fill_pmc_instruction_mix.txt), not HBM and not LDS")
            bound_actual = "single-wave issue"
        else:
            limiter = "instruction issue of ONE wavefront (k_fill runs as 1 workgroup x 64 lanes on one of the 256 CUs; fill_cycles_per_decision below), not HBM"
            bound_actual = "single-wave issue"
        note = (f"{fill_kernel} answers a decision from sets of nodes by free devices / a class index kept on chip: it moves five orders of magnitude less than the streaming formulation of SURVEY 8d "
                "(`traffic` = FETCH_SIZE + WRITE_SIZE per launch) and is bound by how fast its wavefronts issue dependent instructions.  achieved / peak: instructions per second over its working "
                "wavefronts against one instruction per 4 cycles and wavefront; `own_roofline` has the cycles per decision, `algorithmic_vs_streaming` the decision throughput against HBM.")
        # the kernel's OWN roofline: one wavefront issues one instruction per issue slot at best; r04q measured 7.7 cycles per instruction for this kind of dependent scalar / vector mix.  Floor taken
        # here: 4 cycles per instruction (a wave64 VALU instruction occupies its SIMD for 4 cycles; dependent SALU instructions are no faster in practice) x the instructions per decision on file.
        ipd = {"k_fill_levels": 170.3, "k_fill_counts": 147.1, "k_fill_buckets": 133.7, "k_fill": 530.0}.get(fill_kernel)  # profiles/r06c_fill_pmc_instruction_mix.txt (all ten wavefronts), r05z_fill_pmc_instruction_mix.txt (all three), r04q_fill_pmc_instruction_mix.txt, DESIGN.md section 5.2
        chains = 3 if counts else 1  # wavefronts that carry the kernel's dependency chains side by side (k_fill_counts: the counting machine + two set workers)
        if levels: chains = 10     # (k_fill_levels: the counting machine, a worker per level, the bookkeeper)
        own = {"bound": "single-wave instruction issue", "cycles_per_decision": cyc_dec, "clock_GHz": 2.4, "wavefronts_working": chains,
               "instructions_per_decision": ipd, "issue_floor_cycles_per_instruction": 4.0,
               "frac_of_issue_floor": (ipd / chains * 4.0 / cyc_dec) if (ipd and cyc_dec > 0) else None,
               "note": "cycles_per_decision = the fill wavefront's clock / decisions; instructions per decision (all working wavefronts together) from the committed SQ_INSTS_* passes of the same workload; "
                       "floor = a wavefront issues at most one instruction per 4 cycles (a wave64 VALU instruction occupies its SIMD for 4 cycles); measured: one per 7.7 - 8.5 cycles (dependent scalar <-> vector chains)"}
        # `roofline` names the resource the kernel saturates: instruction ISSUE of the wavefronts that carry its dependency chains (one instruction per 4 cycles and wavefront at best).
        # achieved = instructions the kernel issues per second (instructions per decision from the committed SQ_INSTS_* passes x the decisions of this run / the kernel's time, HIP events),
        # peak = working wavefronts x clock / 4.  The figure SURVEY 8d prescribes — decisions x (N x 128 B + 80 B) against HBM — is kept beside it as `algorithmic_vs_streaming`.
        issue_peak = chains * 2.4 / 4.0  # G instructions / s
        issue_ach = (ipd * fill_dec / (fill_ms * 1e-3) / 1e9) if (ipd and fill_ms > 0) else 0.0
        roof = {"bound": "issue", "achieved": issue_ach, "peak": issue_peak, "unit": "Ginstr/s", "frac": issue_ach / issue_peak if issue_peak else None, "traffic": traffic,
                "bound_actual": bound_actual, "limiter": limiter, "own_roofline": own,
                "algorithmic_vs_streaming": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                             "note": "SURVEY 8d's figure: decision throughput against a formulation that re-reads every node per decision; this kernel does not stream the nodes, so the fraction exceeds 1 and says nothing about HBM use"},
                "achieved_physical_GBs": (traffic / (avg_launch_ms * 1e-3) / 1e9) if (traffic and avg_launch_ms > 0) else None, "waves_resident": 10 if levels else 4 if buckets else 1, "waves_working": 10 if levels else 3 if counts else 1,
                "traffic_source": "static: profiles/pmc_traffic.json = FETCH_SIZE + WRITE_SIZE of the fill kernel from committed rocprofv3 --pmc passes of this command (counters need their own passes; not collected in this run)" if traffic else "no --pmc pass of this kernel on file",
                "kernel": fill_kernel, "launches_per_step": rounds, "avg_launch_ms": avg_launch_ms, "decisions_per_launch": fill_dec / max(rounds, 1), "algorithmic_bytes_per_launch": alg_bytes_launch,
                "other_kernels": {"plan (k_plan_setup / leaf / rank / gather / scan or k_seg_* / emit)": {"ms_per_step": plan_ms}, "apply (k_apply_jobs / nodes)": {"ms_per_step": apply_ms},
                                  "k_drain": {"decisions": drained, "note": "jobs popped after no class fits anywhere: resolved chip-wide without touching a node; NOT counted in this roofline"}},
                "note": note}
    else:
        if any(a != "allocate" for a in actions):
            s_a = last_stats[0]
            engine["victim_search"] = {"workgroups": int(s_a.reserved[1]), "ranks": world if (sharded and (host_ag or rccl)) else 1, "collectives": int(s_a.reserved[7]) if (sharded and (host_ag or rccl)) else 0, "waves": int(s_a.reserved[5]), "simulations_run": int(s_a.reserved[6]) >> 32, "simulations_counted": int(s_a.reserved[6]) & 0xffffffff,
                                       "scenarios": int(s_a.reserved[2]), "simulations": int(s_a.reserved[3]), "note": "last victim action of the cycle; one replica of the session arrays per workgroup, simulations of a partial job handed out in waves (DESIGN.md)"}
        engine.update({"path": "sequential engine", "control_cycles": {"allocate": int(st.reserved[5]), "commit_discard": int(st.reserved[6]), "total": int(st.reserved[7])}})
        alg = (decisions - drained) * b_dec; achieved = alg / (k_ms * 1e-3) / 1e9
        traffic = pmc_traffic(desc, "k_action")
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_action",
                "launches_per_step": 1, "avg_launch_ms": k_ms, "algorithmic_bytes_per_launch": alg, "note": "decisions of k_action x (N x 128 B + 80 B) / action time; latency bound (one control lane)"}
        if int(st.node_scans) > 0:  # decisions answered by a pass over the nodes (no class index: shared GPUs, node sets of the topology DFS): the scanner of SURVEY 8d as it runs
            roof["scanner"] = {"scans": int(st.node_scans), "nodes_scanned": int(st.nodes_scanned), "nodes_per_s": int(st.nodes_scanned) / (k_ms * 1e-3),
                               "workgroups": max(1, int(st.reserved[1]) >> 48), "lanes_per_workgroup": "448 (engine's) / 512 (helpers)", "achieved_physical_GBs": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                               "note": "the engine's workgroup (7 scan wavefronts beside the control lane) plus the helper workgroups of the scan grid, one CU each (kai_kernels.hpp ScanGrid); physical rate = PMC FETCH_SIZE + WRITE_SIZE of k_action / its time when a pass is on file"}
    out = {
        "metric": "pod placements/sec (" + " + ".join(actions) + (" action" if len(actions) == 1 else " actions") + ", synthetic snapshot)", "value": value, "unit": "placements/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if (sharded or (world == 1 and os.environ.get("KAI_BENCH_MULTI", "replicas") == "shard")) else "weak",  # (N = 1 carries the label of the N > 1 mode it is the base of: replicas by default)
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "placements_per_s": total_placed / elapsed,
        # every allocateTask execution is a decision; the ones k_drain resolves — jobs popped once no class fits anywhere, turned away without touching a node — are split out
        "decisions_per_s": {"all": total_decisions / elapsed, "fill": (total_decisions - drained * args.steps * (1 if sharded else world)) / elapsed if batch else None,
                            "drained": (drained * args.steps * (1 if sharded else world)) / elapsed if batch else None},
        "config": {"workload": desc, "nodes": N, "pods": snap.n_pods, "jobs": snap.n_jobs, "queues": snap.n_queues,
                   "decisions_per_step": decisions, "placements_per_step": placed, "p50_cycle_latency_ms": lat[len(lat) // 2], "step_wall_ms": {"p50": sorted(step_wall)[len(step_wall) // 2], "min": min(step_wall), "max": max(step_wall), "note": "host clock around each timed step (ms_per_step is their mean): a max far above the p50 is a host hiccup inside the timed region, not device time"}, "action_ms": k_ms,
                   "session_open_ms": float(np.mean(open_ms)), "parallelism": "1 GPU" if world == 1 else (f"node axis of one snapshot sharded over {world} GPUs: per exchange every rank offers its 128 best nodes per scan class, all-gather over RCCL / xGMI, the same virtual fill on every rank"
                                                                      if sharded else f"{world} scheduling shards, one per GPU, no data-path collective"),
                   "snapshot_gen_s": round(gen_s, 2), "host_to_hbm_s": round(upload_s, 3), "engine": engine},
        "roofline": roof,
    }

    if rank == 0:
        # end-to-end pin: SHA-256 of this run's committed operations against the oracle's full-size run of the same workload (tools/pin_full_sizes.py wrote
        # profiles/full_size_pins.json on the CPU: oracle vs host-compiled engine, every operation / pod / node / share equal)
        import kai_testlib as T
        sha = T.ops_sha256(first_ops); pin = None
        try:
            with open(os.path.join(ROOT, "profiles", "full_size_pins.json")) as f:
                pin = next((v for v in json.load(f).values() if v["workload"] == desc and v["nodes"] == N and v["pods"] == snap.n_pods and v.get("actions", list(actions)) == list(actions)), None)
        except (OSError, ValueError, KeyError):
            pin = None
        out["parity_full"] = {"ops": len(first_ops), "ops_sha256": sha, "oracle_ops_sha256": pin["ops_sha256"] if pin else None, "equal_to_oracle": (sha == pin["ops_sha256"]) if pin else None,
                              "oracle_pin": "profiles/full_size_pins.json (oracle end to end on the CPU, %s s)" % pin["oracle_s"] if pin else "no pin on file for this workload"}
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        import kai_testlib as T
        # (1) contract baseline: the oracle (faithful single-thread restatement of the reference path) on a bounded sample of the SAME snapshot:
        #     it walks the same fair order and stops after `sample` decisions; its operations must be the first operations of the GPU's
        sample = args.cpu_sample if args.cpu_sample > 0 else max(200, int(3e8 / max(N, 1)))  # about 10-20 s of single-thread oracle work
        c2 = T.abi.KaiConfig.from_buffer_copy(cfg)
        c2.reserved[0] = sample  # bounds the allocate action only; the victim actions of a C4 run are timed in full (keep --scale small)
        ref = T.Oracle.run(snap, c2, actions)
        done = int(ref.stats.decisions)
        one = done / (ref.elapsed_ms * 1e-3)
        out["cpu_baseline"] = {"value": one, "unit": "decisions/s", "cores": 1, "kind": "port",
                               "sample": f"{'first ' if actions == ('allocate',) else ''}{done} decisions of the same snapshot in {ref.elapsed_ms / 1e3:.1f} s (oracle/liboracle.so, single thread, incl. session open; actions: {', '.join(actions)})"}
        # the reference scores the nodes of one decision on goroutines (framework/session.go:243-261): the same fan-out on 8 threads; the
        # better of the two is the baseline, the other is kept beside it
        nthr = min(8, os.cpu_count() or 1)
        if nthr > 1 and N >= 2048:
            ref8 = T.Oracle.run(snap, c2, actions, threads=nthr)
            v8 = int(ref8.stats.decisions) / (ref8.elapsed_ms * 1e-3)
            assert ref8.ops == ref.ops, "oracle: threaded node scoring changed the operations"
            out["cpu_baseline"]["one_thread"] = one
            out["cpu_baseline"]["threads_%d" % nthr] = v8
            if v8 > one:
                out["cpu_baseline"].update({"value": v8, "cores": nthr, "sample": out["cpu_baseline"]["sample"].replace("single thread", f"node scoring of each decision on {nthr} threads; {ref8.elapsed_ms / 1e3:.1f} s, single thread: {ref.elapsed_ms / 1e3:.1f} s")})
        if actions == ("allocate",):
            n_eq = 0
            for a, b in zip(ref.ops, first_ops):
                if tuple(a) != tuple(b):
                    break
                n_eq += 1
            out["parity_prefix"] = {"oracle_ops": len(ref.ops), "equal_to_gpu": n_eq}
        # (2) one CPU thread on the full step, two ways (both test infrastructure, tests/host_sim):
        #   cpu_sequential_engine : the host-compiled SEQUENTIAL engine (engine_mode 3: the class-index walk of round 1, literal heaps) — another algorithm than the batch path's;
        #   cpu_same_algorithm    : the batch path's FILL as it is — sets of nodes by free devices, whole nodes per step — in plain scalar C++ (native_bucket_fill.hpp), run as a shadow of
        #                           every emulated fill launch with its outputs compared.  Live here on a scaled copy of the workload (the emulated plan kernels make a full-size run a matter
        #                           of minutes); the full-size figure is the committed profiles/r05_native_fill_c5.json (tools/native_fill_timing.py on a GPU box's host).
        try:
            from test_engine_hostsim import HostSim
            c3 = T.abi.KaiConfig.from_buffer_copy(cfg); c3.engine_mode = 3
            tw = HostSim.run(snap, c3, actions)
            out["cpu_sequential_engine"] = {"value": int(tw.stats.decisions) / (tw.elapsed_ms * 1e-3), "unit": "decisions/s", "cores": 1, "kind": "host-compiled sequential engine (tests/host_sim, g++ -O2): NOT the batch path's algorithm",
                                            "ms_per_step": tw.elapsed_ms, "sample": "the full step", "ops_equal_to_gpu": [tuple(o) for o in tw.ops] == [tuple(o) for o in first_ops]}
        except Exception as e:  # the twin is optional evidence
            out["cpu_sequential_engine"] = {"error": str(e)[:200]}
        if batch and actions == ("allocate",) and not args.mixed and args.fractions == 0 and os.environ.get("KAI_BENCH_NATIVE_FILL", "1") != "0":
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import native_fill_timing as NF
                live_scale = min(args.scale, 0.1 if args.config == "C5" else 1.0)
                live = NF.measure(idx, live_scale)
                gpu_ns = fill_ms * 1e6 / max(fill_dec, 1)
                same = {"kind": "the batch path's fill (sets of nodes by free devices) as scalar C++ on ONE host core, tests/host_sim/native_bucket_fill.hpp; outputs compared with the emulated kernel's at every launch",
                        "cores": 1, "live": live, "gpu_fill_ns_per_decision": gpu_ns, "gpu_fill_kernel": fill_kernel,
                        "host_core_over_gpu_fill": gpu_ns / live["ns_per_decision"] if live["ns_per_decision"] > 0 else None,
                        "note": "fill only: the plan (data-parallel segmented scans / merges over the queue tree) and apply kernels have no single-core twin that is not the emulator; "
                                "a ratio above 1 says one host core walks this chain faster than the MI355X's fill wavefronts"}
                try:
                    with open(os.path.join(ROOT, "profiles", "r05_native_fill_c5.json")) as f:
                        same["full_size_on_file"] = json.load(f)
                except (OSError, ValueError):
                    pass
                out["cpu_same_algorithm"] = same
            except Exception as e:
                out["cpu_same_algorithm"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and os.environ.get("KAI_BENCH_OPEN_LEG", "1" if elapsed / args.steps < 5.0 else "0") != "0":  # (not for cycles of many seconds: the leg repeats the cycle several times)
        # What a production scheduler pays per cycle: every cycle opens a session on a NEW snapshot (scheduler.go:112-138, framework/framework.go:32-65), so
        # kai_session_open — host preparation (node permutation, task / job orders, scan classes: on the host's cores), ~80 MB over PCIe, the OnSessionOpen
        # kernels — belongs to the cycle.  `value` above keeps the contract's definition (inputs resident in HBM); these two legs are reported beside it:
        #   serial    : open + actions, one after the other, on one handle;
        #   pipelined : two handles — while cycle k's actions run on one, cycle k+1's snapshot is opened on the other (host thread + its own stream).  A
        #               scheduler can only do this if snapshot k+1 is taken before cycle k's decisions are committed to its cache, so it is an upper bound of
        #               what overlap buys, not a drop-in mode.
        try:
            import threading
            n_cyc = max(3, min(args.steps, 7))
            ser, ser_open = [], []
            core2 = pkg.KaiCore(cfg, gpu_ids=(dev_index,))
            for i in range(n_cyc + 1):
                t0 = time.perf_counter(); s2 = core2.open_session(snap); t1 = time.perf_counter()
                for a in actions:
                    s2.execute(a, copy=False)
                t2 = time.perf_counter(); s2.close()
                if i > 0:
                    ser.append((t2 - t0) * 1e3); ser_open.append((t1 - t0) * 1e3)
            cores = [core2, pkg.KaiCore(cfg, gpu_ids=(dev_index,))]
            pipe = []
            cur = cores[0].open_session(snap)
            for i in range(n_cyc + 1):
                box = {}
                def opener(k=(i + 1) % 2):
                    box["s"] = cores[k].open_session(snap)
                t0 = time.perf_counter(); th = threading.Thread(target=opener); th.start()
                for a in actions:
                    cur.execute(a, copy=False)
                th.join(); t1 = time.perf_counter()
                cur.close(); cur = box["s"]
                if i > 0:
                    pipe.append((t1 - t0) * 1e3)
            cur.close()
            for c2 in cores:
                c2.destroy()
            med = lambda v: sorted(v)[len(v) // 2]
            out["cycle_with_open_ms"] = {"p50": med(ser), "open_p50": med(ser_open), "cycles": n_cyc, "over_ms_per_step": med(ser) / (elapsed / args.steps * 1e3) - 1.0,
                                         "note": "kai_session_open (host prep + PCIe + OnSessionOpen kernels) + the actions, serial, one handle: the cycle a scheduler pays"}
            out["cycle_pipelined_ms"] = {"p50": med(pipe), "cycles": n_cyc, "over_ms_per_step": med(pipe) / (elapsed / args.steps * 1e3) - 1.0,
                                         "note": "two handles: cycle k+1's open overlaps cycle k's actions (valid only if the next snapshot does not wait for this cycle's commits)"}
            out["config"]["p50_cycle_latency_ms_note"] = "reset of the HBM-resident snapshot + actions; with the per-cycle open: cycle_with_open_ms"
        except Exception as e:  # additional evidence
            out["cycle_with_open_ms"] = {"error": str(e)[:200]}
    if sharded and os.environ.get("KAI_BENCH_REPLICAS_LEG", "1") != "0":
        try:
            # second leg, every rank: the same GPUs as independent scheduling shards (how KAI itself scales out: one scheduler instance per node pool,
            # conf/scheduler_conf.go:95-112) — one snapshot of the same shape per rank, no data-path collective, same bracket; reported beside the sharded value
            snap2, cfg2, _ = pkg.synth.config(idx, args.scale, seed_offset=pkg.dist.shard_seed(0, rank))
            core2 = pkg.KaiCore(cfg2, gpu_ids=(dev_index,)); ssn2 = core2.open_session(snap2)

            def step2():
                ssn2.reset(); n = 0
                for a in actions:
                    n += len(ssn2.execute(a))
                return n
            d2 = 0
            for _ in range(args.warmup):
                step2()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                d2 = step2()
            barrier()
            el2 = pkg.dist.max_over_ranks(time.perf_counter() - t0, device=red_dev)
            tot2 = pkg.dist.sum_over_ranks(d2 * args.steps, device=red_dev)
            ssn2.close(); core2.destroy()
            out["replicas"] = {"value": tot2 / el2, "unit": "placements/s", "ms_per_step": el2 / args.steps * 1e3, "scaling": "weak",
                               "note": f"{world} independent scheduling shards of the same shape, one per GPU, no data-path collective (the default for N > 1)"}
        except Exception as e:  # the second leg is additional evidence: it must not take the sharded result down with it (every rank runs the same code, so a failure is common to all)
            out["replicas"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and args.config == "C5" and args.scale == 1.0 and not args.mixed and args.fractions == 0 and actions == ("allocate",) and os.environ.get("KAI_BENCH_OTHER_SHAPES", "1") != "0":
        # beside the headline (plain gangs: every job on the batch path): the same cluster in the shape SURVEY 8d writes down — zone / rack labels, 5 % of the gangs with a
        # required rack, 5 % elastic, minruntime on — and config 3 with 30 % of its one-GPU pods as fractions of a device; both run on the sequential engine with its passes
        # over the nodes on the scan grid.  One warm-up + one timed cycle each; additional evidence, never part of `value`.
        shapes = {}
        # (+ BASELINE configs 2 and 3 as they are — batch path, plan / fill / apply rounds —, so that the driver-run line holds every single-GPU configuration of BASELINE.json)
        for key, (i2, kw, frac) in {"c5_mixed": (4, {"mixed": True}, 0.0), "c3_fractions_30": (2, {}, 0.3), "c3": (2, {}, 0.0), "c2": (1, {}, 0.0)}.items():
            try:
                s2, c2, d2 = pkg.synth.config(i2, 1.0, **kw)
                if frac > 0:
                    pkg.synth.add_fractions(s2, 7, frac=frac)
                core2 = pkg.KaiCore(c2, gpu_ids=(dev_index,)); ssn2 = core2.open_session(s2)
                n2 = 0; ops2 = None
                for it in range(2 if key in ("c5_mixed", "c3_fractions_30") else 4):  # (the last cycle is the one reported)
                    ssn2.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
                    ops2 = ssn2.execute("allocate"); n2 = len(ops2); torch.cuda.synchronize(); el = time.perf_counter() - t0
                st2 = ssn2.stats()
                w2 = d2 + (" + 30 % of the one-GPU pods as fractions of one device" if frac > 0 else "")
                import hashlib
                sha2 = hashlib.sha256(np.stack([ops2["kind"], ops2["pod"], ops2["node"], ops2["job"]], 1).astype("<i4").tobytes()).hexdigest()  # = kai_testlib.ops_sha256
                pin2 = None
                try:
                    with open(os.path.join(ROOT, "profiles", "full_size_pins.json")) as f:
                        pin2 = next((v for v in json.load(f).values() if v.get("workload") == w2 and v.get("nodes") == s2.n_nodes and v.get("pods") == s2.n_pods), None)
                except (OSError, ValueError):
                    pin2 = None
                shapes[key] = {"workload": w2, "ms_per_step": el * 1e3, "placements_per_s": n2 / el, "placements": n2,
                               "ops_sha256": sha2, "equal_to_oracle": (sha2 == pin2["ops_sha256"]) if pin2 else None,
                               "decisions": int(st2.decisions), "path": "batch" if int(st2.reserved[4]) > 0 else "sequential engine", "scan_grid_workgroups": max(1, int(st2.reserved[1]) >> 48) if int(st2.reserved[4]) == 0 else None}
                ssn2.close(); core2.destroy()
            except Exception as e:  # additional evidence: it must not take the headline down
                shapes[key] = {"error": str(e)[:200]}
        # the reference's OWN benchmark shapes (generators restated in tools/ref_benchmarks.py): `kai_session_open` + the benchmark's actions per
        # iteration — what one b.N iteration of the Go benchmark times minus its fixture construction —, the operations' hash against the oracle's pin
        # (profiles/reference_benchmark_pins.json, tools/pin_ref_benchmarks.py).  The victim search's hardware figures for the round; never part of `value`.
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
            import hashlib
            import kai_testlib as T
            import ref_benchmarks as RB
            with open(os.path.join(ROOT, "profiles", "reference_benchmark_pins.json")) as f:
                pins = json.load(f)["pins"]
            rows, t_leg = {}, time.perf_counter()
            for name, build, acts, published in RB.BENCHES:
                if name not in pins or time.perf_counter() - t_leg > 40.0:  # the leg is bounded: a slow box drops the later shapes instead of the line
                    continue
                try:
                    s3, c3, _m = T.case_to_snapshot(build(), acts)
                    times, sha3, n3 = [], None, 0
                    with pkg.KaiCore(c3, gpu_ids=(dev_index,)) as core3:
                        for it in range(4):
                            t0 = time.perf_counter()
                            ssn3 = core3.open_session(s3)
                            parts = [ssn3.execute(a) for a in acts]
                            dt = (time.perf_counter() - t0) * 1e3
                            if it == 0:
                                quad = np.concatenate([np.stack([o["kind"], o["pod"], o["node"], o["job"]], 1).astype("<i4") for o in parts]) if parts else np.zeros((0, 4), "<i4")
                                sha3, n3 = hashlib.sha256(quad.tobytes()).hexdigest(), int(quad.shape[0])
                            ssn3.close()
                            if it:
                                times.append(dt)
                            if dt > 3000 and it >= 1:
                                break
                    times.sort()
                    rows[name] = {"nodes": s3.n_nodes, "pods": s3.n_pods, "actions": list(acts), "open_plus_actions_ms": times[len(times) // 2], "iterations": len(times),
                                  "operations": n3, "ops_sha256": sha3, "equal_to_oracle": sha3 == pins[name]["ops_sha256"],
                                  "reference_published": published, "host_compiled_engine_ms_one_core": pins[name].get("host_compiled_engine_ms")}
                except Exception as e:
                    rows[name] = {"error": str(e)[:200]}
            shapes["reference_benchmarks"] = rows
        except Exception as e:  # additional evidence: it must not take the headline down
            shapes["reference_benchmarks"] = {"error": str(e)[:200]}
        out["other_shapes"] = shapes
    if rank == 0:
        print(json.dumps(out))
    pkg.dist.finish()


if __name__ == "__main__":
    main()
