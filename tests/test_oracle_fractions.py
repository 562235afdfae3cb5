"""Shared GPUs in the oracle beyond the reference's golden tables: invariants of the reference's own accounting
(api/node_info/gpu_sharing_node_info.go) on randomized clusters, full scheduling cycles incl. the victim actions."""
import numpy as np
import pytest

import kai_testlib as T

pkg, abi = T.pkg, T.abi
S = abi.POD_STATUS
HOLDS = S["Allocated"] | S["Binding"] | S["Bound"] | S["Running"] | S["Pipelined"] | S["Releasing"]


def _check(snap, res, label, allocate_only):
    """What must hold whatever the path taken: no shared GPU is handed out beyond one device to pods that keep running on it; after an allocate
    action alone (nothing releasing, nothing pipelined) every GPU of a node is held by whole-GPU pods, is a shared GPU in use, or is idle."""
    a = snap.arrays
    N = snap.n_nodes
    por = a["pod_gpu_portion"]
    whole = np.zeros(N); groups = [dict() for _ in range(N)]
    keeps = S["Allocated"] | S["Binding"] | S["Bound"] | S["Running"]
    for p in range(snap.n_pods):
        st, n = int(res.pod_status[p]), int(res.pod_node[p])
        if not (st & keeps) or n < 0:
            continue
        if por[p] > 0:
            g = int(res.gpu_groups[p])
            assert g >= 0, f"{label}: pod {p} holds a fraction without a GPU group"
            groups[n][g] = groups[n].get(g, 0) + int(por[p] * 100)
        else:
            whole[n] += a["pod_req"][abi.RES_GPU, p]
    for n in range(N):
        for g, used in groups[n].items():
            assert used <= 100, f"{label}: node {n} GPU group {g} holds {used} % of a device"  # enoughResourcesOnGpu: memory of one device
        if allocate_only:
            total = a["node_allocatable"][abi.RES_GPU, n]
            assert res.nodes["idle"][n, abi.RES_GPU] == total - whole[n] - len(groups[n]), f"{label}: node {n} idle {res.nodes['idle'][n, abi.RES_GPU]}"
            assert res.nodes["releasing"][n, abi.RES_GPU] == 0


@pytest.mark.parametrize("seed", range(40))
def test_fraction_cycles_keep_the_accounting_invariants(seed):
    snap = pkg.synth.make_crowded_snapshot(3 + seed % 7, 9100 + seed, fill=0.6 + 0.3 * (seed % 4) / 3, n_pending_jobs=6 + seed % 9, elastic_frac=0.2,
                                           hog_frac=0.5, queue_levels=((2, 2), (3,))[seed % 2])
    pkg.synth.add_fractions(snap, seed, frac=0.5)
    if not (snap.arrays["pod_gpu_portion"] > 0).any():
        pytest.skip("no fraction pod drawn")
    cfg = abi.default_config(max_consolidation_preemptees=-1, gpu_strategy=(abi.BINPACK, abi.SPREAD)[seed % 2])
    if seed % 3 == 0: cfg.plugins = (cfg.plugins & ~abi.PLUGINS["gpupack"]) | abi.PLUGINS["gpuspread"]
    cfg.use_scheduling_signatures = 0
    for acts in (("allocate",), ("allocate", "consolidation", "reclaim", "preempt")):
        res = T.Oracle.run(snap, cfg, acts)
        _check(snap, res, f"seed {seed} {acts}", acts == ("allocate",))
        again = T.Oracle.run(snap, cfg, acts)  # the restatement is deterministic (the reference's map orders are fixed canonically)
        assert again.ops == res.ops and np.array_equal(again.gpu_groups < (1 << 20), res.gpu_groups < (1 << 20))


def test_engine_twin_takes_fractions():
    """The host-compiled engine carries the shared-GPU code (KAI_SHARED_GPUS) for every action and matches the oracle (test_engine_hostsim.py holds
    the goldens and the fuzz).  libkai_core itself is built without the flag and refuses such snapshots outright."""
    import test_engine_hostsim as H
    snap = pkg.synth.make_crowded_snapshot(4, 9000)
    pkg.synth.add_fractions(snap, 1, frac=1.0, portions=(0.25, 0.5, 0.75))
    cfg = abi.default_config()
    for acts in (("allocate",), ("allocate", "reclaim"), ("allocate", "consolidation", "reclaim", "preempt")):
        ref, res = T.Oracle.run(snap, cfg, acts), H.HostSim.run(snap, cfg, acts)
        assert res.ops == ref.ops and np.array_equal(res.pod_status, ref.pod_status) and np.array_equal(res.pod_node, ref.pod_node)


# ------------------------------------------------------------------------------------------------ gpupack / gpuspread / GetNodePreferableGpuForSharing (tools/go_kat_gpu_sharing.py)
import ctypes as C, json, os  # noqa: E402
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_gpu_sharing.json")) as _fh:
    GS = json.load(_fh)


@pytest.mark.parametrize("case", GS["gpu_order"], ids=[f"{c['plugin']}:{c['line']}" for c in GS["gpu_order"]])
def test_device_group_order_plugins(case):
    """plugins/gpupack/gpupack.go:31-45 and plugins/gpuspread/gpuspread.go:31-46 against their six cases each: the score of a device group (or of a whole free GPU)
    from the memory in use on it; a node whose GPU memory is below DefaultGpuMemory is an error, score 0"""
    lib = T.Oracle.lib(); lib.kai_oracle_gpu_order_kat.restype = C.c_double
    lib.kai_oracle_gpu_order_kat.argtypes = [C.c_uint32, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int)]
    err = C.c_int(0)
    score = lib.kai_oracle_gpu_order_kat(abi.PLUGINS[case["plugin"]], case["total_mem"], case["used_mem"], int(case["whole_gpu"]), C.byref(err))
    assert bool(err.value) == case["want_error"] and score == case["want_score"], (score, err.value)


@pytest.mark.parametrize("case", GS["preferable_gpu_for_sharing"], ids=[f"{c['line']}" for c in GS["preferable_gpu_for_sharing"]])
def test_node_preferable_gpu_for_sharing(case):
    """gpu_sharing/gpuSharing.go:39-83 against Test_getNodePreferableGpuForSharing: how many groups the pod takes out of the fitting GPUs, which numbered groups are among
    them, whether the placement waits for something releasing (a numbered group the node holds no allocation for counts as pipelined; the last case asks for two devices)"""
    al = case["node_allocatable"]; pod = case["pod"]
    g = float(pod["gpu_fraction"]) if pod["gpu_fraction"] else float(pod["gpu_request"])
    scene = {"Name": case["name"], "Nodes": {"n1": {"CPUMillis": int(al["cpu"]) * 1000, "GPUs": int(al["nvidia.com/gpu"]), "MaxTaskNum": int(al["pods"])}}, "Queues": [{"Name": "q", "DeservedGPUs": 1}],
             "Jobs": [{"Name": "pg1", "Priority": 50, "QueueName": "q", "RequiredGPUsPerTask": g, "Tasks": [{"State": "Pending"}]}], "JobExpectedResults": {}}
    snap, cfg, _ = T.case_to_snapshot(scene, fractions=True)
    fitting = [-1 if x == "whole" else int(x) for x in case["fitting"]]
    lib = T.Oracle.lib(); lib.kai_oracle_gpu_sharing_kat.restype = C.c_int
    out = (C.c_int32 * 8)(); rel = C.c_int(0); s = snap.as_struct()
    n = lib.kai_oracle_gpu_sharing_kat(C.byref(cfg), C.byref(s), 0, 0, int(pod["num_devices"] or 0), (C.c_int32 * len(fitting))(*fitting), len(fitting), int(case["pipeline_only"]), out, 8, C.byref(rel))
    w = case["want"]
    assert n == w["groups"] and bool(rel.value) == w["releasing"], (n, rel.value)
    assert all(int(x) in list(out[:n]) for x in w["includes"]), list(out[:n])
