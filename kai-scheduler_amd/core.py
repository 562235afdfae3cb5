"""Host-side mirror of the reference's Session / Action interface over the C ABI (libkai_core.so).

    core = KaiCore(cfg)                       # scheduler.NewScheduler  (pkg/scheduler/scheduler.go:53-101)
    ssn  = core.open_session(snapshot)        # framework.OpenSession   (framework/framework.go:32-65)
    ops  = ssn.execute("allocate")            # Action.Execute(ssn)     (actions/allocate/allocate.go:46-77)
    ssn.queue_shares(); ssn.pod_states()      # what the Go shim mirrors back into the Session
    ssn.close()                               # framework.CloseSession

The library is the only implementation: if it (or a HIP device) is missing this module raises — there is
no Python or CPU fallback of the path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KAI_CORE_LIB") or os.path.join(_HERE, "csrc", "libkai_core.so")  # KAI_CORE_LIB: another BUILD of the same HIP library (profiling variants)

EXPORTS = ["kai_core_create", "kai_core_destroy", "kai_session_open", "kai_queue_shares", "kai_action_execute", "kai_best_node",
           "kai_pod_states", "kai_node_states", "kai_pod_gpu_groups", "kai_shard_attach", "kai_shard_attach_host", "kai_shard_rccl_id", "kai_shard_attach_rccl", "kai_shard_allgather_probe", "kai_action_stats_get", "kai_session_reset", "kai_session_close", "kai_last_error", "kai_version"]


_OP_DTYPE = np.dtype([("seq", "<i8"), ("kind", "<i4"), ("pod", "<i4"), ("node", "<i4"), ("job", "<i4"), ("stmt", "<i4"), ("pad", "<i4")])  # kai_op (include/kai_core.h)


class KaiError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__(f"kai_core status {code} ({abi.STATUS_TEXT.get(code, '?')}): {detail}")


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)  # kai_allgather_fn (include/kai_core.h)
_lib = None
_hip = None


def _hip_runtime():
    """libamdhip64 through ctypes: device-to-device copies between the library's exchange buffers and torch's tensors."""
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
        _hip.hipDeviceSynchronize.argtypes = []
        _hip.hipDeviceSynchronize.restype = C.c_int
    return _hip


def load_library(path: str = LIB_PATH):
    """dlopen libkai_core.so and declare prototypes.  Raises if the HIP extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing — build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                                "the scheduling-cycle core has no non-HIP implementation")
    lib = C.CDLL(path)
    lib.kai_version.restype = C.c_char_p
    lib.kai_last_error.restype = C.c_char_p
    lib.kai_last_error.argtypes = [C.c_void_p]
    lib.kai_core_create.argtypes = [C.POINTER(abi.KaiConfig), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    lib.kai_core_destroy.argtypes = [C.c_void_p]
    lib.kai_session_open.argtypes = [C.c_void_p, C.POINTER(abi.KaiSnapshotSoA)]
    lib.kai_session_close.argtypes = [C.c_void_p]
    lib.kai_shard_attach.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p]
    lib.kai_shard_attach_host.argtypes = [C.c_void_p, ALLGATHER_FN, C.c_void_p]
    lib.kai_shard_rccl_id.argtypes = [C.c_void_p, C.c_void_p]
    lib.kai_shard_attach_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.kai_shard_allgather_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.kai_pod_gpu_groups.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    lib.kai_session_reset.argtypes = [C.c_void_p]
    lib.kai_queue_shares.argtypes = [C.c_void_p, C.POINTER(abi.KaiQueueShare), C.c_int]
    lib.kai_action_execute.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.KaiOp), C.c_int64, C.POINTER(C.c_int64)]
    lib.kai_best_node.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int)]
    lib.kai_pod_states.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]
    lib.kai_node_states.argtypes = [C.c_void_p, C.POINTER(abi.KaiNodeState), C.c_int]
    lib.kai_action_stats_get.argtypes = [C.c_void_p, C.POINTER(abi.KaiActionStats)]
    for name in EXPORTS:
        if name not in ("kai_version", "kai_last_error"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


class KaiCore:
    """One handle = one GPU.  `world` > 1: this process is rank `rank` of a group that shards the NODE axis of one session over the GPUs of one
    node (one process per GPU, every rank opens the same snapshot and makes the same calls).  The group's exchange step is an
    all-gather of a few KB per rank: `allgather(send_ptr, recv_ptr, nbytes_per_rank)` on the library's device buffers, by default
    torch.distributed.all_gather_into_tensor on the default process group (backend "nccl" = RCCL over xGMI on the GPU box)."""

    def __init__(self, cfg: abi.KaiConfig | None = None, gpu_ids=(0,), world: int = 1, rank: int = 0, offers_per_class: int = 0, allgather=None, host_allgather=None):
        self.lib = load_library()
        self.cfg = cfg or abi.default_config()
        self.handle = C.c_void_p()
        self.world, self.rank = int(world), int(rank)
        ids = (C.c_int * 1)(gpu_ids[0])
        rc = self.lib.kai_core_create(C.byref(self.cfg), self.world, ids, C.byref(self.handle))
        if rc != 0:
            raise KaiError(rc, "kai_core_create")
        if isinstance(allgather, str) and allgather == "rccl":
            # the library's own communicator (kai_shard_attach_rccl): rank 0 draws the id, the default process group carries it to the others, the exchange itself
            # is an ncclAllGather the library issues on its stream between its kernels
            self._attach_rccl(int(offers_per_class))
        elif self.world > 1:
            self._user_allgather = allgather
            self._stage = None
            self._cb = ALLGATHER_FN(self._allgather)  # kept alive with the handle
            rc = self.lib.kai_shard_attach(self.handle, self.rank, self.world, int(offers_per_class), self._cb, None)
            if rc != 0:
                raise KaiError(rc, "kai_shard_attach")

        if self.world > 1 and host_allgather is not None:
            # the victim actions' waves over the ranks (kai_shard_attach_host): host_allgather(send_ptr, recv_ptr, nbytes_per_rank) on HOST memory, called while the
            # action's kernel runs (it must not synchronise the device); True = torch.distributed on CPU tensors of a gloo group `host_group` (default group if gloo)
            self._host_ag = host_allgather
            self._hcb = ALLGATHER_FN(self._host_allgather)
            rc = self.lib.kai_shard_attach_host(self.handle, self._hcb, None)
            if rc != 0:
                raise KaiError(rc, "kai_shard_attach_host")

    host_group = None  # the process group host_allgather=True uses (a gloo group beside an nccl default group)

    def _host_allgather(self, user, send, recv, nbytes):
        try:
            if callable(self._host_ag):
                return int(self._host_ag(send, recv, nbytes) or 0)
            import numpy as np
            import torch
            import torch.distributed as dist
            s = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,)))
            r = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * self.world,)))
            dist.all_gather_into_tensor(r, s, group=self.host_group)
            return 0
        except Exception:  # a ctypes callback must not raise
            import traceback
            traceback.print_exc()
            return 1

    def _attach_rccl(self, offers_per_class):
        idbuf = (C.c_ubyte * 128)()
        if self.rank == 0:
            rc = self.lib.kai_shard_rccl_id(self.handle, idbuf)
            if rc != 0:
                raise KaiError(rc, self.lib.kai_last_error(self.handle).decode())
        if self.world > 1:
            import torch
            import torch.distributed as dist
            t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(t, 0)
            idbuf = (C.c_ubyte * 128)(*t.cpu().tolist())
        rc = self.lib.kai_shard_attach_rccl(self.handle, self.rank, self.world, offers_per_class, idbuf)
        if rc != 0:
            raise KaiError(rc, self.lib.kai_last_error(self.handle).decode())

    def _allgather(self, user, send, recv, nbytes):
        try:
            if self._user_allgather is not None:
                return int(self._user_allgather(send, recv, nbytes) or 0)
            import torch
            import torch.distributed as dist
            hip = _hip_runtime()
            on_host = dist.get_backend() != "nccl"  # a gloo group (rehearsal of the multi-process path without RCCL): staged through host memory
            if self._stage is None or self._stage[0].numel() != nbytes:  # torch-owned staging tensors: the collective runs on memory torch knows
                dev = "cpu" if on_host else "cuda"
                self._stage = (torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.empty(nbytes * self.world, dtype=torch.uint8, device=dev))
            s, r = self._stage
            if hip.hipMemcpy(C.c_void_p(s.data_ptr()), C.c_void_p(send), C.c_size_t(nbytes), 2 if on_host else 3) != 0:  # hipMemcpyDeviceToHost / DeviceToDevice
                return 1
            dist.all_gather_into_tensor(r, s)
            if not on_host: torch.cuda.synchronize()
            if hip.hipMemcpy(C.c_void_p(recv), C.c_void_p(r.data_ptr()), C.c_size_t(nbytes * self.world), 1 if on_host else 3) != 0:
                return 1
            # the library goes on with kernels on its own non-blocking stream, which has no implicit order with the null stream this copy ran on
            return 0 if hip.hipDeviceSynchronize() == 0 else 1
        except Exception:  # a ctypes callback must not raise
            import traceback
            traceback.print_exc()
            return 1

    def _check(self, rc):
        if rc != 0:
            raise KaiError(rc, self.lib.kai_last_error(self.handle).decode())

    def open_session(self, snap: abi.Snapshot) -> "Session":
        s = snap.as_struct()
        self._check(self.lib.kai_session_open(self.handle, C.byref(s)))
        return Session(self, snap)

    def destroy(self):
        if self.handle:
            self.lib.kai_core_destroy(self.handle)
            self.handle = C.c_void_p()

    def _ops_buffer(self, cap: int):
        """The caller-owned `ops_out` array of kai_action_execute: one per handle, kept across sessions (untouched pages cost nothing; a fresh 68 MB array per session of config 5 would)."""
        buf = getattr(self, "_ops_buf", None)
        if buf is None or len(buf) < cap:
            buf = self._ops_buf = np.empty(cap, _OP_DTYPE)
        return buf

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.destroy()


class Session:
    """An open scheduling session resident in HBM."""

    def __init__(self, core: KaiCore, snap: abi.Snapshot):
        self.core, self.snap = core, snap

    def execute(self, action: str | int, copy: bool = True):
        """Run one Action; returns the committed operations [(kind, pod, node, job)] in commit order.

        `copy=False` returns a view of the handle's output buffer (the caller-owned `ops_out` of the C ABI): valid until the next `execute` on this handle."""
        lib, h = self.core.lib, self.core.handle
        act = abi.ACTIONS[action] if isinstance(action, str) else int(action)
        cap = 2 * self.snap.n_pods + 64
        ops = self.core._ops_buffer(cap)
        n = C.c_int64(0)
        self.core._check(lib.kai_action_execute(h, act, ops.ctypes.data_as(C.POINTER(abi.KaiOp)), cap, C.byref(n)))
        out = ops[:n.value]
        if not copy:
            return out
        # (a flat byte copy: numpy copies a structured array record by record, 3.5 ms for config 5's 150 k operations)
        return out.view(np.uint8).copy().view(_OP_DTYPE)

    def best_node(self, pod: int, pipeline_only: bool = False, nodeset=None):
        """Session.OrderedNodesByTask + FittingNode for one task; `nodeset` = iterable of node indices (None = all nodes)."""
        node, pipe = C.c_int32(-1), C.c_int(0)
        bits = None
        if nodeset is not None:
            words = np.zeros(max((self.snap.n_nodes + 31) // 32, 1), np.uint32)
            for n in nodeset:
                words[n >> 5] |= np.uint32(1 << (n & 31))
            bits = words.ctypes.data_as(C.POINTER(C.c_uint32))
        self.core._check(self.core.lib.kai_best_node(self.core.handle, pod, bits, int(pipeline_only), C.byref(node), C.byref(pipe)))
        return node.value, bool(pipe.value)

    def gpu_groups(self):
        """PodInfo.GPUGroups[0] of the active fraction pods (-1 elsewhere): kai_pod_gpu_groups."""
        P = self.snap.n_pods
        out = np.full(max(P, 1), -1, np.int32)
        self.core._check(self.core.lib.kai_pod_gpu_groups(self.core.handle, out.ctypes.data_as(C.POINTER(C.c_int32)), max(P, 1)))
        return out[:P]

    def queue_shares(self):
        Q = self.snap.n_queues
        out = (abi.KaiQueueShare * max(Q, 1))()
        self.core._check(self.core.lib.kai_queue_shares(self.core.handle, out, max(Q, 1)))
        res = {}
        for f in ("fair_share", "allocated", "allocated_non_preemptible", "request", "deserved", "max_allowed"):
            res[f] = np.array([[getattr(out[q], f)[r] for r in range(3)] for q in range(Q)], dtype=np.float64).reshape(Q, 3)
        return res

    def pod_states(self):
        P = self.snap.n_pods
        st, nd = np.zeros(max(P, 1), np.int32), np.zeros(max(P, 1), np.int32)
        self.core._check(self.core.lib.kai_pod_states(self.core.handle, st.ctypes.data_as(C.POINTER(C.c_int32)), nd.ctypes.data_as(C.POINTER(C.c_int32)), max(P, 1)))
        return st[:P], nd[:P]

    def node_states(self):
        N, R = self.snap.n_nodes, self.snap.n_res
        out = (abi.KaiNodeState * max(N, 1))()
        self.core._check(self.core.lib.kai_node_states(self.core.handle, out, max(N, 1)))
        raw = np.frombuffer(out, dtype=np.float64).reshape(max(N, 1), 3, abi.MAX_RES)
        return {"idle": raw[:N, 0, :R].copy(), "releasing": raw[:N, 1, :R].copy(), "used": raw[:N, 2, :R].copy()}

    def stats(self) -> abi.KaiActionStats:
        st = abi.KaiActionStats()
        self.core._check(self.core.lib.kai_action_stats_get(self.core.handle, C.byref(st)))
        return st

    def reset(self):
        """Re-open the session from the HBM-resident snapshot (no host traffic)."""
        self.core._check(self.core.lib.kai_session_reset(self.core.handle))

    def close(self):
        self.core.lib.kai_session_close(self.core.handle)
