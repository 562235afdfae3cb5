/*
 * kai_core.h — C ABI of libkai_core, the MI355X-native scheduling-cycle core.
 *
 * This is the drop-in boundary for the reference's hot path.  A cgo shim on the
 * reference side implements framework.Action (pkg/scheduler/framework/interface.go:41-47) for
 * allocate|consolidation|reclaim|preempt and one framework.Plugin (interface.go:49-55) whose
 * OnSessionOpen packs cache.Snapshot() (pkg/scheduler/cache/cluster_info/cluster_info.go:118-228)
 * into kai_snapshot_soa and calls kai_session_open.  See INTEGRATION.md for the binding.
 *
 * Conventions
 *  - every function returns 0 (KAI_OK) or a negative kai_status; nothing throws, nothing calls back.
 *  - one caller thread per handle; a handle is not re-entrant; several handles may coexist.
 *  - all input buffers are caller-owned host memory (C.malloc'd on the Go side; cgo forbids keeping
 *    Go pointers) and may be freed when the call returns: the library copies them to HBM.
 *  - output buffers are caller-allocated, with a capacity argument.
 *  - indices, not names: the host ranks every string once (node names, UIDs, pod-set names) in
 *    byte-wise order and passes the rank, because the reference's tie-breaks are string compares
 *    (framework/session.go:480-485 node name; session_plugins.go:227-260 UID).
 *  - all quantities are float64 exactly as the reference holds them: cpu in milli-cores, memory in
 *    bytes, gpu in devices, pods as a count (api/resource_info/resource_vector.go:23-36).
 */
#ifndef KAI_CORE_H
#define KAI_CORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KAI_ABI_VERSION 5u

/* resource vector layout: api/resource_info/resource_vector.go:23-36 (cpu, memory, gpu, pods, extras…) */
#define KAI_RES_CPU 0
#define KAI_RES_MEM 1
#define KAI_RES_GPU 2
#define KAI_RES_PODS 3
#define KAI_MAX_RES 8

/* quota resources of the proportion plugin, in its fixed iteration order
 * (plugins/proportion/resource_share/resource_quantities.go:20) */
#define KAI_Q_CPU 0
#define KAI_Q_MEM 1
#define KAI_Q_GPU 2
#define KAI_Q_NRES 3

#define KAI_UNLIMITED (-1.0) /* pkg/common/constants/constants.go:11 */

typedef enum kai_status {
    KAI_OK = 0,
    KAI_ERR_INVALID_ARG = -1,
    KAI_ERR_NO_DEVICE = -2,     /* no usable HIP device: the library never computes on the CPU */
    KAI_ERR_HIP = -3,           /* a HIP runtime call failed (see kai_last_error) */
    KAI_ERR_CAPACITY = -4,      /* an output buffer is too small */
    KAI_ERR_UNSUPPORTED = -5,   /* snapshot needs a feature outside the device path (see kai_pod_flags) */
    KAI_ERR_STATE = -6,         /* call order violated (e.g. action before session_open) */
    KAI_ERR_DEVICE_FAULT = -7,  /* the device engine reported an internal fault / spin timeout */
    KAI_ERR_COMM = -8,          /* multi-GPU exchange failed */
    KAI_ERR_NO_MEMORY = -9      /* host memory ran out while a session was prepared (std::bad_alloc): not a malformed snapshot */
} kai_status;

/* pod status bit-set: api/pod_status/pod_status.go:25-71 */
typedef enum kai_pod_status {
    KAI_POD_PENDING = 1 << 0,
    KAI_POD_GATED = 1 << 1,
    KAI_POD_ALLOCATED = 1 << 2,
    KAI_POD_PIPELINED = 1 << 3,
    KAI_POD_BINDING = 1 << 4,
    KAI_POD_BOUND = 1 << 5,
    KAI_POD_RUNNING = 1 << 6,
    KAI_POD_RELEASING = 1 << 7,
    KAI_POD_SUCCEEDED = 1 << 8,
    KAI_POD_FAILED = 1 << 9,
    KAI_POD_UNKNOWN = 1 << 10,
    KAI_POD_DELETED = 1 << 11
} kai_pod_status;

/* node flag bits (host pre-evaluates the per-node booleans the predicates read) */
#define KAI_NODE_NOT_READY 0x1u      /* scheduler_util/scheduler_utils.go:12-40 says "not fit" */
#define KAI_NODE_MIG_ENABLED 0x2u    /* api/node_info/node_info.go:704-718 */
#define KAI_NODE_MIG_MIXED 0x4u      /* MigStrategy == mixed (node_info.go:720-732) */
#define KAI_NODE_HAS_DRA_GPUS 0x8u   /* node_info.go:95 HasDRAGPUs */
#define KAI_NODE_GPU_WORKER 0x10u    /* has conf GPUWorkerNodeLabelKey (plugins/predicates/predicates.go:243-259) */
#define KAI_NODE_CPU_WORKER 0x20u    /* has conf CPUWorkerNodeLabelKey */
#define KAI_NODE_MIG_SINGLE 0x40u    /* MigStrategy == single (node_info.go:720-732): only whole-GPU tasks may run there (:349-352) */

/* pod flag bits */
#define KAI_POD_FOREIGN_SCHEDULER 0x1u /* spec.schedulerName != ours: plugins/proportion/proportion.go:276-285 */
#define KAI_POD_HAS_TASK_PRIORITY 0x2u /* carries the task-order label: plugins/taskorder/task_order.go:28-63 */
#define KAI_POD_LEGACY_MIG 0x10u       /* PodInfo.IsLegacyMIGtask (pod_info.go:500-516): never schedulable (node_info.go:317-320); a node that holds one takes no MIG request (:342-346) */
#define KAI_POD_CPU_FALLBACK 0x4u      /* needs a state-dependent upstream predicate, fractional GPU, MIG, DRA …:
                                          not placed by the device path (SURVEY §8b fallback rule) */
#define KAI_POD_GPU_UNMODELLED 0x8u    /* holds / asks for GPU state the device's node accounting does not carry (gpu-memory request, several fractional
                                          devices, MIG profile, DRA claim): an ACTIVE pod of that kind would leave its node's idle GPUs overstated
                                          (api/node_info/node_info.go:457-493 addSharedTaskResources), so kai_session_open refuses the snapshot */

typedef enum kai_action {
    KAI_ACTION_ALLOCATE = 0,      /* actions/allocate/allocate.go:46-77 */
    KAI_ACTION_CONSOLIDATION = 1, /* actions/consolidation/consolidation.go:32-78 */
    KAI_ACTION_RECLAIM = 2,       /* actions/reclaim/reclaim.go:47-100 */
    KAI_ACTION_PREEMPT = 3        /* actions/preempt/preempt.go:46-97 */
} kai_action;

typedef enum kai_op_kind {
    KAI_OP_ALLOCATE = 0, /* framework/statement.go:297-358  → cache.Bind on commit  */
    KAI_OP_PIPELINE = 1, /* framework/statement.go:197-295  → cache.TaskPipelined   */
    KAI_OP_EVICT = 2     /* framework/statement.go:63-126   → cache.Evict           */
} kai_op_kind;

typedef enum kai_placement_strategy { KAI_BINPACK = 0, KAI_SPREAD = 1 } kai_placement_strategy;

/* plugins of the default tier that register callbacks on the path (conf_util/scheduler_conf_util.go:39-60).
 * A plugin that is absent registers nothing: its order fn / predicate / score simply does not run. */
#define KAI_PLUGIN_PREDICATES 0x001u
#define KAI_PLUGIN_PROPORTION 0x002u
#define KAI_PLUGIN_PRIORITY 0x004u
#define KAI_PLUGIN_ELASTIC 0x008u
#define KAI_PLUGIN_NODEAVAILABILITY 0x010u
#define KAI_PLUGIN_RESOURCETYPE 0x020u
#define KAI_PLUGIN_SUBGROUPORDER 0x040u
#define KAI_PLUGIN_TASKORDER 0x080u
#define KAI_PLUGIN_NOMINATEDNODE 0x100u
#define KAI_PLUGIN_NODEPLACEMENT 0x200u
#define KAI_PLUGIN_MINRUNTIME 0x400u
#define KAI_PLUGIN_TOPOLOGY 0x800u
/* shared-GPU plugins (fractions of one device, ABI v4): plugins/gpusharingorder (node order), plugins/gpupack and plugins/gpuspread (GPU order inside a node) */
#define KAI_PLUGIN_GPUSHARINGORDER 0x1000u
#define KAI_PLUGIN_GPUPACK 0x2000u
#define KAI_PLUGIN_GPUSPREAD 0x4000u
#define KAI_PLUGIN_ALL 0x3FFFu /* the default tier list: everything above except gpuspread */

/* knobs of conf.SchedulerParams / plugin arguments that the path reads
 * (conf/scheduler_conf.go:31-61, plugins/proportion/proportion.go:67-93,
 *  plugins/nodeplacement/nodeplacement.go:33-47) */
typedef struct kai_config {
    uint32_t abi_version;
    int32_t gpu_strategy;                 /* kai_placement_strategy */
    int32_t cpu_strategy;
    double k_value;                       /* proportion kValue (<=0 → 0) */
    double reclaimer_saturation_multiplier;
    uint32_t plugins;                     /* KAI_PLUGIN_* bit-set: which plugins of the tier list are loaded */
    int32_t restrict_node_scheduling;
    int32_t max_consolidation_preemptees; /* -1 = unlimited */
    int32_t use_scheduling_signatures;
    int32_t allow_consolidating_reclaim;
    int32_t full_hierarchy_fairness;
    int64_t min_node_gpu_memory;          /* ClusterInfo.MinNodeGPUMemory */
    int32_t queue_depth[4];               /* per kai_action; -1 = infinite (framework/session.go:398-404) */
    int32_t engine_mode;                  /* 0 = default (allocate: batch plan/fill/apply path when the action qualifies, else the sequential engine);
                                             1 = force brute-force node scans; 2 = class index without the staged job path; 3 = sequential engine only (debug / A-B) */
    int32_t reserved[7];
    /* minruntime plugin (plugins/minruntime/minruntime.go:40-100): "now" of the cycle, plugin-argument defaults, reclaim resolve method */
    int64_t now_ns;
    int64_t default_preempt_min_runtime_ns;
    int64_t default_reclaim_min_runtime_ns;
    int32_t reclaim_resolve_method;       /* 0 = lca (default), 1 = queue */
    int32_t pad0;
} kai_config;

/* Structure-of-arrays session snapshot.  [R][N] means resource-major: element (r, i) at r*N + i. */
typedef struct kai_snapshot_soa {
    uint32_t abi_version;
    int32_t n_res; /* R, 4..KAI_MAX_RES */

    /* ---- nodes: api/node_info/node_info.go:68-105 ---- */
    int32_t n_nodes;
    const double* node_allocatable;   /* [R][N] status.allocatable */
    const uint32_t* node_flags;       /* [N] KAI_NODE_* (bits 28-31 are ignored: the library keeps flags of its own there) */
    const int32_t* node_gpu_count;    /* [N] nvidia.com/gpu.count label, -1 when absent (node_info.go:619-640) */
    const uint32_t* node_name_rank;   /* [N] rank of the node name, byte-wise ascending, unique */
    const int32_t* node_class;        /* [N] column of class_fit */

    /* ---- pods: api/pod_info/pod_info.go:70-112; pods of one job are contiguous ---- */
    int32_t n_pods;
    const double* pod_req;            /* [R][P] ResReq incl. pods:=1 (pod_info.go:373-393) */
    const int32_t* pod_job;           /* [P] */
    const int32_t* pod_podset;        /* [P] global pod-set index */
    const int32_t* pod_status;        /* [P] kai_pod_status */
    const int32_t* pod_node;          /* [P] node index or -1 */
    const uint32_t* pod_flags;        /* [P] KAI_POD_* flag bits */
    const int32_t* pod_task_priority; /* [P] value of the task-order label (valid with KAI_POD_HAS_TASK_PRIORITY) */
    const int64_t* pod_created_ns;    /* [P] Pod.CreationTimestamp */
    const uint32_t* pod_uid_rank;     /* [P] rank of the pod UID, unique */
    const int32_t* pod_class;         /* [P] row of class_fit */
    const int32_t* pod_nominated_node;/* [P] node index of status.nominatedNodeName or -1 */

    /* ---- pod-sets (gang sub-groups): api/podgroup_info/subgroup_info/podset.go:17-29 ---- */
    int32_t n_podsets;
    const int32_t* podset_job;           /* [S] */
    const int32_t* podset_min_available; /* [S] */
    const uint32_t* podset_name_rank;    /* [S] rank of the pod-set name inside its job */

    /* ---- jobs (PodGroups): api/podgroup_info/job_info.go:65-103 ---- */
    int32_t n_jobs;
    const int32_t* job_queue;         /* [J] leaf queue index, -1 if the queue does not exist */
    const int32_t* job_priority;      /* [J] */
    const int32_t* job_preemptible;   /* [J] 0/1 (pkg/common/podgroup/preemptible.go:10-26) */
    const int64_t* job_created_ns;    /* [J] */
    const uint32_t* job_uid_rank;     /* [J] unique */
    const int32_t* job_first_pod;     /* [J] */
    const int32_t* job_n_pods;        /* [J] */
    const int32_t* job_first_podset;  /* [J] */
    const int32_t* job_n_podsets;     /* [J] */

    /* ---- queues: api/queue_info/queue_info.go:32-43 ---- */
    int32_t n_queues;
    const int32_t* queue_parent;      /* [Q] -1 for top queues */
    const int32_t* queue_priority;    /* [Q] */
    const int64_t* queue_created_ns;  /* [Q] */
    const uint32_t* queue_uid_rank;   /* [Q] unique */
    const double* queue_deserved;     /* [3][Q] quota; cpu milli, memory in 10^6-byte units as in the CRD, gpu devices; -1 unlimited */
    const double* queue_limit;        /* [3][Q] */
    const double* queue_oqw;          /* [3][Q] over-quota weight */
    const double* queue_usage;        /* [3][Q] normalised historical usage (api/queue_info/quota_info.go) */

    /* ---- static predicate classes (upstream kube-scheduler Filters pre-evaluated by the host) ---- */
    int32_t n_pod_classes;
    int32_t n_node_classes;
    const uint8_t* class_fit;         /* [n_pod_classes][n_node_classes] 1 = every static Filter passes */

    /* ---- topologies (Topology CR: ordered spec.levels[].nodeLabel; plugins/topology/topology_plugin.go:57-110).
     * Levels of topology t are the global level rows topo_level_off[t] .. topo_level_off[t+1]-1, top level first.  A domain is one
     * distinct prefix of level label values (plugins/topology/topology_structs.go:94-101); the host builds the domain table.
     * All of this may be absent (n_topologies = 0). ---- */
    int32_t n_topologies;
    const int32_t* topo_level_off;    /* [T+1] */
    int32_t n_topo_levels;            /* = topo_level_off[T] */
    const int32_t* node_domain;       /* [n_topo_levels][N] domain of the node at that level, -1 at every level of a topology the node is not
                                         part of (a node joins a topology only if it has every level label: topology/common.go:70-77) */
    int32_t n_domains;
    const int32_t* domain_level;      /* [D] global level row */
    const int32_t* domain_parent;     /* [D] domain at the level above, -1 for a top-level domain */
    const uint32_t* domain_id_rank;   /* [D] rank of the domain ID string ("zone1.rack3") inside its topology, byte-wise ascending */

    /* ---- sub-group tree of every job (api/podgroup_info/subgroup_info/subgroupset.go): groups = SubGroupSets incl. each job's root;
     * pod-sets are the leaves.  Absent (n_groups = 0) ⇒ every job has a root group without constraint holding all its pod-sets.
     * A topology constraint (api/topology_info) is {topology index or -1 (none) or -2 (named topology does not exist),
     * required level, preferred level} with levels counted inside the topology (0 = top) or -1. ---- */
    int32_t n_groups;
    const int32_t* group_job;         /* [G] */
    const int32_t* group_parent;      /* [G] parent group, -1 for the job's root */
    const uint32_t* group_name_rank;  /* [G] rank of the SubGroupSet name inside its job (framework/session_plugins.go:273-282) */
    const int32_t* group_topology;    /* [G] */
    const int32_t* group_required_level;
    const int32_t* group_preferred_level;
    const int32_t* job_root_group;    /* [J] */
    const int32_t* podset_group;      /* [S] parent group of the pod-set */
    const int32_t* podset_topology;   /* [S] the pod-set's own constraint */
    const int32_t* podset_required_level;
    const int32_t* podset_preferred_level;

    /* ---- scheduling-constraints signature (api/podgroup_info/job_info.go:547-570, api/pod_info/scheduling_constraints_signature.go) ----
     * [J] any injective id of the job's signature (the reference hashes node selector, affinity, tolerations, priority class, ...).
     * Only equality is used (actions/common/minimal_job_comparison.go:15-44).  NULL: the victim actions refuse to run with
     * use_scheduling_signatures set. */
    const int64_t* job_signature;

    /* ---- minruntime plugin inputs (all optional; NULL = nothing is protected) ----
     * PodGroupInfo.LastStartTimestamp (ns since the epoch, 0 = never started) and the queues' preemptMinRuntime / reclaimMinRuntime
     * (ns, -1 = not set on that queue; resolved up the tree, plugins/minruntime/resolver.go:33-190) */
    const int64_t* job_last_start_ns;            /* [J] */
    const int64_t* queue_preempt_min_runtime_ns; /* [Q] */
    const int64_t* queue_reclaim_min_runtime_ns; /* [Q] */

    /* ---- fractional GPU requests (ABI v4; all optional, NULL = no pod asks for a fraction) ----
     * pod_gpu_portion: 0 = a whole-GPU or CPU-only pod; in (0, 1) = a fraction of ONE device (annotation gpu-fraction,
     *   api/pod_info/pod_info.go:472-477); pod_req's gpu column then holds ResourceRequirements.GPUs() of it: the portion rounded to 1/100
     *   (fixed point, api/resource_info/gpu_resource_requirment.go:230-234) — 0.125 counts as 0.13 against quota, and as int64(0.125 * memory) on the device.
     * pod_gpu_group: the shared-GPU group an ACTIVE fraction pod runs in (label runai-gpu-group; PodInfo.GPUGroups), as an id >= 0 that is
     *   unique on its node; -1 = none.  Equality on one node is what the accounting uses (api/node_info/gpu_sharing_node_info.go); in addition
     *   ids below 2^20 stand for numeric group names and ids from 2^20 on for any other name, because the predicates plugin takes a
     *   non-numeric name for a group that is being created (plugins/predicates/predicates.go:320-330).
     * node_gpu_memory: NodeInfo.MemoryOfEveryGpuOnNode in MiB (label nvidia.com/gpu.memory floored to a multiple of 100, node_info.go:673-687);
     *   NULL = 100 on every node (DefaultGpuMemory).
     * pod_gpu_memory (ABI v5): > 0 = the pod asks for that many MiB of ONE device (annotation gpu-memory, pod_info.go:463-468): its gpu
     *   column in pod_req and its pod_gpu_portion are 0 (ResourceRequirements.GPUs() of such a request is 0); on a node it takes that memory of a
     *   shared device and counts as ceil(memory / node_gpu_memory * 100) / 100 of a device (node_info.go:329-332, 661-666, 734-744); while pending it
     *   weighs memory / kai_config.min_node_gpu_memory in its queue's request and in its job's resources (proportion.go:360-366,
     *   allocation_info.go:103-107).  NULL = no such pod. */
    const double* pod_gpu_portion;   /* [P] */
    const int32_t* pod_gpu_group;    /* [P] */
    const int64_t* node_gpu_memory;  /* [N] */
    const int64_t* pod_gpu_memory;   /* [P] */

    /* ---- MIG profiles (ABI v5; optional, NULL = no resource row is a MIG profile) ----
     * A MIG instance type (nvidia.com/mig-<g>g.<m>gb) is a resource row >= 4 of node_allocatable / pod_req, counted in instances.  res_mig_gpus[r] = g (its GPU
     * weight, api/common_info/resources/mig.go:13-33), 0 for every other row; res_mig_memory[r] = m.  A pod that requests such a row is a MIG request
     * (RequestTypeMigInstance, pod_info.go:493-497): its GPU quota is sum(g x instances) while ResourceRequirements.GPUs() stays 0
     * (gpu_resource_requirment.go:163-178); nodes count idle instances by their weight (resource_info.go:177-194, node_info.go:592-628); the predicates
     * of api/node_info/node_info.go:315-359 apply (KAI_NODE_MIG_* flags, KAI_POD_LEGACY_MIG). */
    const int32_t* res_mig_gpus;     /* [R] */
    const int64_t* res_mig_memory;   /* [R] */
} kai_snapshot_soa;

typedef struct kai_op {
    int64_t seq;   /* position in commit order */
    int32_t kind;  /* kai_op_kind */
    int32_t pod;
    int32_t node;
    int32_t job;   /* the pod's own job */
    int32_t stmt;  /* Statement the operation was committed by (framework/statement.go:536-575), numbered from 0 per action in commit order:
                      the shim replays the operations of one id through ONE Statement — a reclaim statement is "evict A, evict B, pipeline C" */
    int32_t pad;
} kai_op;

/* per queue, in KAI_Q_* order: plugins/proportion/resource_share/resource_share.go:12-21 */
typedef struct kai_queue_share {
    double fair_share[KAI_Q_NRES];
    double allocated[KAI_Q_NRES];
    double allocated_non_preemptible[KAI_Q_NRES];
    double request[KAI_Q_NRES];
    double deserved[KAI_Q_NRES];
    double max_allowed[KAI_Q_NRES];
} kai_queue_share;

typedef struct kai_node_state {
    double idle[KAI_MAX_RES];
    double releasing[KAI_MAX_RES];
    double used[KAI_MAX_RES];
} kai_node_state;

/* engine statistics of the last kai_action_execute (measurement) */
typedef struct kai_action_stats {
    int64_t decisions;          /* allocateTask executions (ended in allocate / pipeline / fail) */
    int64_t node_scans;         /* full passes over a node set */
    int64_t nodes_scanned;      /* Σ node-set sizes over those passes */
    int64_t jobs_attempted;
    int64_t jobs_committed;
    int64_t rollbacks;
    double kernel_ms;           /* HIP-event time of the action kernel on its stream */
    double upload_ms;           /* snapshot → HBM (session_open only) */
    int64_t reserved[8];        /* [0] index queries, [1] block refreshes / loads, [2] drained jobs | scenarios, [3] drained decisions | simulations,
                                   [4] rounds of the batch path (0 = sequential engine), [5..7] cycle / time counters of the path that ran (kai_core.hip).
                                   Victim actions (consolidation / reclaim / preempt): [1] workgroups the action ran on (default 32, environment KAI_VICTIM_WGS;
                                   1 with shared GPUs in the session), [5] waves of simulations, [6] simulations run << 32 | simulations the reference's order reaches.
                                   Allocate on the sequential engine of a cluster of >= 1024 nodes: bits 48.. of [1] = workgroups that took the passes over the nodes
                                   (scan grid: the engine's own + the helpers; default 16 launched, 32 from 4096 nodes, environment KAI_SCAN_WGS, 1 = off) */
} kai_action_stats;

typedef struct kai_core kai_core; /* opaque */

/* replaces: scheduler.NewScheduler wiring (pkg/scheduler/scheduler.go:53-101); gpu_ids = HIP device ordinals.
 * n_gpus == 1: the whole session on gpu_ids[0].  n_gpus > 1: ONE PROCESS PER GPU — this handle is one rank of a group of n_gpus handles that
 * shard the NODE axis of one session (contiguous name-rank ranges; pods, jobs, queues replicated) on the GPUs of one node; gpu_ids[0] is this
 * rank's device and kai_shard_attach names the rank and the group's all-gather.  Every rank opens the same snapshot and makes the same calls. */
int kai_core_create(const kai_config* cfg, int n_gpus, const int* gpu_ids, kai_core** out);

/* The group's exchange step: an all-gather of `bytes_per_rank` bytes from every rank's `send` into `recv` (rank-major), on DEVICE
 * memory of the library.  The library has no communicator of its own: the caller supplies the collective (the Python mirror:
 * torch.distributed.all_gather_into_tensor over RCCL / xGMI) and the library calls it between kernels, with its stream synchronised.
 * offers_per_class: nodes a rank offers per scan class and exchange (0 = default 128).  Returns 0 / a kai_status. */
typedef int (*kai_allgather_fn)(void* user, const void* send, void* recv, int64_t bytes_per_rank);
int kai_shard_attach(kai_core* core, int rank, int world, int offers_per_class, kai_allgather_fn fn, void* user);
/* Victim actions (reclaim, preempt, consolidation) on a group: every rank holds the whole session, the SIMULATIONS of a partial job are dealt out over the ranks
 * (simulation i of a wave to rank i mod n_gpus; the reference's loop: actions/common/solvers/job_solver.go:95-126, by_pod_solver.go:61-144) and a wave ends with one
 * all-gather of its outcomes (68 bytes per simulation).  That exchange happens while the action's kernel is running and waits for it, so it is the host's: `fn` is
 * called from inside kai_action_execute with HOST memory of the library and must not synchronise the device (the kernel only goes on when fn has returned).
 * Without it — and without the library's own communicator below — a group runs its victim actions replicated (same results, nothing shortened). */
int kai_shard_attach_host(kai_core* core, kai_allgather_fn fn, void* user);
/* The same exchange from the library itself: its own RCCL communicator (librccl is resolved at run time), the all-gather issued on the library's stream between the
 * kernels it separates — no host round trip, no staging.  Rank 0 draws the 128-byte id (ncclGetUniqueId), the caller carries it to the other ranks by whatever it has
 * (the Python mirror: torch.distributed.broadcast), every rank attaches with it (ncclCommInitRank: collective over the group, one device per rank).
 * kai_shard_allgather_probe runs the group's exchange step once on caller-provided device buffers (diagnostics; what the fill calls between kernels). */
#define KAI_RCCL_ID_BYTES 128
int kai_shard_rccl_id(kai_core* core, void* id_out /* KAI_RCCL_ID_BYTES */);
int kai_shard_attach_rccl(kai_core* core, int rank, int world, int offers_per_class, const void* id /* KAI_RCCL_ID_BYTES, the same on every rank */);
int kai_shard_allgather_probe(kai_core* core, const void* send, void* recv, int64_t bytes_per_rank);
int kai_core_destroy(kai_core* core);

/* replaces: framework.OpenSession + every OnSessionOpen on the path (framework/framework.go:32-65):
 * node accounting from the pods (api/node_info/node_info.go:457-493), proportion totals / queue usage /
 * fair-share division (plugins/proportion/proportion.go:242-423).
 * The snapshot's arrays are read during the call only.  The handle keeps what it derives from them on the host (name-rank permutation, job lists, scan
 * classes: about 150 bytes per pod) and its device slabs for the next open, as a scheduler opens a session per cycle (scheduler.go:112-138); kai_core_destroy
 * releases both.  The loops of the preparation run on a process-wide pool of worker threads (KAI_HOST_THREADS, KAI_HOST_POOL: INTEGRATION.md). */
int kai_session_open(kai_core* core, const kai_snapshot_soa* snap);

/* Re-opens the session from the snapshot copy that is already resident in HBM (no host traffic): same math as
 * kai_session_open.  Lets a caller replay scheduling cycles on one snapshot (benchmarks, what-if runs). */
int kai_session_reset(kai_core* core);

/* replaces: ssn.QueueFairShare / QueueAllocatedResources / QueueDeservedResources
 * (plugins/proportion/proportion.go:508-521) */
int kai_queue_shares(kai_core* core, kai_queue_share* out, int cap);

/* replaces: Action.Execute(ssn) (actions/allocate/allocate.go:46-77, reclaim.go:47-100, preempt.go:46-97,
 * consolidation.go:32-78) up to, not including, the cache side effects of Statement.Commit
 * (framework/statement.go:536-575): the committed operations come back in commit order and the Go shim
 * replays them through cache.Bind / cache.Evict / cache.TaskPipelined. */
int kai_action_execute(kai_core* core, int action, kai_op* ops_out, int64_t ops_cap, int64_t* n_ops);

/* replaces: Session.OrderedNodesByTask + FittingNode for one task (framework/session.go:201-264),
 * against the session's current node state.  nodeset_bitmap (bit n of word n/32 = node index n, ceil(N/32) words) restricts the
 * node set as SubsetNodesFn would (framework/session_plugins.go:345-366); NULL = all nodes.  node_idx_out = -1 when nothing fits. */
int kai_best_node(kai_core* core, int32_t pod_idx, const uint32_t* nodeset_bitmap, int pipeline_only,
                  int32_t* node_idx_out, int* is_pipeline_out);

/* session-state read-back (what the shim mirrors into PodInfo.Status/NodeName and NodeInfo.Idle/Releasing) */
int kai_pod_states(kai_core* core, int32_t* status_out, int32_t* node_out, int cap);
int kai_node_states(kai_core* core, kai_node_state* out, int cap);

/* PodInfo.GPUGroups[0] of every active fraction pod after the actions run so far (what a BindRequest carries as SelectedGPUGroups,
 * cache/cache.go:290-330): the snapshot's group id, or an id >= 2^20 for a group the cycle opened (gpu_sharing/gpuSharing.go:73-83); -1 otherwise */
int kai_pod_gpu_groups(kai_core* core, int32_t* groups_out, int cap);

int kai_action_stats_get(kai_core* core, kai_action_stats* out);

/* replaces: framework.CloseSession (framework/framework.go:67-78) — frees the session's HBM */
int kai_session_close(kai_core* core);

/* human-readable detail for the last non-zero status on this handle (never NULL) */
const char* kai_last_error(kai_core* core);

/* library build info: "kai_core <abi> gfx950 …" */
const char* kai_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KAI_CORE_H */
