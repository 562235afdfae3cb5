// kai_engine.hpp — device-resident scheduling-cycle engine (MI355X / gfx950).
//
// The whole session lives in HBM as structure-of-arrays (KaiCtx).  One persistent kernel executes an
// Action: lane 0 of wave 0 runs the sequential control flow of the reference (fair job order, gang
// loop, statement log — all of it inherently serial) and every other wavefront of the
// workgroup serves it: they keep the per-class arg-max index over the node SoA current (class index,
// below) and run brute-force node scans where the index does not apply.  This header holds the control
// flow and the per-node arithmetic; kai_kernels.hpp holds the cooperative parts, the session-open /
// action-init kernels and the launch code.
//
// Class index.  Pending pods fall into a few scan classes (same request vector and predicate class).
// For a class the reference's order over nodes (Σ NodeOrderFns desc, name asc — framework/session.go:234-264,
// 466-485) is the order of a per-node 64-bit key that depends only on that node's own state (class_key()),
// so  OrderedNodesByTask + FittingNode  ==  arg-max of the key over the fitting nodes.  The engine keeps, per class,
// the arg-max of every 64-node block (L1, HBM), of every 64-block super-block (L2, LDS) and of the whole cluster,
// and after a placement / rollback re-evaluates only the blocks whose nodes changed.  Node indices inside the
// engine are name ranks (the host permutes), so "name asc" is "index asc".
//
// Everything here is `KAI_HD` so that tests/host_sim can compile the identical control flow for the
// host to debug it without a GPU.  The product library only ever instantiates the device backend and
// fails without a HIP device (see kai_core.hip); nothing in it runs on the CPU.
//
// Reference citations are file:line under pkg/scheduler.
#pragma once
#include <stdint.h>

#include "../../include/kai_core.h"

#if defined(__HIPCC__)
#define KAI_HD __host__ __device__ __forceinline__
#define KAI_HD_NOINLINE __host__ __device__
#define KAI_HDS __host__ __device__ __attribute__((noinline))  // victim search: real calls, or the one persistent kernel inlines itself into hours of compile time
#else
#define KAI_HD inline
#define KAI_HD_NOINLINE
#define KAI_HDS inline
#endif

// KAI_GP(T): pointer to T in HBM.  On the device it is an address-space-1 ("global") pointer so that every array access is a
// global_load / global_store (vmcnt only); through generic pointers they would be flat_* operations, which also count on
// lgkmcnt and so serialise against every LDS read of the engine.  Same size and layout on the host.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> using kai_gptr = T __attribute__((address_space(1)))*;
#else
template <class T> using kai_gptr = T*;
#endif
#define KAI_GP(T) kai_gptr<T>

#include "kai_batch_types.hpp"

namespace kai {

// pod status groups: api/pod_status/pod_status.go:64-71
KAI_HD bool st_active_used(int s) { return s & (KAI_POD_ALLOCATED | KAI_POD_PIPELINED | KAI_POD_BINDING | KAI_POD_BOUND | KAI_POD_RUNNING | KAI_POD_RELEASING); }
KAI_HD bool st_active_allocated(int s) { return s & (KAI_POD_ALLOCATED | KAI_POD_PIPELINED | KAI_POD_BINDING | KAI_POD_BOUND | KAI_POD_RUNNING); }
KAI_HD bool st_alive(int s) { return s & (KAI_POD_ALLOCATED | KAI_POD_PIPELINED | KAI_POD_BINDING | KAI_POD_BOUND | KAI_POD_RUNNING | KAI_POD_PENDING | KAI_POD_GATED); }
KAI_HD bool st_allocated(int s) { return s & (KAI_POD_ALLOCATED | KAI_POD_BOUND | KAI_POD_BINDING | KAI_POD_RUNNING); }

KAI_HD double kmin(double a, double b) { return a < b ? a : b; }  // math.Min / math.Max on finite, non-NaN operands
KAI_HD double kmax(double a, double b) { return a > b ? a : b; }

// plugins/proportion/resource_share/resource_share.go:12-21
struct QShare {
    double deserved, fair, max_allowed, oqw, allocated, allocated_np, request, usage;
};
KAI_HD double qs_requestable(const QShare& s) { return s.max_allowed == KAI_UNLIMITED ? s.request : kmin(s.max_allowed, s.request); }  // :41-46
KAI_HD double qs_allocatable(const QShare& s) {  // :51-61
    if (s.deserved == KAI_UNLIMITED) return s.max_allowed;
    double a = kmax(s.deserved, s.fair);
    if (s.max_allowed != KAI_UNLIMITED) a = kmin(s.max_allowed, a);
    return a;
}

enum OpName : int32_t { OP_EVICT = 0, OP_PIPELINE = 1, OP_ALLOCATE = 2, OP_UNDO = 3 };
// framework/operations.go
struct StmtOp {
    int32_t name, pod, prev_status, prev_node, next_node, prev_virtual, op_index, pad;
    int32_t undo_link, pad2;  // undo_link: 1 + index of the FIRST OP_UNDO on the log that names this operation (0 = none was ever pushed; a link into a truncated part of the log is checked before use)
};

// scan class: every pod with the same request vector and static-predicate class (host: kai_host_prep.hpp)
struct ClassRec {
    double req[KAI_MAX_RES];
    int32_t pod_class, cpu_only, best_effort, r_place, strategy, pad[3];
};
constexpr int KAI_CMAX = 64;      // classes the index tracks (the most frequent ones); the rest use brute-force scans
constexpr int KAI_BLOCK = 64;     // nodes per L1 block = one wavefront
constexpr int KAI_NSB_MAX = 64;   // super-blocks the LDS level holds → the index covers N <= 64*64*64 = 262144 nodes
constexpr int KAI_MAXD = 16;      // dirty blocks buffered before a refresh is forced

// One node of the job-order tree (actions/utils/job_order_by_queue.go queueNode) with the cached operands of the queue
// comparator (plugins/proportion/queue_order/queue_order.go:19-73).  40 bytes, so the whole tree fits the LDS of one CU
// for a few thousand queues (kai_kernels.hpp puts it there when it does).
struct QNode {
    double dom_with_job, dom_no_job;  // dominant share with / without the best job of the subtree
    int32_t best_job;                 // best job of the subtree (valid with QF_VALID)
    int32_t prio;                     // queue priority (static)
    int32_t len;                      // children in the node's heap; for a leaf: jobs queued
    int32_t heap_off;                 // base of the node's child heap inside qheap (static)
    int32_t parent;                   // parent queue or -1 (static)
    uint32_t flags;                   // QF_*
};
enum : uint32_t { QF_OVER = 1, QF_STARVED = 2, QF_VIOL = 4, QF_VALID = 8, QF_REORDER = 16, QF_EXISTS = 32, QF_LINKED = 64, QF_LEAF = 128, QF_DNJ = 512,
                  QF_TOP = 256 };  // leaf: best_job holds the top job of the leaf (kept across key invalidations, dropped on pop / push)

// Working set of one job attempt on the staged ("fast") path: the chunk's pods with their request vectors and the shares of
// the queues on the job's leaf→root chain, held in LDS for the duration of the attempt (kai_kernels.hpp) and written back once.
constexpr int KAI_FMAX = 32;   // tasks per chunk the frame holds; larger gangs take the general path
constexpr int KAI_FDEPTH = 8;  // queue levels
constexpr int KAI_TDEPTH = 16;  // frames of the sub-group DFS (group depth + the pod-set level)
constexpr int KAI_TKEYS = 32;   // sub-groups of one job that carry preferred-level node scores
struct FastFrame {
    int32_t depth, np, n_lim, n_dsv;   // n_lim / n_dsv: entries of lim_idx / dsv_idx
    uint8_t lim_idx[KAI_FDEPTH * 3], dsv_idx[KAI_FDEPTH * 3];  // (level * 3 + resource) of the finite limits / finite quotas on the chain: the only entries the capacity checks can trip on
    int32_t q[KAI_FDEPTH];
    int32_t p[KAI_FMAX], cls[KAI_FMAX], node[KAI_FMAX];
    double req[KAI_FMAX][KAI_MAX_RES];
    double alloc[KAI_FDEPTH][3], alloc_np[KAI_FDEPTH][3], max_allowed[KAI_FDEPTH][3], deserved[KAI_FDEPTH][3];
};

// The frame is addressed directly (not through a generic pointer) so that the device code uses ds_read / ds_write, which the
// compiler may batch and reorder against global memory traffic.
#if defined(__HIPCC__)
__shared__ FastFrame kai_frame_lds;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define KAI_FRAME kai_frame_lds
#else
inline FastFrame& kai_frame_host() { static thread_local FastFrame f; return f; }
#define KAI_FRAME kai_frame_host()
#endif

// Job-level fields of the staged path, gathered ahead of the attempt (Backend::stage_async): on the device a service wave loads them, the
// chunk's pods, classes and request vectors straight into the frame — one wide gather per dependent level instead of ~40 loads issued one by
// one by the control lane — while the control lane finishes the pop (heap fixes, key recomputation).
struct JobPf {
    int32_t job, ok;      // ok: the frame (p, cls, req) and the fields below are staged for `job` and the job has the staged path's shape
    int32_t shape, n_ps, has_topo, s, first, tta_valid, tta_n, jq, jpre, s_pipelined;
    double ja[3];
};
#if defined(__HIPCC__)
__shared__ JobPf kai_pf_lds;  // declared in both passes of hipcc: the device-only code of kai_kernels.hpp names it directly
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define KAI_JOBPF kai_pf_lds
#else
inline JobPf& kai_pf_host() { static thread_local JobPf f; return f; }
#define KAI_JOBPF kai_pf_host()
#endif

constexpr int KAI_NPROF = 40;
struct EngineState {  // mutable scalars of the running action
    int32_t ops_len, n_undo;
    int32_t fault;            // != 0: engine gave up (see FAULT_*)
    int32_t fault_line, pad_f; // source line of the fault() call (diagnostics)
    int32_t drain_pending;    // allocate: every class is dead at a committed state → the rest of the queue is counted by k_drain
    int64_t out_len, stmts;   // committed operations / Statements that committed at least one operation, of this action
    int64_t decisions, node_scans, nodes_scanned, jobs_attempted, jobs_committed, rollbacks;
    int64_t index_queries, index_refreshes, drained_jobs, drained_decisions;
    int64_t scenarios, simulations, scenarios_filtered;  // victim search (actions/common/solvers)
    int64_t non_allocate_commits;  // evictions / pipelines committed by this action: from then on something is releasing or pipelined in the session
    int64_t prof[KAI_NPROF];  // control-lane cycles per phase, see PF_* (16..39: victim search, -DKAI_PROF_VICTIM)
    double total[3];          // proportion totalResource (CPU, Memory, GPU)
};
enum { PF_POP = 0, PF_ALLOC = 2, PF_FINISH = 3, PF_DRAINCHK = 4, PF_INIT = 5, PF_TOTAL = 7, PF_TTA = 8, PF_GATE = 9, PF_TASKCAP = 10, PF_FIND = 11,
       PF_REFRESH = 12, PF_STMT = 13, PF_ROLLBACK = 14, PF_PUSH = 15 };
enum { FAULT_NONE = 0, FAULT_OPS_CAP = 1, FAULT_OUT_CAP = 2, FAULT_HEAP = 3, FAULT_INTERNAL = 4, FAULT_SPIN = 5 };

// Victim search on several workgroups (kai_engine_solver.inc solve_partial_multi): every workgroup runs the SAME control flow on its own replica of the session
// state (KaiCtx pointers rebased into the replica); the simulations of one partial job are dealt out to them in waves, and this block — the only memory they share —
// carries each wave's outcomes and the grid barrier.  One instance per victim action, zeroed by the host before the launch.
struct XMail { int32_t req, resp, buf, pad; };  // mailbox of a wave's exchange over the GPUs of a group (kai_victim_shard.hpp), pinned host memory: device → host req = number of the
                                                // exchange asked for, buf = the wave's buffer; host → device resp = req once the merged wave is in MultiCtx
constexpr int KAI_MW_MAX = 256;    // workgroups of one victim action at most (one per compute unit)
constexpr int KAI_MW_WAVE = 1024;  // simulations of one wave at most
constexpr int KAI_MW_CNT = 8;      // counter deltas a simulation reports (see Engine::mw_cnt_get)
struct MultiCtx {
    int32_t world, fault, bar_count, bar_gen;
    int32_t next[2], hit[2];                     // per wave (double buffered): the next simulation to hand out; the lowest simulation so far that did not simply fail (INT_MAX: none)
    int32_t res[2][KAI_MW_WAVE];                 // … SIM_* status | MW_TOUCHED of simulation i
    int64_t cnt[2][KAI_MW_WAVE][KAI_MW_CNT];     // … and the counters it bumped
    int64_t waves, sims_run, sims_used, replays;  // diagnostics (rank 0 adds what it sees)
    // the same waves dealt out over the GPUs of a node-sharded group (kai_victim_shard.hpp): written between the two barriers of a wave's end by whoever carries the
    // exchange (the host behind the mailbox on the device, engine 0 itself on the emulator); res / cnt / hit of buffer b then describe the wave of the whole group
    int32_t xrun[2];                              // per wave: simulations 0 .. xrun-1 were run by the ranks that own them (the counted prefix when nothing hit)
    int32_t xdone, xpad;                          // a rank left the protocol (fault): every engine of every rank gives up
};

// Scratch of the victim search (reclaim / preempt / consolidation, kai_engine_solver.inc), all in HBM.  "View" of a victim job =
// its pods that are not yet taken into a task group of the scenario (and, once the builder met a recorded victim of the job, not
// recorded): what the reference holds as CloneWithTasks(remainingTasks) in the victims queue (solvers/pod_scenario_builder.go:91-133).
struct SolverCtx {
    uint8_t *vq_state, *tpl_state;                    // [J+1] elastic state of a victim job's view, cached while the view stands (255 = not known); state of a pending job for the template (255 = not pending)
    uint8_t *vq_in, *vq_excl, *ja_in;                 // [J] victims-queue membership, "view excludes recorded tasks", jobsToAllocate membership (bit 1 = victim job)
    uint8_t *p_taken, *p_recorded, *p_partial, *ig_cache;  // [P]
    int32_t *p_grp;                                   // [P] task group (representative clone) of a victim pod in the current scenario
#ifdef KAI_SHARED_GPUS
    int32_t* xr_group;  // shared-GPU group of that further copy
#endif
    int64_t* xr_key; int32_t* xr_status; int32_t xr_mask, xr_pad;  // further node residencies of a pod, hash (pod, node) → status the node's copy was added with:
                                                      // an evicted task stays on its node as Releasing while it is pipelined elsewhere (statement.go:197-295)
    // the two solver-internal job-order instances (index 0 = victims queue, 1 = jobs to allocate): same records and heaps as the action's own
    QNode* i_qn[2]; int32_t *i_qheap[2], *i_root[2], *i_cur[2], *i_end[2], *i_side[2], *i_side_len[2];  // [Q], [Q+1], [Q+1], [Q], [Q], [J], [Q]
    double* vq_pop;                                   // [Q][3] Σ Allocated of the jobs popped from a leaf (poppedJobsByQueue, job_order_by_queue.go:80-82)
    int32_t* s_ov_min;                                // [S] minAvailable of the partial preemptor representative (job_solver.go:128-151)
    int32_t *mw_end, *mw_filt;                        // [P+J+2] scenarios the builder has produced for the running partial job: group count of scenario k, scenarios the filters dropped before it
    uint32_t* mw_feas; int32_t *mw_nstamp, *mw_nmaxk, *mw_wk, *mw_wn, *mw_ctr;  // [W+1] feasible nodes a wave starts from; [N+1] x 2 per node: wave stamp and the last scenario of the wave's consumed simulations on it; [KAI_MW_WAVE] x 2 the wave's simulations
    int32_t *mw_sj, *mw_sv, *mw_sn, *mw_stta; double* mw_sres;  // [J+1] x 3, [P+1], [4(J+1)]: tasks-to-allocate caches of the jobs a speculative simulation touches, saved aside
    int32_t *grp_job, *grp_off, *grp_pods, *grp_ord;  // task groups of the scenario: recorded first, then potential in the order they were added; grp_pods = a group's pods in canonical order
                                                      // (what ranging the representative's pod map stands for), grp_ord = in the order they were handed over (VictimInfo.Tasks, potentialVictimsTasks: slices)
    int32_t *rec_job, *rec_off, *rec_pods;            // recorded victim jobs handed to the next partial job (result.victimJobs)
    int32_t *res_tasks, *ev_tasks, *vt_tasks, *pend, *tmp, *tmp2, *tmp3;  // [P] result.victimsTasks, GetTasksToEvict output, victims of the running simulation, tasks of the job being solved, scratch
    uint32_t *feas, *feas0;                           // [W] byPodSolver.feasibleNodes / JobSolver.feasibleNodes as node bitmaps
    double* ig_idle; int32_t* ig_sorted;              // [N], [P] AccumulatedIdleGpus state
    QShare* q_sim;                                    // [Q][3] proportion.jobSimulationQueues
    double *ta_cap, *ta_req, *ta_virt; int32_t *ta_sorted, *ta_off, *ta_row, *ta_sg, *ta_sg_row;  // TopologyAwareIdleGpus: capacity per domain [D+T], per constraint key (level row) the domains by capacity
                                                      // descending [D+T] with offsets [TL+1] and rows [TL], the preemptor's sub-groups with a required level [G+1] x 2
    int32_t *tpl_sorted, *tpl_end; uint8_t* ja_skip;  // [J] (CSR by q_job_off), [Q], [J]: every leaf's pending jobs in JobOrderFn order at the committed state (the
                                                      // simulation queues start from it), and the jobs a simulation takes out of it because their state is in flux
    int32_t *mjr_q, *mjr_job;                         // [J] MinimalJobRepresentatives: (queue or -1, representative job) per signature met so far
    double *rc_rem, *rc_ent; int32_t* rc_ent_q; uint8_t *rc_has, *rc_inv, *rc_ent_g;  // reclaimable validator scratch (rc_ent_g: the entry frees whole / shared GPUs, Resource.GPUs() > 0 — MIG instances do not count there)
    int32_t *job_head, *job_tail, *grp_link, *sc_jobs, *sc_jobs_n;  // [J+1] x 2, [P+2], [P+1], [1]: the scenario's task groups chained per job in the order they were added, and its distinct
                                                      // victim jobs in ascending index (what the validators range: kai_engine_solver.inc grp_add / scenario_jobs)
    int32_t *q_live, *q_dead, *q_stack, *q_markl, *q_live_n;  // [Q+2] x 4, [4]: the nodes of a simulation's queue (parents before children), the children left out, scratch, the marked nodes; counts
    uint8_t *q_total, *q_relc, *q_pruned;             // [Q+1] simulation queues without bystander subtrees (kai_engine_solver.inc sim_prune): the sibling order below this node is a strict total order /
                                                      // children with relevant jobs below them in this simulation / subtree left out of this simulation's queue
    // The victims log (kai_engine_solver.inc vl_*): the pops of the victims queue of the job being solved, once per pending job — every partial job walks the same sequence.
    // Entry e = the job popped, its GetTasksToEvict slice (vl_tasks[vl_off[e] .. vl_off[e+1])) and whether the job was pushed back; p_vl[p] = the entry that took pod p (INT_MAX: none yet)
    int32_t* grp_mark;  // [P+2] stamp per task group (victim_groups: first-seen order without a search per task; the running stamp is mw_ctr[3])
    uint32_t* sc_bits;  // [(J + 31) / 32 + 1] the scenario's distinct victim jobs as a bitmap (scenario_jobs puts them in ascending index from it); sc_jobs_n[2], [3] = its lowest / highest word in use
    int32_t *vl_node; double* vl_free;  // [P+1] x 2: a logged task's node and the devices its eviction frees there (AcceptedResource.GPUs(); the log is written and read at the committed state)
    int32_t *vl_job, *vl_off, *vl_tasks, *vl_canon, *p_vl; uint8_t* vl_more;  // [P+J+2], [P+J+3], [P+1], [P+1] (an entry's tasks in canonical pod order), [P], [P+J+2]
    int32_t P_cap;
};
inline size_t solver_scratch_bytes(int N, int P, int S, int J, int Q, int W, int DT = 0, int TL = 0, int G = 0) {
    size_t b = 0; auto add = [&](size_t n) { b += (n + 15) & ~size_t(15); };
    add(J); add(J); add(J); add(P); add(P); add(P); add(P);
    add(sizeof(int32_t) * P); { size_t h = 16; while (h < 2 * (size_t)P + 16) h <<= 1; add(sizeof(int64_t) * h); add(sizeof(int32_t) * h); }
    for (int i = 0; i < 2; i++) { add(sizeof(QNode) * (Q + 1)); add(sizeof(int32_t) * (Q + 2)); add(sizeof(int32_t) * (Q + 2)); add(sizeof(int32_t) * (Q + 1)); add(sizeof(int32_t) * (Q + 1)); add(sizeof(int32_t) * (J + 1)); add(sizeof(int32_t) * (Q + 1)); }
    add(sizeof(double) * 3 * Q); add(sizeof(int32_t) * (S + 1));
    for (int i = 0; i < 2; i++) { add(sizeof(int32_t) * (P + 1)); add(sizeof(int32_t) * (P + 2)); add(sizeof(int32_t) * (P + 1)); }
    add(sizeof(int32_t) * (P + 1));  // grp_ord
    add(J + 1); add(J + 1);  // vq_state, tpl_state
    add(sizeof(int32_t) * ((size_t)P + J + 2)); add(sizeof(int32_t) * ((size_t)P + J + 2)); for (int i = 0; i < 3; i++) add(sizeof(int32_t) * (J + 1)); add(sizeof(int32_t) * (P + 1)); add(sizeof(double) * 4 * (J + 1));  // mw_*
    add(sizeof(uint32_t) * (W + 1)); add(sizeof(int32_t) * (N + 1)); add(sizeof(int32_t) * (N + 1)); add(sizeof(int32_t) * KAI_MW_WAVE); add(sizeof(int32_t) * KAI_MW_WAVE); add(sizeof(int32_t) * 4);
    for (int i = 0; i < 7; i++) add(sizeof(int32_t) * (P + 1));
    add(sizeof(uint32_t) * (W + 1)); add(sizeof(uint32_t) * (W + 1));
    add(sizeof(double) * (N + 1)); add(sizeof(int32_t) * (P + 1));
    add(sizeof(QShare) * 3 * (Q + 1)); add(sizeof(int32_t) * (J + 1)); add(sizeof(int32_t) * (J + 1));
    add(sizeof(int32_t) * (J + 1)); add(sizeof(int32_t) * (Q + 1)); add(J + 1);
    add(sizeof(double) * (DT + 1)); add(sizeof(double) * (G + 1)); add(sizeof(double) * (DT + 1)); add(sizeof(int32_t) * (DT + 1)); add(sizeof(int32_t) * (TL + 2)); add(sizeof(int32_t) * (TL + 1)); add(sizeof(int32_t) * (G + 1)); add(sizeof(int32_t) * (G + 1));
    add(sizeof(double) * 3 * (Q + 1)); add(sizeof(double) * 3 * (2 * (size_t)P + J + 2)); add(sizeof(int32_t) * (2 * (size_t)P + J + 2)); add(Q + 1); add(Q + 1);
    add(Q + 2); add(Q + 2); add(Q + 2);
    add(sizeof(int32_t) * (J + 1)); add(sizeof(int32_t) * (J + 1)); add(sizeof(int32_t) * (P + 2)); add(sizeof(int32_t) * (P + 1)); add(sizeof(int32_t) * 4);
    add(2 * (size_t)P + J + 2);
    for (int i = 0; i < 4; i++) add(sizeof(int32_t) * (2 * (size_t)Q + 4)); add(sizeof(int32_t) * 4);
    add(sizeof(int32_t) * ((size_t)P + J + 2)); add(sizeof(int32_t) * ((size_t)P + J + 3)); add(sizeof(int32_t) * (P + 1)); add(sizeof(int32_t) * (P + 1)); add(sizeof(int32_t) * (P + 1)); add((size_t)P + J + 2); add(sizeof(int32_t) * (P + 1)); add(sizeof(double) * (P + 1)); add(sizeof(uint32_t) * ((size_t)J / 32 + 2)); add(sizeof(int32_t) * (P + 2));  // vl_*, sc_bits, grp_mark
    return b + 64;
}
inline void solver_scratch_bind(SolverCtx& v, char* base, int N, int P, int S, int J, int Q, int W, int DT = 0, int TL = 0, int G = 0) {
    char* c = base; auto take = [&](size_t n) { char* r = c; c += (n + 15) & ~size_t(15); return r; };
    v.vq_in = (uint8_t*)take(J); v.vq_excl = (uint8_t*)take(J); v.ja_in = (uint8_t*)take(J);
    v.p_taken = (uint8_t*)take(P); v.p_recorded = (uint8_t*)take(P); v.p_partial = (uint8_t*)take(P); v.ig_cache = (uint8_t*)take(P);
    v.p_grp = (int32_t*)take(sizeof(int32_t) * P);
    { size_t h = 16; while (h < 2 * (size_t)P + 16) h <<= 1; v.xr_key = (int64_t*)take(sizeof(int64_t) * h); v.xr_status = (int32_t*)take(sizeof(int32_t) * h); v.xr_mask = (int32_t)(h - 1); }
    for (int i = 0; i < 2; i++) {
        v.i_qn[i] = (QNode*)take(sizeof(QNode) * (Q + 1)); v.i_qheap[i] = (int32_t*)take(sizeof(int32_t) * (Q + 2)); v.i_root[i] = (int32_t*)take(sizeof(int32_t) * (Q + 2));
        v.i_cur[i] = (int32_t*)take(sizeof(int32_t) * (Q + 1)); v.i_end[i] = (int32_t*)take(sizeof(int32_t) * (Q + 1)); v.i_side[i] = (int32_t*)take(sizeof(int32_t) * (J + 1)); v.i_side_len[i] = (int32_t*)take(sizeof(int32_t) * (Q + 1));
    }
    v.vq_pop = (double*)take(sizeof(double) * 3 * Q); v.s_ov_min = (int32_t*)take(sizeof(int32_t) * (S + 1));
    v.grp_job = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.grp_off = (int32_t*)take(sizeof(int32_t) * (P + 2)); v.grp_pods = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.grp_ord = (int32_t*)take(sizeof(int32_t) * (P + 1));
    v.vq_state = (uint8_t*)take(J + 1); v.tpl_state = (uint8_t*)take(J + 1);
    v.mw_end = (int32_t*)take(sizeof(int32_t) * ((size_t)P + J + 2)); v.mw_filt = (int32_t*)take(sizeof(int32_t) * ((size_t)P + J + 2)); v.mw_sj = (int32_t*)take(sizeof(int32_t) * (J + 1)); v.mw_sv = (int32_t*)take(sizeof(int32_t) * (J + 1));
    v.mw_sn = (int32_t*)take(sizeof(int32_t) * (J + 1)); v.mw_stta = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.mw_sres = (double*)take(sizeof(double) * 4 * (J + 1));
    v.mw_feas = (uint32_t*)take(sizeof(uint32_t) * (W + 1)); v.mw_nstamp = (int32_t*)take(sizeof(int32_t) * (N + 1)); v.mw_nmaxk = (int32_t*)take(sizeof(int32_t) * (N + 1)); v.mw_wk = (int32_t*)take(sizeof(int32_t) * KAI_MW_WAVE); v.mw_wn = (int32_t*)take(sizeof(int32_t) * KAI_MW_WAVE); v.mw_ctr = (int32_t*)take(sizeof(int32_t) * 4);
    v.rec_job = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.rec_off = (int32_t*)take(sizeof(int32_t) * (P + 2)); v.rec_pods = (int32_t*)take(sizeof(int32_t) * (P + 1));
    v.res_tasks = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.ev_tasks = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.vt_tasks = (int32_t*)take(sizeof(int32_t) * (P + 1));
    v.pend = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.tmp = (int32_t*)take(sizeof(int32_t) * (P + 1));
    v.tmp2 = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.tmp3 = (int32_t*)take(sizeof(int32_t) * (P + 1));
    v.feas = (uint32_t*)take(sizeof(uint32_t) * (W + 1)); v.feas0 = (uint32_t*)take(sizeof(uint32_t) * (W + 1));
    v.ig_idle = (double*)take(sizeof(double) * (N + 1)); v.ig_sorted = (int32_t*)take(sizeof(int32_t) * (P + 1));
    v.q_sim = (QShare*)take(sizeof(QShare) * 3 * (Q + 1)); v.mjr_q = (int32_t*)take(sizeof(int32_t) * (J + 1)); v.mjr_job = (int32_t*)take(sizeof(int32_t) * (J + 1));
    v.tpl_sorted = (int32_t*)take(sizeof(int32_t) * (J + 1)); v.tpl_end = (int32_t*)take(sizeof(int32_t) * (Q + 1)); v.ja_skip = (uint8_t*)take(J + 1);
    v.ta_cap = (double*)take(sizeof(double) * (DT + 1)); v.ta_req = (double*)take(sizeof(double) * (G + 1)); v.ta_virt = (double*)take(sizeof(double) * (DT + 1)); v.ta_sorted = (int32_t*)take(sizeof(int32_t) * (DT + 1)); v.ta_off = (int32_t*)take(sizeof(int32_t) * (TL + 2));
    v.ta_row = (int32_t*)take(sizeof(int32_t) * (TL + 1)); v.ta_sg = (int32_t*)take(sizeof(int32_t) * (G + 1)); v.ta_sg_row = (int32_t*)take(sizeof(int32_t) * (G + 1));
    v.rc_rem = (double*)take(sizeof(double) * 3 * (Q + 1)); v.rc_ent = (double*)take(sizeof(double) * 3 * (2 * (size_t)P + J + 2));
    v.rc_ent_q = (int32_t*)take(sizeof(int32_t) * (2 * (size_t)P + J + 2)); v.rc_has = (uint8_t*)take(Q + 1); v.rc_inv = (uint8_t*)take(Q + 1);
    v.q_total = (uint8_t*)take(Q + 2); v.q_relc = (uint8_t*)take(Q + 2); v.q_pruned = (uint8_t*)take(Q + 2);
    v.job_head = (int32_t*)take(sizeof(int32_t) * (J + 1)); v.job_tail = (int32_t*)take(sizeof(int32_t) * (J + 1)); v.grp_link = (int32_t*)take(sizeof(int32_t) * (P + 2)); v.sc_jobs = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.sc_jobs_n = (int32_t*)take(sizeof(int32_t) * 4);
    v.rc_ent_g = (uint8_t*)take(2 * (size_t)P + J + 2);
    v.q_live = (int32_t*)take(sizeof(int32_t) * (2 * (size_t)Q + 4)); v.q_dead = (int32_t*)take(sizeof(int32_t) * (2 * (size_t)Q + 4)); v.q_stack = (int32_t*)take(sizeof(int32_t) * (2 * (size_t)Q + 4)); v.q_markl = (int32_t*)take(sizeof(int32_t) * (2 * (size_t)Q + 4)); v.q_live_n = (int32_t*)take(sizeof(int32_t) * 4);
    v.vl_job = (int32_t*)take(sizeof(int32_t) * ((size_t)P + J + 2)); v.vl_off = (int32_t*)take(sizeof(int32_t) * ((size_t)P + J + 3)); v.vl_tasks = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.vl_canon = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.p_vl = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.vl_more = (uint8_t*)take((size_t)P + J + 2); v.vl_node = (int32_t*)take(sizeof(int32_t) * (P + 1)); v.vl_free = (double*)take(sizeof(double) * (P + 1)); v.sc_bits = (uint32_t*)take(sizeof(uint32_t) * ((size_t)J / 32 + 2)); v.grp_mark = (int32_t*)take(sizeof(int32_t) * (P + 2));
    v.P_cap = P;
}

// All pointers are device memory (HBM).  [R][N] arrays are resource-major.  Node index = name rank.
struct ScanGrid;
struct KaiCtx {
    int32_t N, P, S, J, Q, R, n_pod_classes, n_node_classes;
    uint32_t plugins; int32_t gpu_strategy, cpu_strategy, restrict_nodes; double k_value;
    int32_t C, NB, NSB, use_index, all_tracked, queue_depth, fast_ok, pad1;
    // nodes
    KAI_GP(const double) n_alloc; KAI_GP(uint32_t) n_flags /* static but for the two internal summary bits of the node's shared GPUs (SgNode::refit) */; KAI_GP(const int32_t) n_gpu_count; KAI_GP(const int32_t) n_class;
    KAI_GP(double) n_idle, n_rel, n_used;
    // pods
    KAI_GP(const double) p_req; KAI_GP(const int32_t) p_job, p_podset; KAI_GP(const uint32_t) p_flags; KAI_GP(const int32_t) p_class, p_nominated, p_scls;
    KAI_GP(int32_t) p_status, p_node, p_on_node, p_on_node_status; KAI_GP(uint8_t) p_virtual, p_accepted;
    // pod-sets
    KAI_GP(const int32_t) s_job, s_min; KAI_GP(const uint32_t) s_name_rank;
    KAI_GP(int32_t) s_active_alloc, s_active_used, s_alive, s_gated, s_pipelined;
    // jobs
    KAI_GP(const int32_t) j_queue, j_prio, j_preempt; KAI_GP(const int64_t) j_created; KAI_GP(const uint32_t) j_uid_rank;
    KAI_GP(const int32_t) j_first_pod, j_n_pods, j_first_ps, j_n_ps;
    KAI_GP(const int32_t) j_pods_sorted;  // [P] each job's pods in TaskOrderFn order (session_plugins.go:244-260), same region as the job's pods
    KAI_GP(int32_t) j_n_pending, j_tta_valid, j_tta_n, tta;  // tasks-to-allocate cache (allocation_info.go:31-33), region = job's pod range
    KAI_GP(double) j_tta_res;  // [3][J] init resource of the cached chunk (CPU, Memory, GPU)
    KAI_GP(double) j_allocated;  // [3][J] PodGroupInfo.Allocated
    // queues
    KAI_GP(const int32_t) q_parent, q_prio; KAI_GP(const int64_t) q_created; KAI_GP(const uint32_t) q_uid_rank;
    KAI_GP(const int32_t) q_child_off, q_children, q_job_off, q_depth_order;  // CSR children (virtual root at Q); leaf job regions; deepest first
    KAI_GP(QShare) q_share;  // [Q][3]
    KAI_GP(QNode) qn;  // [Q] HBM home of the job-order tree (static fields + leaf lengths set by k_leaf_init)
    KAI_GP(const uint8_t) class_fit;
    // scan classes + class index
    KAI_GP(const ClassRec) cls; KAI_GP(uint64_t) sum1_key; KAI_GP(int32_t) sum1_node;  // [C][NB]
    // job-order tree (actions/utils/job_order_by_queue.go).  Leaves: a sorted region + a side heap; inner nodes: array heaps.
    KAI_GP(const int32_t) jobs_static;  // [J] CSR by q_job_off: each queue's jobs by (priority desc, creation, uid)
    KAI_GP(int32_t) lq_sorted, lq_cur, lq_end, lq_side, lq_side_len; KAI_GP(uint8_t) j_state;
    KAI_GP(int32_t) qheap, root_heap;
    // statement + committed operations
    KAI_GP(StmtOp) ops; int32_t ops_cap; KAI_GP(kai_op) out_ops; int64_t out_cap;
    KAI_GP(int32_t) scratch;  // [P] ints
    KAI_GP(EngineState) st;
    // topologies (plugins/topology) — node_domain rows are in engine node order; index D + t = root domain of topology t
    int32_t T, TL, D, G; int32_t W, pad2, pad3, pad4;               // topologies, level rows, domains, groups; words per node-set bitmap
    KAI_GP(const int32_t) topo_level_off, node_domain, dom_level /* inside its topology, -1 root */, dom_topo, dom_parent; KAI_GP(const uint32_t) dom_id_rank;
    KAI_GP(const int32_t) dom_child_off;                             // [D+T+1] CSR of children
    KAI_GP(int32_t) dom_children;                                    // [..] current child order (sortTree re-orders in place, like the reference)
    KAI_GP(int32_t) dom_alloc_pods; KAI_GP(double) dom_free;         // [D+T], [D+T][KAI_MAX_RES] AllocatablePods / IdleOrReleasingResources
    KAI_GP(int32_t) dom_tmp;                                         // [3*(D+T)+4] scratch: chosen flags, BFS queue
    KAI_GP(double) dom_ratio;                                        // [D+T] scratch of sortTree
    KAI_GP(int64_t) dom_key;                                         // [2 (D+T)] scratch of sortDomainInfos on the scan lanes: a domain's path from the root as one number
    // sub-group tree
    KAI_GP(const int32_t) g_job, g_parent; KAI_GP(const uint32_t) g_name_rank; KAI_GP(const int32_t) g_topo, g_req, g_pref, j_root_group;
    KAI_GP(const int32_t) g_child_off, g_children;                   // CSR of child groups
    KAI_GP(const int32_t) s_group, s_topo, s_req, s_pref; KAI_GP(const uint8_t) j_has_topology;
    // the general path's node sets (one bitmap per DFS level), set lists and preferred-level scores of the current job
    KAI_GP(uint32_t) ns_bits;                                        // [KAI_TDEPTH][W]
    KAI_GP(int32_t) ns_sets;                                         // [KAI_TDEPTH][D+T+1] domains of the node sets of a frame
    KAI_GP(double) sg_score;                                         // [KAI_TKEYS][D+T] node score of a domain at the key's preferred level, <0 = not scored
    KAI_GP(int32_t) sg_key, sg_row;                                  // [KAI_TKEYS] sub-group key and its preferred level row
    // victim actions
    int32_t action, max_consolidation_preemptees, allow_consolidating_reclaim, use_signatures; double saturation_multiplier;
    KAI_GP(const int64_t) j_signature;  // [J] scheduling-constraints signature id (null when the snapshot has none)
    // minruntime plugin inputs (null = nothing is protected)
    KAI_GP(const int64_t) j_last_start, q_preempt_mr, q_reclaim_mr; int64_t now_ns, def_preempt_mr, def_reclaim_mr; int32_t reclaim_method, pad8;
    SolverCtx sv;
    BatchCtx bt;  // batch path of the allocate action (kai_batch.hpp)
#ifdef KAI_SHARED_GPUS
    // shared GPUs (ABI v4; compiled into the host twin only until the device path is verified on the MI355X): fractions of one device
    KAI_GP(const double) p_portion;       // [P] 0 = a whole-GPU / CPU-only pod
    // derived per pod on the host (HostPrep::SharedPods; one GPU memory size M for the whole cluster):
    KAI_GP(const uint8_t) p_shared;       // [P] IsSharedGPURequest: a fraction of one device, or MiB of one device (pod_info.go:332-334)
    KAI_GP(const int64_t) p_mem;          // [P] NodeInfo.GetResourceGpuMemory of the request (node_info.go:653-659): the MiB it takes on its device
    KAI_GP(const int64_t) p_gmem;         // [P] ResReq.GpuMemory(): > 0 = a gpu-memory request
    KAI_GP(const double) p_acc_gpu;       // [P] AcceptedResource.GetGpusQuota() once a node holds the pod (node_info.go:746-766)
    KAI_GP(const double) p_pend_gpu;      // [P] GPU weight while pending: ResReq.GPUs() + memory / MinNodeGPUMemory (proportion.go:360-366, allocation_info.go:103-107)
    KAI_GP(const double) p_quota_gpu;     // [P] ResReq.GetGpusQuota(): GPUs() + MIG instances by weight (gpu_resource_requirment.go:163-178)
    KAI_GP(const double) p_mig_q;         // [P] the MIG part of the three quantities above (AcceptedResource.GPUs() = p_acc_gpu - p_mig_q)
    KAI_GP(const uint8_t) p_kind;         // [P] SharedPods::K_*: MIG candidate / legacy MIG task / regular GPU request / GPUs() > 0
    int32_t quota_on, mig_on;             // the per-pod arrays above are valid (shared GPUs or MIG rows in the snapshot) / some resource row is a MIG profile
    int32_t res_mig_g[KAI_MAX_RES]; int64_t res_mig_m[KAI_MAX_RES];  // per resource row: GPU weight and memory of a MIG profile, 0 otherwise (mig.go:13-33)
    KAI_GP(int32_t) p_group, p_on_group;  // [P] PodInfo.GPUGroups[0]; the group in the node's own copy of the pod (node_info.go:397-398)
    KAI_GP(const int64_t) n_gpu_mem;      // [N] MemoryOfEveryGpuOnNode
    KAI_GP(int32_t) ng_id;                // [N][KAI_GMAX] GpuSharingNodeInfo: group id of the slot, -1 = free
    KAI_GP(int64_t) ng_used, ng_rel, ng_alloc;  // [N][KAI_GMAX] UsedSharedGPUsMemory / ReleasingSharedGPUsMemory / AllocatedSharedGPUsMemory
    KAI_GP(uint32_t) ng_mark, ng_has_alloc;     // [N] bit per slot: ReleasingSharedGPUs; the group has an entry in AllocatedSharedGPUsMemory
    KAI_GP(int32_t) next_new_group;       // [1] uuid.NewUUID() of findGpuForSharingOnNode
    int32_t shared_on, pad_sh;
#endif
    struct ScanGrid* sg; int32_t sg_wgs, sg_per;  // allocate action on the sequential engine: node scans spread over sg_wgs workgroups (0 / 1: this workgroup only), sg_per nodes each (kai_kernels.hpp)
    MultiCtx* mw; int32_t mw_rank, mw_world;  // victim search on several workgroups: the block they share (null / world 1 = one workgroup), this replica's rank
    int32_t mw_xworld, mw_xrank, mw_xcap, mw_xpad;  // ... and over the GPUs of a node-sharded group: ranks of the group (0 / 1 = this GPU alone), this GPU's rank, simulations of one wave at most
    XMail* mw_mail;                    // the mailbox the exchange of a wave's outcomes goes through on the device (pinned host memory, kai_victim_shard.hpp)
    int32_t exact_sums, pad_es;  // HostPrep::exact_sums: integral quantities with totals below 2^52 units — parallel sums of them are exact
};

// ======================================================================================================
// per-node arithmetic shared by every scanner
// ======================================================================================================
#ifndef KAI_DOM_LANES_MIN
#define KAI_DOM_LANES_MIN 16  // domains of a topology from which the loops over them go to the scan lanes (tests/host_sim overrides it to cover the path on small trees)
#endif
#ifdef KAI_PROF_VICTIM
#define KAI_TCLK(slot) { int64_t tnow_ = be.clock(); el().h.prof[slot] += tnow_ - tclk_; tclk_ = tnow_; }
#define KAI_TCLK0 int64_t tclk_ = be.clock();
#else
#define KAI_TCLK(slot)
#define KAI_TCLK0
#endif
// The node loops of the topology plugin's subSetNodesFn (plugins/topology/job_filtering.go) as one request to the backend's scan lanes: Backend::topo_scan
// returns false when it has none (the engine then walks the nodes itself).
constexpr int KAI_TOPO_SCAN_LEVELS = 8;
struct TopoScan {
    int32_t op;          // 1 = per level the smallest / largest domain id over the nodes of `parent` that are part of the topology (lowestCommonDomainID, common.go:17-67)
                         // 2 = Idle + Releasing of the nodes of `domain` summed into their leaf domains (calcSubTreeFreeResources :192-211; integral quantities: the order
                         //     of addition does not show)   3 = pods of the maximal request every node of `domain` takes, summed per leaf domain (calcNodeAccommodation :213-246)
    int32_t row0, L, domain, root, dl, R, tasks, one_pod, any;
    // ops over the DOMAINS of topology `topo` (Engine::topo_dom_body, one domain per lane step): 5 = treeAllocatableCleanup (:438-445)   6 = one level of the bottom-up roll-up inside the
    // sub-tree of `domain` (what & 1: IdleOrReleasingResources, what & 2: AllocatablePods) — level `lvl` into its parents   7 = AllocatablePods := 0 inside the sub-tree
    // 8 = getJobRatioToFreeResources of every domain (sortTree's keys)   9 = checkJobDomainFit of the domains whose level is in `what` (bit l + 1) → chosen flags, `any` = their number
    // 10 / 11 = sortTreeFromRoot: every child's place among its siblings, the new order back   12 / 13 / 14 = sortDomainInfos: positions, paths as numbers, the chosen domains in order → out
    // 15 = the SURVEY: ops 1, 2 and 3 in ONE pass over the nodes (level minima / maxima over `parent`; Idle + Releasing and — what & 2 — the pod counts of EVERY node of the
    // topology into its leaf domain, counts in dom_tmp[2 DT ..])   16 = what the survey gathered outside the sub-tree of `domain` is dropped, the counts inside become AllocatablePods
    int32_t topo, lvl, what, pad_t;
    double tr[KAI_MAX_RES];  // summed request of the sub-group's tasks (ops 8, 9)
    double mx[KAI_MAX_RES];
    KAI_GP(const uint32_t) parent;
    KAI_GP(uint32_t) out;  // op 4: the node-set bitmap parent ∩ nodes of `domain` (domain < 0: the parent set itself; dl < 0: every node of the topology) — build_node_set
    int32_t lvl_min[KAI_TOPO_SCAN_LEVELS], lvl_max[KAI_TOPO_SCAN_LEVELS];  // results of op 1 (any = some node qualified)
};
// Index loops of the victim search (resets, the victims-queue filter, feasible nodes, idle GPUs per node) as one request to the backend's scan lanes:
// Backend::pfor returns false when it has none (the engine then runs the loop itself); the bodies are Engine::pfor_body (kai_engine_solver.inc).
enum PforOp : int32_t { PFO_PARTIAL_RESET = 1, PFO_VICTIM_FILTER = 2, PFO_FEASIBLE = 3, PFO_IG_IDLE = 4, PFO_TPL_STATE = 5 };
struct PforReq { int32_t op, n, a, b; };
KAI_HD bool topo_node_in_domain(const KaiCtx& c, const TopoScan& t, int n) {
    if (t.L <= 0) return false;
    return t.domain == t.root ? c.node_domain[(size_t)t.row0 * c.N + n] >= 0 : c.node_domain[(size_t)(t.row0 + t.dl) * c.N + n] == t.domain;
}
// calcNodeAccommodation (plugins/topology/job_filtering.go:213-246) against av = Idle + Releasing of one node: how many pods of the maximal request fit, the k-th test pod
// being k x that request by repeated addition (the comparisons of fits(), on values held in registers)
KAI_HD int topo_node_count(const TopoScan& t, const double* av) {
    if (t.one_pod) return t.tasks;
    double cur[KAI_MAX_RES]; for (int r = 0; r < KAI_MAX_RES; r++) cur[r] = t.mx[r];
    int count = 0;
    for (;;) {
        bool ok = true;
        for (int r = 0; r < KAI_MAX_RES; r++) if (r < t.R) { const double rq = cur[r]; if (r >= KAI_RES_PODS && !(rq > 0)) continue; if (rq > av[r]) ok = false; }
        if (!ok) break;
        count++;
        for (int r = 0; r < KAI_MAX_RES; r++) if (r < t.R) cur[r] += t.mx[r];
    }
    return count;
}
struct ScanReq {
    int32_t pod, cpu_only, best_effort, pod_class, nominated, r_place, strategy, pad;
    double req[KAI_MAX_RES];
    double min_a, max_a;  // nodeplacement.setBinpackPreOrder range (plugins/nodeplacement/pack.go:35-43)
#ifdef KAI_SHARED_GPUS
    double portion;       // > 0: the task asks for this fraction of one device
    int64_t gmem;         // >= 0: GetResourceGpuMemory of a shared request (fraction or gpu-memory), -1: derive it from the portion
    int32_t shared, kind;   // IsSharedGPURequest; SharedPods::K_* of the pod (MIG candidate / legacy MIG / regular / GPUs() > 0)
#endif
};

KAI_HD bool fits(const KaiCtx& c, const double* req, int n, bool with_releasing) {
    // ResourceRequirements.LessEqualResource (api/resource_info/resource_requirment.go:126-140) against Idle, or against
    // NodeInfo.NonAllocatedResources = 0 + Idle + Releasing (api/node_info/node_info.go:157-162)
    for (int r = 0; r < c.R; r++) {
        double rq = req[r];
        if (r >= KAI_RES_PODS && !(rq > 0)) continue;  // scalar keys exist only for non-zero requests
        double avail = c.n_idle[(size_t)r * c.N + n];
        if (with_releasing) avail = avail + c.n_rel[(size_t)r * c.N + n];
        if (rq > avail) return false;
    }
    return true;
}

#ifdef KAI_SHARED_GPUS
// ------------------------------------------------------------------------------------------------------
// Shared GPUs: api/node_info/gpu_sharing_node_info.go on fixed tables — KAI_GMAX group slots per node.  A Go map entry exists once it was
// written, whatever its value; a slot is that entry for the three maps together (ng_has_alloc tells whether AllocatedSharedGPUsMemory has
// the key).  Maps are ranged in ascending group id (the oracle's canonical order).
// ------------------------------------------------------------------------------------------------------
constexpr uint32_t KAI_NODE_SHARE_FIT0_I = 0x10000000u, KAI_NODE_SHARE_FIT1_I = 0x20000000u;  // internal, dynamic (SgNode::refit): some used shared GPU of the node can take a task of 0 MiB / of a whole
                                                                                                  // device's memory — the gpusharingorder score of the classes that ask for no fraction (class_key_regs)
constexpr int KAI_GMAX = 32;  // (the slot masks ng_mark / ng_has_alloc are 32 bits; 16 slots overflowed once in 6·10^5 campaign cycles: entries that rolled-back simulations leave
                              // booked — the reference's own residue, gpu_sharing_node_info.go:102-105 — cannot be taken over)
constexpr int KAI_NEW_GROUP = 1 << 20;  // ids from here on: non-numeric group names (UUIDs) — "new" for predicates.go:320-330
constexpr int KAI_WHOLE_GPU = -1;       // pod_info.WholeGpuIndicator
struct SgNode {
    const KaiCtx& c; int n;
    KAI_HD int32_t& id(int s) const { return c.ng_id[(size_t)n * KAI_GMAX + s]; }
    KAI_HD int64_t& used(int s) const { return c.ng_used[(size_t)n * KAI_GMAX + s]; }
    KAI_HD int64_t& rel(int s) const { return c.ng_rel[(size_t)n * KAI_GMAX + s]; }
    KAI_HD int64_t& alloc(int s) const { return c.ng_alloc[(size_t)n * KAI_GMAX + s]; }
    KAI_HD double& idle_gpu() const { return c.n_idle[(size_t)KAI_RES_GPU * c.N + n]; }
    KAI_HD double& rel_gpu() const { return c.n_rel[(size_t)KAI_RES_GPU * c.N + n]; }
    KAI_HD double used_gpu() const { return c.n_used[(size_t)KAI_RES_GPU * c.N + n]; }
    KAI_HD int64_t gpu_mem() const { return c.n_gpu_mem[n]; }
    KAI_HD int64_t mem_of(double portion) const { return (int64_t)(portion * (double)gpu_mem()); }  // GetResourceGpuMemory (node_info.go:653-659)
    KAI_HD double frac_of(int64_t mem) const { double x = (double)mem / (double)gpu_mem() * 100; double f = (double)(int64_t)x; if (f < x) f += 1; return f / 100; }  // getGpuMemoryFractionalOnNode :329-332 (x >= 0)
    KAI_HD int find(int g) const { for (int s = 0; s < KAI_GMAX; s++) if (id(s) == g) return s; return -1; }
    KAI_HD int slot(int g) const {  // the map entry of group g, created on first write; -1 = table full
        int s = find(g); if (s >= 0) return s;
        for (s = 0; s < KAI_GMAX; s++) if (id(s) < 0) { id(s) = g; used(s) = 0; rel(s) = 0; alloc(s) = 0; c.ng_mark[n] &= ~(1u << s); c.ng_has_alloc[n] &= ~(1u << s); return s; }
        // table full: take over the entry of a group that was opened inside a rolled-back simulation and is empty again.  Its id came from the
        // new-group counter, so nothing can name it any more, and an all-zero unmarked entry takes part in no sum, fit or count of the node.
        for (s = 0; s < KAI_GMAX; s++) if (id(s) >= KAI_NEW_GROUP && used(s) == 0 && rel(s) == 0 && alloc(s) == 0 && !marked(s)) { id(s) = g; c.ng_has_alloc[n] &= ~(1u << s); return s; }
        return -1;
    }
    KAI_HD bool marked(int s) const { return (c.ng_mark[n] >> s) & 1u; }
    KAI_HD int64_t n_gpus() const { int lbl = c.n_gpu_count[n]; return lbl >= 0 ? lbl : (int64_t)c.n_alloc[(size_t)KAI_RES_GPU * c.N + n]; }  // GetNumberOfGPUsInNode
    KAI_HD int used_shared() const { int k = 0; for (int s = 0; s < KAI_GMAX; s++) if (id(s) >= 0 && used(s) > 0) k++; return k; }     // :265-273
    KAI_HD int used_gpus() const { return (int)used_gpu() + used_shared(); }                                                            // :275-277
    KAI_HD bool releasing_from_shared(int s) const { return used(s) != 0 && rel(s) == used(s); }                                        // :253-263 (a found key with value 0 != used)
    KAI_HD bool fit_on_group(int s, int64_t mem) const { return used(s) != 0 && gpu_mem() - alloc(s) + rel(s) - mem >= 0 && alloc(s) != rel(s); }  // IsTaskFitOnGpuGroup :350-354
    KAI_HD bool enough_idle(int s, int64_t mem) const { return ((c.ng_has_alloc[n] >> s) & 1u) && gpu_mem() - alloc(s) - mem >= 0; }              // EnoughIdleResourcesOnGpu :356-363
    // the node's summary bits for the class keys: kept in n_flags by everything that changes a group (node accounting at session open, node_apply), read by load_node
    KAI_HD void refit() const {
        uint32_t f = 0;
        for (int s = 0; s < KAI_GMAX; s++) if (id(s) >= 0) { if (fit_on_group(s, 0)) f |= KAI_NODE_SHARE_FIT0_I; if (fit_on_group(s, mem_of(1.0))) f |= KAI_NODE_SHARE_FIT1_I; }
        c.n_flags[n] = (c.n_flags[n] & ~(KAI_NODE_SHARE_FIT0_I | KAI_NODE_SHARE_FIT1_I)) | f;
    }
    KAI_HD int count_fit(int64_t mem) const { for (int s = 0; s < KAI_GMAX; s++) if (id(s) >= 0 && fit_on_group(s, mem)) return 1; return 0; }  // fractionTaskGpusAllocatableDeviceCount: stops at the device count of the task, 1
    // addSharedTaskResourcesPerPodGroup :83-136 / removeSharedTaskResourcesPerPodGroup :153-235; false = table full
    KAI_HD bool add(int status, int64_t mem, int g) const {
        int s = slot(g); if (s < 0) return false;
        used(s) += mem;
        if (status == KAI_POD_RELEASING) {
            rel(s) += mem; alloc(s) += mem; c.ng_has_alloc[n] |= 1u << s;
            if (used(s) == rel(s)) {
                if (!marked(s)) { rel_gpu() += 1; c.ng_mark[n] |= 1u << s; }
                if ((int)n_gpus() < (int)idle_gpu() + used_gpus()) idle_gpu() -= 1;
            }
        } else if (status == KAI_POD_PIPELINED) {
            rel(s) -= mem;
            if (used(s) - mem == rel(s) + mem) rel_gpu() -= 1;
        } else {
            alloc(s) += mem; c.ng_has_alloc[n] |= 1u << s;
            if (used(s) <= mem) { if ((int)n_gpus() < (int)idle_gpu() + used_gpus()) idle_gpu() -= 1; }
            if (marked(s)) { rel_gpu() -= 1; c.ng_mark[n] &= ~(1u << s); }
        }
        return true;
    }
    KAI_HD bool remove(int status, int64_t mem, int g) const {
        int s = slot(g); if (s < 0) return false;
        used(s) -= mem;
        if (status == KAI_POD_RELEASING) {
            rel(s) -= mem; alloc(s) -= mem; c.ng_has_alloc[n] |= 1u << s;
            if (used(s) <= 0) {
                if ((int)n_gpus() >= (int)idle_gpu() + used_gpus()) idle_gpu() += 1;
                if (marked(s)) { rel_gpu() -= 1; c.ng_mark[n] &= ~(1u << s); }
            }
        } else if (status == KAI_POD_PIPELINED) {
            rel(s) += mem;
            const bool to_releasing = (used(s) + mem == rel(s) - mem) || (used(s) == 0 && rel(s) == 0);  // isPipelinedToReleasingGpu :237-245
            if (to_releasing) rel_gpu() += 1;
        } else {
            alloc(s) -= mem; c.ng_has_alloc[n] |= 1u << s;
            if (used(s) <= 0) { if ((int)n_gpus() >= (int)idle_gpu() + used_gpus()) idle_gpu() += 1; }
            if (releasing_from_shared(s) && !marked(s)) { rel_gpu() += 1; c.ng_mark[n] |= 1u << s; }
        }
        return true;
    }
    KAI_HD int next_slot_by_id(int prev_id) const { int best = -1; for (int s = 0; s < KAI_GMAX; s++) if (id(s) > prev_id && (best < 0 || id(s) < id(best))) best = s; return best; }  // ranges the maps in ascending group id
    KAI_HD double sum_available_shared() const {  // getSumOfAvailableSharedGPUs :289-300
        double sum = 0; for (int s = next_slot_by_id(-1); s >= 0; s = next_slot_by_id(id(s))) if (((c.ng_has_alloc[n] >> s) & 1u) && alloc(s) > 0) sum += 1 - frac_of(alloc(s)); return sum;
    }
    KAI_HD double sum_releasing_shared() const {  // getSumOfReleasingSharedGPUs :302-313
        double sum = 0; for (int s = next_slot_by_id(-1); s >= 0; s = next_slot_by_id(id(s))) if (rel(s) > 0 && !releasing_from_shared(s)) sum += frac_of(rel(s)); return sum;
    }
    KAI_HD int64_t mem_available_shared() const { int64_t m = 0; for (int s = 0; s < KAI_GMAX; s++) if (id(s) >= 0 && ((c.ng_has_alloc[n] >> s) & 1u) && alloc(s) > 0) m += gpu_mem() - alloc(s); return m; }  // second result of :314-326
    KAI_HD int64_t mem_releasing_shared() const { int64_t m = 0; for (int s = 0; s < KAI_GMAX; s++) if (id(s) >= 0 && rel(s) > 0 && !releasing_from_shared(s)) m += rel(s); return m; }       // … of :328-339
    // FittingGPUs (framework/session.go:163-199): groups that can take the task in ascending id, then one entry per idle-or-releasing whole GPU,
    // stably ordered by the GPU order score (gpupack: used portion; gpuspread: 1 - used portion, 1 for a whole GPU).  out[] holds slots, or
    // KAI_WHOLE_GPU entries; returns the count (at most KAI_GMAX + whole GPUs, capped).
    KAI_HD double gpu_score(int s) const {
        double sc = 0, usedp = s == KAI_WHOLE_GPU ? 0.0 : (double)used(s) / (double)gpu_mem();
        if (c.plugins & KAI_PLUGIN_GPUPACK) sc += s == KAI_WHOLE_GPU ? 0.0 : usedp;
        if (c.plugins & KAI_PLUGIN_GPUSPREAD) sc += s == KAI_WHOLE_GPU ? 1.0 : 1 - usedp;
        return sc;
    }
    // GetNodePreferableGpuForSharing (gpu_sharing/gpuSharing.go:39-71) for one device: the first entry of FittingGPUs.  Returns false when none;
    // slot = the group's slot or KAI_WHOLE_GPU (a new group), releasing = the task has to be pipelined there
    KAI_HD bool preferable(int64_t mem, bool pipeline_only, bool allocatable_now, int& slot_out, bool& releasing) const {
        int best = -2; double bs = 0; int best_id = 0;
        for (int s = 0; s < KAI_GMAX; s++) {  // stable order: ascending group id among equal scores; groups come before whole GPUs
            if (id(s) < 0 || !fit_on_group(s, mem)) continue;
            double sc = gpu_score(s);
            if (best == -2 || sc > bs || (sc == bs && id(s) < best_id)) { best = s; bs = sc; best_id = id(s); }
        }
        const int whole = (idle_gpu() > 0 || rel_gpu() > 0) ? (int)idle_gpu() + (int)rel_gpu() : 0;
        if (whole > 0) { double sc = gpu_score(KAI_WHOLE_GPU); if (best == -2 || sc > bs) { best = KAI_WHOLE_GPU; bs = sc; } }
        if (best == -2) return false;
        slot_out = best;
        if (best == KAI_WHOLE_GPU) releasing = pipeline_only ? true : !allocatable_now;              // findGpuForSharingOnNode :73-83
        else releasing = !enough_idle(best, mem) || !allocatable_now;
        return true;
    }
};
// isTaskAllocatableOnNonAllocatedResources for a fraction of one device (node_info.go:361-382)
KAI_HD int64_t req_gpu_mem(const SgNode& g, const ScanReq& q) { return q.gmem >= 0 ? q.gmem : g.mem_of(q.portion); }  // NodeInfo.GetResourceGpuMemory (node_info.go:653-659)
KAI_HD bool fits_shared(const KaiCtx& c, const ScanReq& q, int n, bool with_releasing) {
    for (int r = 0; r < c.R; r++) {
        if (r == KAI_RES_GPU) continue;
        double rq = q.req[r];
        if (r >= KAI_RES_PODS && !(rq > 0)) continue;
        double avail = c.n_idle[(size_t)r * c.N + n];
        if (with_releasing) avail = avail + c.n_rel[(size_t)r * c.N + n];
        if (rq > avail) return false;
    }
    SgNode g{c, n};
    double ag = g.idle_gpu(); if (with_releasing) ag = ag + g.rel_gpu();
    double fl = (double)(int64_t)ag; if (fl > ag) fl -= 1;  // math.Floor
    return (int64_t)fl + g.count_fit(req_gpu_mem(g, q)) >= 1;  // (isValidGpuPortion :668-671: the host admits gpu-memory requests of at most one device)
}
#endif

// plugins/predicates/predicates.go:173-262 minus the queue-capacity step (node independent, done by the control lane)
constexpr uint32_t KAI_NODE_LEGACY_MIG_I = 0x40000000u;  // internal: the node holds a legacy MIG task (NodeInfo.LegacyMIGTasks is not empty; set by the host at session open)
constexpr int KAI_KIND_MIG = 1, KAI_KIND_LEGACY = 2, KAI_KIND_REGULAR = 4, KAI_KIND_GPUS = 8;  // = SharedPods::K_*
KAI_HD bool node_predicates(const KaiCtx& c, bool cpu_only, int pod_class, int n, int kind = KAI_KIND_REGULAR | KAI_KIND_GPUS) {
    if (!(c.plugins & KAI_PLUGIN_PREDICATES)) return true;
    uint32_t f = c.n_flags[n];
    if (kind & KAI_KIND_LEGACY) return false;  // NodeInfo.PredicateByNodeResourcesType (api/node_info/node_info.go:315-359): "Legacy MIG jobs cannot be scheduled"
    if (!cpu_only) {
        if ((kind & KAI_KIND_GPUS) && (f & KAI_NODE_HAS_DRA_GPUS)) return false;
        const bool mig_node = f & KAI_NODE_MIG_ENABLED, mig_task = kind & KAI_KIND_MIG;
        if (!mig_node && mig_task) return false;                                   // :336-340
        if (mig_node) {
            if (mig_task && (f & KAI_NODE_LEGACY_MIG_I)) return false;             // :342-346
            if ((f & KAI_NODE_MIG_SINGLE) && !(kind & KAI_KIND_REGULAR)) return false;  // :349-352
            if ((f & KAI_NODE_MIG_MIXED) && !mig_task) return false;               // :353-356
        }
    }
    double pods = c.n_idle[(size_t)KAI_RES_PODS * c.N + n] + c.n_rel[(size_t)KAI_RES_PODS * c.N + n];  // :264-285
    if (!(pods > 0)) return false;
    if (f & KAI_NODE_NOT_READY) return false;
    if (!c.class_fit[(size_t)pod_class * c.n_node_classes + c.n_class[n]]) return false;
    if (c.restrict_nodes) {
        if (!cpu_only) { if (!(f & KAI_NODE_GPU_WORKER)) return false; }
        else if (!(f & KAI_NODE_CPU_WORKER)) return false;
    }
    return true;
}
#ifdef KAI_SHARED_GPUS
// the same for a task that asks for a fraction of one device: MigStrategy single admits whole-GPU tasks only (node_info.go:349-352) and the pod
// count check becomes checkMaxPodsWithGpuGroupReservation + willCreateNewGpuGroup (plugins/predicates/predicates.go:264-330): two free pod slots
// when the task would open a new GPU group (the reservation pod), none asked for when it joins a group of the snapshot
KAI_HD bool node_predicates_shared(const KaiCtx& c, const ScanReq& q, int n) {
    if (!(c.plugins & KAI_PLUGIN_PREDICATES)) return true;
    uint32_t f = c.n_flags[n];
    if (f & KAI_NODE_HAS_DRA_GPUS) return false;
    if ((f & KAI_NODE_MIG_ENABLED) && (f & (KAI_NODE_MIG_MIXED | KAI_NODE_MIG_SINGLE))) return false;
    SgNode g{c, n}; const int64_t mem = req_gpu_mem(g, q);
    const bool alloc_now = q.best_effort || fits_shared(c, q, n, false);
    int slot = 0; bool releasing = false; bool needs_new = true;
    if (g.preferable(mem, false, alloc_now, slot, releasing)) needs_new = slot == KAI_WHOLE_GPU || g.id(slot) >= KAI_NEW_GROUP;
    double pods = c.n_idle[(size_t)KAI_RES_PODS * c.N + n] + c.n_rel[(size_t)KAI_RES_PODS * c.N + n];
    if (needs_new && pods < 2) return false;
    if (f & KAI_NODE_NOT_READY) return false;
    if (!c.class_fit[(size_t)q.pod_class * c.n_node_classes + c.n_class[n]]) return false;
    if (c.restrict_nodes && !(f & KAI_NODE_GPU_WORKER)) return false;
    return true;
}
#endif
KAI_HD bool cpu_only_node(const KaiCtx& c, int n) {  // node_info.go:697-702
    uint32_t f = c.n_flags[n];
    return !(f & KAI_NODE_MIG_ENABLED) && c.n_alloc[(size_t)KAI_RES_GPU * c.N + n] <= 0 && !(f & KAI_NODE_HAS_DRA_GPUS);
}

// Σ NodeOrderFns in registration order (framework/session_plugins.go:427-437, conf_util/scheduler_conf_util.go:39-60)
KAI_HD double node_score(const KaiCtx& c, const ScanReq& q, int n, bool fit_idle) {
    double score = 0.0;
    if (c.plugins & KAI_PLUGIN_NODEAVAILABILITY) score += fit_idle ? 100.0 : 0.0;  // plugins/nodeavailability/nodeavailability.go:29-40
#ifdef KAI_SHARED_GPUS
    if (c.shared_on && (c.plugins & KAI_PLUGIN_GPUSHARINGORDER)) {  // plugins/gpusharingorder/gpusharingorder.go:29-44: 1000 when a used shared GPU of the node can take the task
        SgNode g{c, n}; const int64_t mem = req_gpu_mem(g, q);  // GetResourceGpuMemory: portion 1 for whole GPUs, 0 for a CPU-only task (which therefore "fits" every active shared GPU)
        double sc = 0.0; for (int s2 = 0; s2 < KAI_GMAX; s2++) if (g.id(s2) >= 0 && g.fit_on_group(s2, mem)) sc = 1000.0;
        score += sc;
    }
#endif
    if (c.plugins & KAI_PLUGIN_RESOURCETYPE) score += (q.cpu_only && cpu_only_node(c, n)) ? 10.0 : 0.0;  // plugins/resourcetype/resourcetype.go:29-41
    if (c.plugins & KAI_PLUGIN_NOMINATEDNODE) score += (q.nominated >= 0 && q.nominated == n) ? 1000000.0 : 0.0;
    if (c.plugins & KAI_PLUGIN_NODEPLACEMENT) {
        int r = q.r_place;
        double cur = c.n_idle[(size_t)r * c.N + n] + c.n_rel[(size_t)r * c.N + n];
        double overall = c.n_alloc[(size_t)r * c.N + n];
        double place;
        if (q.strategy == KAI_SPREAD) {  // plugins/nodeplacement/spread.go:16-36
            double count = overall;
            if (r == KAI_RES_GPU) { int lbl = c.n_gpu_count[n]; count = lbl >= 0 ? (double)lbl : (double)(int64_t)overall; }
            place = count == 0 ? 0.0 : cur / count;
        } else {  // plugins/nodeplacement/pack.go:45-64
            if (overall == 0) place = 0.0;
            else if (q.max_a == 0) place = 0.0;
            else if (q.min_a == q.max_a) place = 9.0;
            else place = 9.0 * (1 - (cur - q.min_a) / (q.max_a - q.min_a));
        }
        score += place;
    }
    return score;
}
// What a brute-force decision over ALL nodes evaluates per node (no node set, no topology scores): is the node a candidate, and its score.  The passes of the scan
// lanes, the single-node re-evaluations of Engine::best_node_kept and tests/host_sim share it.
KAI_HD bool scan_node_score(const KaiCtx& c, const ScanReq& q, int n, double& sc) {
#ifdef KAI_SHARED_GPUS
    const bool frac = c.shared_on && q.shared;  // a fraction (or MiB) of one device: fit / predicates over the node's GPU groups
    if (!(frac ? fits_shared(c, q, n, true) : fits(c, q.req, n, true))) return false;                                      // IsTaskAllocatableOnReleasingOrIdle
    if (!(frac ? node_predicates_shared(c, q, n) : node_predicates(c, q.cpu_only != 0, q.pod_class, n, q.kind))) return false;  // ssn.PredicateFn
    const bool fit_idle = q.best_effort || (frac ? fits_shared(c, q, n, false) : fits(c, q.req, n, false));
#else
    if (!fits(c, q.req, n, true)) return false;
    if (!node_predicates(c, q.cpu_only != 0, q.pod_class, n)) return false;
    const bool fit_idle = q.best_effort || fits(c, q.req, n, false);
#endif
    sc = node_score(c, q, n, fit_idle);
    return true;
}

// The class key: 0 = the node does not pass FittingNode for this class; otherwise a 64-bit value whose order over nodes is the
// order of the reference's f64 score sum for every task of the class:
//   bit 63  nodeavailability (+100): the task fits on Idle alone           (dominates 10 + 9)
//   bit 62  resourcetype (+10): CPU-only task on a CPU-only node           (dominates 9)
//   bits 0-61  nodeplacement: bin-pack → 2^53-1 - (Idle+Releasing)[r] (the pack score 9·(1-(cur-min)/(max-min)) is strictly
//              decreasing in cur for the integer quantities the host admits to the index, kai_host_prep.hpp guards);
//              spread → bit pattern of the f64 score cur/count ∈ [0,1], plus one.
// The nominated-node bonus (+1e6) concerns one node per pod and is handled by the control lane.
// Everything class_key reads of one node, fetched in one batch of independent loads (the refresh path is latency bound).
struct NodeRegs {
    double idle[KAI_MAX_RES], rel[KAI_MAX_RES], alloc_cpu, alloc_gpu;
    uint32_t flags; int32_t ncls, gpu_count;
    uint32_t share_fit;  // shared GPUs: bit 0 / bit 1 = some used shared GPU of the node can take a task of 0 MiB (a CPU-only task) / of a whole device's memory (gpusharingorder)
};
#ifdef KAI_SHARED_GPUS
// The gpusharingorder score of a class that asks for no fraction (plugins/gpusharingorder/gpusharingorder.go:29-44 with GetResourceGpuMemory of portion 0 or 1):
// a layout of the key with one more bit above the others (1000 dominates 100 + 10 + 9), used whenever the session has shared GPUs and the plugin.
KAI_HD bool key_shared_layout(const KaiCtx& c) { return c.shared_on && (c.plugins & KAI_PLUGIN_GPUSHARINGORDER); }
#else
KAI_HD bool key_shared_layout(const KaiCtx&) { return false; }
#endif
KAI_HD int key_avail_bit(const KaiCtx& c) { return key_shared_layout(c) ? 62 : 63; }
KAI_HD void load_node(const KaiCtx& c, int n, NodeRegs& s) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < KAI_MAX_RES; r++) {
        bool on = r < c.R;
        s.idle[r] = on ? c.n_idle[(size_t)r * c.N + n] : 0.0;
        s.rel[r] = on ? c.n_rel[(size_t)r * c.N + n] : 0.0;
    }
    s.alloc_cpu = c.n_alloc[(size_t)KAI_RES_CPU * c.N + n]; s.alloc_gpu = c.n_alloc[(size_t)KAI_RES_GPU * c.N + n];
    s.flags = c.n_flags[n]; s.ncls = c.n_class[n]; s.gpu_count = c.n_gpu_count[n];
    s.share_fit = 0;
#ifdef KAI_SHARED_GPUS
    if (key_shared_layout(c)) s.share_fit = (s.flags >> 28) & 3u;  // KAI_NODE_SHARE_FIT0_I / FIT1_I, kept by SgNode::refit
#endif
}
KAI_HD uint64_t class_key_regs(const KaiCtx& c, const ClassRec& k, const NodeRegs& s) {
    // FittingNode: fit on Idle+Releasing (api/node_info/node_info.go:190-206, 361-382) …
    bool fit_rel = true, fit_idle = true;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < KAI_MAX_RES; r++) {
        if (r >= c.R) continue;
        double rq = k.req[r];
        if (r >= KAI_RES_PODS && !(rq > 0)) continue;  // scalar keys exist only for non-zero requests
        if (rq > s.idle[r] + s.rel[r]) fit_rel = false;
        if (rq > s.idle[r]) fit_idle = false;
    }
    if (!fit_rel) return 0;
    // … and the node-dependent predicates (plugins/predicates/predicates.go:173-262)
    const bool cpu_node = !(s.flags & KAI_NODE_MIG_ENABLED) && s.alloc_gpu <= 0 && !(s.flags & KAI_NODE_HAS_DRA_GPUS);  // node_info.go:697-702
    if (c.plugins & KAI_PLUGIN_PREDICATES) {
        if (!k.cpu_only) {
            if (s.flags & KAI_NODE_HAS_DRA_GPUS) return 0;
            if ((s.flags & KAI_NODE_MIG_ENABLED) && (s.flags & KAI_NODE_MIG_MIXED)) return 0;
        }
        if (!(s.idle[KAI_RES_PODS] + s.rel[KAI_RES_PODS] > 0)) return 0;
        if (s.flags & KAI_NODE_NOT_READY) return 0;
        if (!c.class_fit[(size_t)k.pod_class * c.n_node_classes + s.ncls]) return 0;
        if (c.restrict_nodes) {
            if (!k.cpu_only) { if (!(s.flags & KAI_NODE_GPU_WORKER)) return 0; }
            else if (!(s.flags & KAI_NODE_CPU_WORKER)) return 0;
        }
    }
    uint64_t key = 0;
    const bool lay = key_shared_layout(c);  // (then spread classes are not admitted: their 62 placement bits leave no room, kai_host_prep.hpp)
    if (lay && (s.share_fit & (k.req[KAI_RES_GPU] >= 1 ? 2u : 1u))) key |= 1ull << 63;
    if ((c.plugins & KAI_PLUGIN_NODEAVAILABILITY) && (k.best_effort || fit_idle)) key |= 1ull << (lay ? 62 : 63);
    if ((c.plugins & KAI_PLUGIN_RESOURCETYPE) && k.cpu_only && cpu_node) key |= 1ull << (lay ? 61 : 62);
    uint64_t v = 1;
    if (c.plugins & KAI_PLUGIN_NODEPLACEMENT) {
        const bool gpu = k.r_place == KAI_RES_GPU;
        double cur = gpu ? s.idle[KAI_RES_GPU] + s.rel[KAI_RES_GPU] : s.idle[KAI_RES_CPU] + s.rel[KAI_RES_CPU];
        if (k.strategy == KAI_SPREAD) {
            double overall = gpu ? s.alloc_gpu : s.alloc_cpu, count = overall;
            if (gpu) count = s.gpu_count >= 0 ? (double)s.gpu_count : (double)(int64_t)overall;
            double place = count == 0 ? 0.0 : cur / count;
            union { double d; uint64_t u; } cv; cv.d = place;
            v = cv.u + 1;
        } else {
            v = (uint64_t)((((int64_t)1 << 53) - 1) - (int64_t)cur);  // cur >= 0 here: a node with a negative free amount fits nothing
        }
    }
    return key | v;
}
KAI_HD uint64_t class_key(const KaiCtx& c, const ClassRec& k, int n) { NodeRegs s; load_node(c, n, s); return class_key_regs(c, k, s); }
KAI_HD bool key_better(uint64_t k, int n, uint64_t bk, int bn) { return k > bk || (k == bk && k != 0 && n < bn); }

// Stage job j (see JobPf): cooperative over nl lanes, lane handles the chunk's pods lane, lane + nl, …  Returns this lane's "some pod does not
// qualify" bit; lane 0 writes the job-level fields.  Reads exactly what allocate_job_fast reads when it loads the job itself.
KAI_HD int stage_job_lane(const KaiCtx& c, int j, FastFrame& f, JobPf& out, int lane, int nl) {
    JobPf v;
    v.job = j; v.ok = 0;
    v.n_ps = c.j_n_ps[j]; v.has_topo = c.j_has_topology[j]; v.s = c.j_first_ps[j]; v.first = c.j_first_pod[j];
    v.tta_valid = c.j_tta_valid[j]; v.tta_n = c.j_tta_n[j]; v.jq = c.j_queue[j]; v.jpre = c.j_preempt[j];
    v.ja[0] = c.j_allocated[(size_t)j * 4 + 0]; v.ja[1] = c.j_allocated[(size_t)j * 4 + 1]; v.ja[2] = c.j_allocated[(size_t)j * 4 + 2];
    v.s_pipelined = (v.n_ps == 1) ? c.s_pipelined[v.s] : 1;
    v.shape = v.n_ps == 1 && !v.has_topo && v.s_pipelined == 0 && v.tta_valid && v.tta_n > 0 && v.tta_n <= KAI_FMAX;
    int bad = 0;
    const bool nominated = c.plugins & KAI_PLUGIN_NOMINATEDNODE;
    if (v.shape) for (int i = lane; i < v.tta_n; i += nl) {
        int p = c.tta[v.first + i];
        int k = c.p_scls[p], st = c.p_status[p], on = c.p_on_node[p], nom = c.p_nominated[p];
        f.p[i] = p; f.cls[i] = k;
        bad |= (k < 0) | (st != KAI_POD_PENDING) | (on >= 0) | (nominated && nom >= 0);
        for (int r = 0; r < KAI_MAX_RES; r++) f.req[i][r] = r < c.R ? c.p_req[(size_t)r * c.P + p] : 0.0;
    }
    if (lane == 0) out = v;
    return bad;
}

// ======================================================================================================
// Engine<Backend>: the control flow.  Backend provides
//    void minmax(const KaiCtx&, int r, double& mn, double& mx)          — pack.go:66-86 over the node set (brute force)
//    bool topo_scan(const KaiCtx&, TopoScan&)                           — the node loops of subSetNodesFn on the scan lanes; false = the backend has none
//    bool pfor(const KaiCtx&, const PforReq&)                           — an index loop of the victim search on the scan lanes; false = none
//    void or32(uint32_t* word, uint32_t bits)                           — *word |= bits (atomic where lanes share words)
//    int  best_node(const KaiCtx&, const ScanReq&, double* score)       — arg-max of (score, -index) over fitting nodes (brute force); its score when asked for
//    bool eval_nodes(const KaiCtx&, const ScanReq&, const int32_t* n, int m, double* sc, uint8_t* ok) — scan_node_score of m <= 32 nodes; false = the backend has no lanes for it (unscoped requests only)
//    void begin(const KaiCtx&)                                          — build the in-LDS levels of the class index
//    bool dirty_add(int block) / int dirty_count()                      — list of 64-node blocks whose node state changed
//    void refresh(const KaiCtx&)                                        — re-evaluate the listed blocks, clear the list
//    void class_top(const KaiCtx&, int cls, uint64_t& key, int& node)   — arg-max of class_key over all nodes
//    bool all_dead(const KaiCtx&)                                       — no class has a fitting node
//    (brute-force scans honour local().scope_bits / scope_row / scope_score: node-set bitmap and preferred-level topology scores)
//    void hot(const KaiCtx&, QNode*&, int32_t*& qheap, int32_t*& root_heap) — where the job-order tree lives (LDS if it fits, else HBM)
//    const KaiCtx& ctx() / EngineLocal& local() / void bind(const KaiCtx&) — context and engine scalars (LDS objects on the device)
//    int64_t clock()
// ======================================================================================================
// the engine's own scalars (one set per running action)
// Counters the control lane bumps on every pop / decision / committed operation: kept with the engine's scalars (LDS on the device, a dependent
// global read-modify-write each otherwise) and written to EngineState once, when the action ends.
struct EngineHot {
    int64_t decisions, index_queries, index_refreshes, rollbacks, jobs_attempted, jobs_committed, out_len, stmts;
    int64_t prof[KAI_NPROF];
};
// A brute-force decision over all nodes whose answer is kept: the request (everything but the pod), its best node and that node's score, and how far into the log of changed
// nodes the answer is up to date (Engine::best_node_kept)
constexpr int KAI_BN_ENT = 16, KAI_BN_LOG = 32;
struct BnEnt { ScanReq q; uint64_t h; double score; int32_t node, valid; uint32_t sync, used; };  // h: bn_hash(q), compared first
struct EngineLocal {
    EngineHot h;
    QNode* qn; int32_t *qheap, *root_heap;  // where the job-order tree lives (LDS if it fits, else HBM)
    int32_t root_len, root_init, fail_no_node, pop_leaf;  // pop_leaf: leaf queue the last job was popped from (= its queue)
    double total0, total1, total2;          // proportion totalResource (read-only during an action)
    // scope of the next node scans (general path): node-set bitmap (null = every node) and the preferred-level scores that apply
    KAI_GP(const uint32_t) scope_bits; KAI_GP(const double) scope_score; int32_t scope_row, n_keys;
    uint32_t restricted, pad5;               // bit d: the node set of DFS depth d is narrower than "every node"
    KAI_GP(const uint32_t) base_bits;        // node set every job attempt starts from (null = every node; a scenario's feasible nodes in the victim search)
    int32_t ov_job, ov_orig_valid;           // job currently stood in for by its partial representative (job_solver.go:128-151), -1 = none;
                                             // whether the SESSION job's own tasks-to-allocate cache (saved aside) is still valid
    // victim search: the active job-order instance (0 = the action's, 1 = victims queue, 2 = jobs to allocate of a simulation)
    int32_t *i_sorted, *i_cur, *i_end, *i_side, *i_side_len;  // leaf storage of the active instance
    int32_t jo_kind, cur_inst;               // jo_kind 1 = victims ordering (reversed comparators, victims operands)
    int32_t tpl_valid, mw_poll;              // the pending-job template of the simulation queues matches the committed state; >= 0: this simulation's index in its wave — it is
                                             // given up as soon as an earlier simulation of the wave is known not to have simply failed (MultiCtx::hit), buffer mw_buf
    uint32_t bn_seq, bn_tick; int32_t bn_last;  // (the kept best nodes themselves and their work lists: EngineBig)
  // best nodes of recent brute-force decisions, kept and patched (best_node_kept); the nodes changed since, in order
    int32_t mm_valid[2], mm_nchg, mm_pend;   // NodePreOrderFn's range over ALL nodes, kept between decisions (preorder_range): valid per placement resource (0 CPU, 1 GPU), nodes changed since, the node whose old amounts mm_before holds
    int32_t mm_cn[8]; double mm_lo[2], mm_hi[2], mm_old[8][2];
    int32_t sg_gen, pad_sg;                  // scan grid: number of the last command the control lane put on the table (kai_kernels.hpp)
    int32_t mw_buf, rc_early;                // rc_early: the running simulation stopped right after placing the preemptor because the reclaim validator's verdict (known then) is "no"
    // the victims log of the job being solved (kai_engine_solver.inc): vl_job0 = that job (-1: none), vl_n entries, vl_done = the base queue ran empty behind them, vl_live = the
    // victims-queue instance stands at the log's end in base mode (it can be extended), vl_cur = the running partial job's cursor, vl_mode 1 = this partial job left the log
    // (a recorded victim changed the pop sequence) and pops the queue itself
    // requests that found NO fitting node among a simulation's feasible nodes (victim search): inside one simulation resources are only taken — until a rollback hands some
    // back, which empties the list — so the same request finds none again (a simulation re-places hundreds of evicted one-device tasks of one shape)
    int32_t mw_cow, mw_nsaved;  // victim search on several engines: != 0 = the wave's stamp while a speculative simulation runs — a job's tasks-to-allocate cache is set aside the first time the wave touches it (mw_touch); jobs set aside so far
    int32_t sim_dead_n, sim_dead_on;  // (the requests: EngineBig::sim_dead; so are the DFS frames' node sets, EngineBig::fbits)
    int32_t vl_job0, vl_n, vl_done, vl_live, vl_cur, vl_mode;
    int32_t vl_mat, vl_div_grp;  // entries below vl_mat have their task groups in the scenario (materialised lazily: a scenario the filters drop never needs them); groups the scenario held when the partial job left the log
    struct JoSave { QNode* qn; int32_t *qheap, *root_heap, *sorted, *cur, *end, *side, *side_len; int32_t root_len, root_init, kind, pad; } save[3];
};

// The control lane's larger working data.  Not part of EngineLocal: every scan lane that borrows the engine's pure helpers (Engine<NullBackend>: index loops of the victim search, the
// domain loops of subSetNodesFn) carries an EngineLocal in scratch memory, and the action kernels' private segment — 8 KB per lane times every wave slot of the chip per queue — is what a
// node-sharded group of three ranks on one device ran out of (HSA_STATUS_ERROR_OUT_OF_RESOURCES) when these arrays grew.  Backend::big(): LDS on the device, a member on the host.
struct EngineBig {
    BnEnt bn[KAI_BN_ENT]; int32_t bn_log[KAI_BN_LOG];  // best nodes of recent brute-force decisions, kept and patched (best_node_kept); the nodes changed since, in order
    int32_t bn_m[KAI_BN_LOG]; double bn_sc[KAI_BN_LOG]; uint8_t bn_ok[KAI_BN_LOG];  // work lists of best_node_kept
    ScanReq sim_dead[4];                       // requests that found no fitting node among the running simulation's feasible nodes (EngineLocal::sim_dead_n of them)
    KAI_GP(const uint32_t) fbits[KAI_TDEPTH];  // node set of the DFS frame at each depth (frame_bits)
};

template <class Backend>
struct Engine {
    Backend& be;
    // The context and the engine's own scalars are reached through the backend: on the device they are directly addressed LDS
    // objects (static accessors, nothing goes through this object), on the host plain members of the backend.
    KAI_HD const KaiCtx& cx() const { return be.ctx(); }
    KAI_HD EngineLocal& el() const { return be.local(); }
    KAI_HD EngineBig& eb() const { return be.big(); }
    static constexpr bool kVictim = Backend::kVictim;  // compiled with the victim search (reclaim / preempt / consolidation)
    KAI_HD const SolverCtx& sx() const { return cx().sv; }
    // the job-order tree through pointers the compiler may treat as LDS pointers when the backend says the tree lives there
    // (an assumption on a generic pointer: flat_load/flat_store become ds_read/ds_write, which do not wait on global traffic)
    KAI_HD QNode* qnp() const { QNode* p = el().qn; Backend::assume_tree(p); return p; }
    KAI_HD int32_t* qheapp() const { int32_t* p = el().qheap; Backend::assume_tree(p); return p; }
    KAI_HD int32_t* rootheapp() const { int32_t* p = el().root_heap; Backend::assume_tree(p); return p; }
    KAI_HD Engine(const KaiCtx& ctx, Backend& b) : be(b) {
        be.bind(ctx);
        EngineLocal& e = el();
        e.qn = ctx.qn; e.qheap = ctx.qheap; e.root_heap = ctx.root_heap; e.root_len = 0; e.root_init = 0; e.fail_no_node = 0;
        e.total0 = ctx.st->total[0]; e.total1 = ctx.st->total[1]; e.total2 = ctx.st->total[2];
        e.scope_bits = nullptr; e.scope_score = nullptr; e.scope_row = -1; e.n_keys = 0; e.restricted = 0; e.base_bits = nullptr; e.ov_job = -1; e.jo_kind = 0; e.cur_inst = 0; e.tpl_valid = 0; e.mw_poll = -1; e.mw_buf = 0; e.sim_dead_n = 0; e.sim_dead_on = 0; e.mw_cow = 0; e.mw_nsaved = 0; e.vl_job0 = -1; e.vl_n = 0; e.vl_done = 0; e.vl_live = 0; e.vl_cur = 0; e.vl_mode = 0; e.vl_mat = 0; e.vl_div_grp = 0; e.rc_early = 0; e.mm_valid[0] = e.mm_valid[1] = 0; e.mm_nchg = 0; e.mm_pend = -1; for (int i = 0; i < KAI_BN_ENT; i++) { if constexpr (Backend::kBig) be.big().bn[i].valid = 0; } e.bn_seq = 0; e.bn_tick = 0; e.bn_last = 0;
        e.i_sorted = (int32_t*)ctx.lq_sorted; e.i_cur = (int32_t*)ctx.lq_cur; e.i_end = (int32_t*)ctx.lq_end; e.i_side = (int32_t*)ctx.lq_side; e.i_side_len = (int32_t*)ctx.lq_side_len;
    }

    KAI_HD void fault(int code, int line = __builtin_LINE()) { if (!cx().st->fault) { cx().st->fault = code; cx().st->fault_line = line; } }
    KAI_HD double preq(int p, int r) const { return cx().p_req[(size_t)r * cx().P + p]; }
#ifdef KAI_SHARED_GPUS
    KAI_HD bool pod_shared(int p) const { return cx().shared_on && cx().p_shared[p]; }
    KAI_HD int64_t pod_gmem(int p) const { return cx().quota_on ? cx().p_gmem[p] : 0; }
    KAI_HD double pacc_gpu(int p) const { return cx().quota_on ? cx().p_acc_gpu[p] : preq(p, KAI_RES_GPU); }   // AcceptedResource.GetGpusQuota()
    KAI_HD double pacc_gpus(int p) const { return cx().quota_on ? cx().p_acc_gpu[p] - cx().p_mig_q[p] : preq(p, KAI_RES_GPU); }  // AcceptedResource.GPUs(): no MIG instances
    KAI_HD double ppend_gpu(int p) const { return cx().quota_on ? cx().p_pend_gpu[p] : preq(p, KAI_RES_GPU); }
    KAI_HD double pquota_gpu(int p) const { return cx().quota_on ? cx().p_quota_gpu[p] : preq(p, KAI_RES_GPU); }  // ResReq.GetGpusQuota()
    KAI_HD int pod_kind(int p) const { return cx().quota_on ? cx().p_kind[p] : (KAI_KIND_REGULAR | (preq(p, KAI_RES_GPU) > 0 ? KAI_KIND_GPUS : 0)); }
    KAI_HD bool mig_row(int r) const { return cx().mig_on && cx().res_mig_g[r] > 0; }
#else
    KAI_HD bool pod_shared(int) const { return false; }
    KAI_HD int64_t pod_gmem(int) const { return 0; }
    KAI_HD double pacc_gpu(int p) const { return preq(p, KAI_RES_GPU); }
    KAI_HD double pacc_gpus(int p) const { return preq(p, KAI_RES_GPU); }
    KAI_HD double ppend_gpu(int p) const { return preq(p, KAI_RES_GPU); }
    KAI_HD double pquota_gpu(int p) const { return preq(p, KAI_RES_GPU); }
    KAI_HD int pod_kind(int p) const { return KAI_KIND_REGULAR | (preq(p, KAI_RES_GPU) > 0 ? KAI_KIND_GPUS : 0); }
    KAI_HD bool mig_row(int) const { return false; }
#endif
    KAI_HD bool pod_cpu_only(int p) const { return !(preq(p, KAI_RES_GPU) > 0 || pod_gmem(p) > 0 || (pod_kind(p) & KAI_KIND_MIG)); }  // pod_info.go:340-347 IsRequireAnyKindOfGPU
    KAI_HD bool pod_best_effort(int p) const {  // ResourceRequirements.IsEmpty (resource_requirment.go:99-104, base_resources.go:119-130)
        if (preq(p, KAI_RES_GPU) > 0.01 || pod_gmem(p) > 0) return false;  // (node_info.go:169-170: a gpu-memory request is never best effort)
        if (pod_kind(p) & KAI_KIND_MIG) return false;                      // gpu_resource_requirment.go:89-104: MIG instances make the request non-empty
        if (preq(p, KAI_RES_CPU) >= 10.0 || preq(p, KAI_RES_MEM) >= 10.0 * 1024 * 1024) return false;
        for (int r = KAI_RES_PODS; r < cx().R; r++) if (!mig_row(r) && preq(p, r) >= 10.0) return false;
        return true;
    }
    KAI_HD bool should_allocate(int p, bool real) const {  // pod_info.go:518-521
        int s = cx().p_status[p];
        return s == KAI_POD_PENDING || (!real && s == KAI_POD_RELEASING && cx().p_virtual[p]);
    }
    // quota triple of a pod: utils.QuantifyResourceRequirements (plugins/proportion/utils/utils.go:15-17)
    KAI_HD double pquota(int p, int k) const { return k == KAI_Q_CPU ? preq(p, KAI_RES_CPU) : k == KAI_Q_MEM ? preq(p, KAI_RES_MEM) : pquota_gpu(p); }  // QuantifyResourceRequirements(ResReq)

    // ------------------------------------------------------------------ class index bookkeeping
    // the dirty-block list lives in the backend (LDS on the device) so that this object holds no dynamically indexed storage
    KAI_HD void flush_index() {
        int nd = be.dirty_count();
        if (nd) { int64_t t = be.clock(); be.refresh(cx()); el().h.index_refreshes += nd; el().h.prof[PF_REFRESH] += be.clock() - t; }
    }
    // getMinMaxPerNode (plugins/nodeplacement/pack.go:66-86) over all nodes is a pass per decision when decisions have no class index (shared GPUs).  Between two decisions one or
    // two nodes change, so the range is kept and patched: a node that was strictly inside the range can only widen it; one that sat ON a bound and moves outward takes the bound
    // along; one that sat on a bound and moves inward may or may not have been alone there — then the next decision makes the pass again.  Same minima / maxima of the same
    // amounts, so the same doubles.  mm_before(n) holds the amounts before a change (node_apply); a change that did not announce itself (the staged job path) drops the range.
    KAI_HD double mm_cur(int r, int n) const { return cx().n_idle[(size_t)r * cx().N + n] + cx().n_rel[(size_t)r * cx().N + n]; }
    KAI_HD void mm_drop() { el().mm_valid[0] = el().mm_valid[1] = 0; el().mm_nchg = 0; el().mm_pend = -1; }
    KAI_HD void mm_before(int n) {
        if (!(el().mm_valid[0] | el().mm_valid[1])) return;
        el().mm_pend = n;
        for (int i = 0; i < el().mm_nchg; i++) if (el().mm_cn[i] == n) return;  // already listed: its first amounts are the ones the kept range knows
        if (el().mm_nchg >= 8) { mm_drop(); return; }
        const int i = el().mm_nchg++;
        el().mm_cn[i] = n; el().mm_old[i][0] = mm_cur(KAI_RES_CPU, n); el().mm_old[i][1] = mm_cur(KAI_RES_GPU, n);
    }
    KAI_HD void mm_apply() {
        for (int i = 0; i < el().mm_nchg; i++) {
            const int n = el().mm_cn[i];
            for (int k = 0; k < 2; k++) {
                if (!el().mm_valid[k]) continue;
                const int r = k ? KAI_RES_GPU : KAI_RES_CPU;
                if (cx().n_alloc[(size_t)r * cx().N + n] == 0) continue;  // not part of the range (pack.go:70-73)
                const double o = el().mm_old[i][k], v = mm_cur(r, n);
                double& lo = el().mm_lo[k]; double& hi = el().mm_hi[k];
                if (o > lo) { if (v < lo) lo = v; } else if (v <= lo) lo = v; else { el().mm_valid[k] = 0; continue; }
                if (o < hi) { if (v > hi) hi = v; } else if (v >= hi) hi = v; else el().mm_valid[k] = 0;
            }
        }
        el().mm_nchg = 0;
    }
    KAI_HD void preorder_range(int r, double& mn, double& mx) {
        if (el().scope_bits || (r != KAI_RES_CPU && r != KAI_RES_GPU)) { be.minmax(cx(), r, mn, mx); return; }  // a node set: not kept
        const int k = r == KAI_RES_GPU ? 1 : 0;
        mm_apply();
        if (!el().mm_valid[k]) { be.minmax(cx(), r, el().mm_lo[k], el().mm_hi[k]); el().mm_valid[k] = 1; }
        mn = el().mm_lo[k]; mx = el().mm_hi[k];
    }
    KAI_HD void mark_dirty(int n) {
        eb().bn_log[el().bn_seq % KAI_BN_LOG] = n; el().bn_seq++;  // every change of a node's amounts or GPU groups passes here
        if (el().mm_valid[0] | el().mm_valid[1]) { if (el().mm_pend != n) mm_drop(); el().mm_pend = -1; }
        if (!cx().use_index) return;
        int b = n / KAI_BLOCK;
        if (be.dirty_add(b)) return;
        flush_index();  // list full
        be.dirty_add(b);
    }
    KAI_HD void invalidate_path(int q) { for (int x = q; x >= 0; x = qnp()[x].parent) qnp()[x].flags &= ~QF_VALID; }

    // ------------------------------------------------------------------ status bookkeeping
    // PodGroupInfo.UpdateTaskStatus (api/podgroup_info/job_info.go:228-287) + PodSet.AssignTask (subgroup_info/podset.go:56-99)
    KAI_HD void update_task_status(int p, int status) {
        int j = cx().p_job[p], s = cx().p_podset[p], old = cx().p_status[p];
        if (st_allocated(old)) for (int k = 0; k < 3; k++) cx().j_allocated[(size_t)j * 4 + k] -= pquota(p, k);
        if (st_active_allocated(old)) cx().s_active_alloc[s]--;
        if (st_active_used(old)) cx().s_active_used[s]--;
        if (st_alive(old)) cx().s_alive[s]--;
        if (old == KAI_POD_GATED) cx().s_gated[s]--;
        if (old == KAI_POD_PIPELINED) cx().s_pipelined[s]--;
        if (old == KAI_POD_PENDING) cx().j_n_pending[j]--;
        cx().p_status[p] = status;
        if (st_allocated(status)) for (int k = 0; k < 3; k++) cx().j_allocated[(size_t)j * 4 + k] += pquota(p, k);
        if (st_active_allocated(status)) cx().s_active_alloc[s]++;
        if (st_active_used(status)) cx().s_active_used[s]++;
        if (st_alive(status)) cx().s_alive[s]++;
        if (status == KAI_POD_GATED) cx().s_gated[s]++;
        if (status == KAI_POD_PIPELINED) cx().s_pipelined[s]++;
        if (status == KAI_POD_PENDING) cx().j_n_pending[j]++;
        // invalidateTasksCache (job_info.go:253-256) — of the SESSION's job: the partial representative the victim search stands in for
        // it keeps the chunk it cached when it was made (statement operations re-index the original job, not the clone)
        if constexpr (kVictim) { if (j == el().ov_job) { el().ov_orig_valid = 0; return; } mw_touch(j); }
        cx().j_tta_valid[j] = 0;
    }

    // NodeInfo.GetSumOfIdleGPUs / GetSumOfReleasingGPUs (node_info.go:592-628): whole GPUs plus, with shared GPUs, the free / releasing portions on them
    template <class A> KAI_HD double mig_sum(A arr, int n) const {  // idle / releasing MIG instances by their GPU weight (node_info.go:596-606, 615-625)
        double q = 0;
#ifdef KAI_SHARED_GPUS
        if (cx().mig_on) for (int r = KAI_RES_PODS + 1; r < cx().R; r++) if (cx().res_mig_g[r] > 0) q += (double)((int64_t)cx().res_mig_g[r] * (int64_t)arr[(size_t)r * cx().N + n]);
#endif
        return q;
    }
    KAI_HD double gpus_idle_sum(int n) const {
#ifdef KAI_SHARED_GPUS
        if (cx().shared_on) { SgNode g{cx(), n}; return g.sum_available_shared() + cx().n_idle[(size_t)KAI_RES_GPU * cx().N + n] + mig_sum(cx().n_idle, n); }
#endif
        return cx().n_idle[(size_t)KAI_RES_GPU * cx().N + n] + mig_sum(cx().n_idle, n);
    }
    KAI_HD int64_t gpus_free_mem(int n) const {  // GPU memory behind GetSumOfIdleGPUs + GetSumOfReleasingGPUs (node_info.go:592-628): whole devices count in full
#ifdef KAI_SHARED_GPUS
        if (cx().quota_on) { SgNode g{cx(), n}; return (cx().shared_on ? g.mem_available_shared() + g.mem_releasing_shared() : 0) + (int64_t)(cx().n_idle[(size_t)KAI_RES_GPU * cx().N + n] + mig_sum(cx().n_idle, n)) * g.gpu_mem() + (int64_t)(cx().n_rel[(size_t)KAI_RES_GPU * cx().N + n] + mig_sum(cx().n_rel, n)) * g.gpu_mem(); }
#endif
        return 0;
    }
    KAI_HD double gpus_rel_sum(int n) const {
#ifdef KAI_SHARED_GPUS
        if (cx().shared_on) { SgNode g{cx(), n}; return g.sum_releasing_shared() + cx().n_rel[(size_t)KAI_RES_GPU * cx().N + n] + mig_sum(cx().n_rel, n); }
#endif
        return cx().n_rel[(size_t)KAI_RES_GPU * cx().N + n] + mig_sum(cx().n_rel, n);
    }
    // ------------------------------------------------------------------ node accounting (api/node_info/node_info.go)
    KAI_HD void node_apply(int n, int p, int status, double sign, int grp_of_copy = -2) {  // addTaskResources :457-493 / removeTaskResources :515-551
        mm_before(n);
        for (int r = 0; r < cx().R; r++) {
            double v = preq(p, r); if (v == 0) continue;
#ifdef KAI_SHARED_GPUS
            if (r == KAI_RES_GPU && pod_shared(p)) continue;  // getAcceptedTaskResourceWithoutSharedGPU (gpu_sharing_node_info.go:52-66)
#endif
            size_t i = (size_t)r * cx().N + n;
            cx().n_used[i] += sign * v;
            if (status == KAI_POD_RELEASING) { cx().n_rel[i] += sign * v; cx().n_idle[i] -= sign * v; }
            else if (status == KAI_POD_PIPELINED) cx().n_rel[i] -= sign * v;
            else cx().n_idle[i] -= sign * v;
        }
#ifdef KAI_SHARED_GPUS
        if (pod_shared(p)) {  // addSharedTaskResources / removeSharedTaskResources with the group of the node's own copy
            SgNode g{cx(), n}; const int grp = grp_of_copy != -2 ? grp_of_copy : cx().p_on_group[p];
            if (grp >= 0) { bool ok = sign > 0 ? g.add(status, cx().p_mem[p], grp) : g.remove(status, cx().p_mem[p], grp); if (!ok) fault(FAULT_INTERNAL); if (cx().use_index) g.refit(); }
        }
#endif
        mark_dirty(n);
    }
    // further residencies (victim search only): open addressing, key = pod << 32 | node, -1 empty, -2 deleted
    KAI_HD int xr_find(int p, int n) const {
        const int64_t key = ((int64_t)p << 32) | (uint32_t)n; uint32_t h = (uint32_t)(((uint64_t)key * 0x9E3779B97F4A7C15ull) >> 33) & (uint32_t)sx().xr_mask;
        for (;;) { int64_t k = sx().xr_key[h]; if (k == key) return (int)h; if (k == -1) return -1; h = (h + 1) & (uint32_t)sx().xr_mask; }
    }
    KAI_HD void xr_insert(int p, int n, int status) {
        const int64_t key = ((int64_t)p << 32) | (uint32_t)n; uint32_t h = (uint32_t)(((uint64_t)key * 0x9E3779B97F4A7C15ull) >> 33) & (uint32_t)sx().xr_mask;
        for (;;) { int64_t k = sx().xr_key[h]; if (k < 0) { sx().xr_key[h] = key; sx().xr_status[h] = status; return; } h = (h + 1) & (uint32_t)sx().xr_mask; }
    }
    KAI_HD bool node_add_task(int n, int p) {  // AddTask :384-417
        if (st_active_used(cx().p_status[p])) cx().p_accepted[p] = 1;  // setAcceptedResources :746-766
        if (cx().p_on_node[p] == n) return false;                    // "task already on node"
        if constexpr (kVictim) { if (xr_find(p, n) >= 0) return false; }
        if (cx().p_on_node[p] >= 0) {
            if constexpr (kVictim) {
                xr_insert(p, n, cx().p_status[p]);
#ifdef KAI_SHARED_GPUS
                if (cx().shared_on) { sx().xr_group[xr_find(p, n)] = cx().p_group[p]; node_apply(n, p, cx().p_status[p], 1.0, cx().p_group[p]); return true; }
#endif
                node_apply(n, p, cx().p_status[p], 1.0);
                return true;
            }
            fault(FAULT_INTERNAL); return false;
        }
        cx().p_on_node[p] = n; cx().p_on_node_status[p] = cx().p_status[p];
#ifdef KAI_SHARED_GPUS
        if (cx().shared_on) cx().p_on_group[p] = cx().p_group[p];
#endif
        node_apply(n, p, cx().p_status[p], 1.0);
        return true;
    }
    KAI_HD bool on_node(int p, int n) const {
        if (cx().p_on_node[p] == n) return true;
        if constexpr (kVictim) return xr_find(p, n) >= 0;
        return false;
    }
    KAI_HD bool node_remove_task(int n, int p) {  // RemoveTask :495-513 — with the status the node's copy was added with
        if (cx().p_on_node[p] == n) {
            node_apply(n, p, cx().p_on_node_status[p], -1.0);
            cx().p_on_node[p] = -1;
            return true;
        }
        if constexpr (kVictim) { int h = xr_find(p, n); if (h >= 0) {
#ifdef KAI_SHARED_GPUS
            if (cx().shared_on) { node_apply(n, p, sx().xr_status[h], -1.0, sx().xr_group[h]); sx().xr_key[h] = -2; return true; }
#endif
            node_apply(n, p, sx().xr_status[h], -1.0); sx().xr_key[h] = -2; return true; } }
        return false;
    }
#ifdef KAI_SHARED_GPUS
    KAI_HD int group_on_node(int p, int n) const {  // GPUGroups of the node's own copy of the task
        if (cx().p_on_node[p] == n) return cx().p_on_group[p];
        if constexpr (kVictim) { int h = xr_find(p, n); if (h >= 0) return sx().xr_group[h]; }
        return -1;
    }
    // ConsolidateSharedPodInfoToDifferentGPU (gpu_sharing_node_info.go:247-249 → addTask(ti, true)): the node's copy of the task is REPLACED — its
    // amounts on the old GPU group stay booked as they are — and the task is added on its new group
    KAI_HD void node_consolidate_shared(int n, int p) {
        if (st_active_used(cx().p_status[p])) cx().p_accepted[p] = 1;
        if (cx().p_on_node[p] == n) { cx().p_on_node_status[p] = cx().p_status[p]; cx().p_on_group[p] = cx().p_group[p]; node_apply(n, p, cx().p_status[p], 1.0); return; }
        if constexpr (kVictim) { int h = xr_find(p, n); if (h >= 0) { sx().xr_status[h] = cx().p_status[p]; sx().xr_group[h] = cx().p_group[p]; node_apply(n, p, cx().p_status[p], 1.0, cx().p_group[p]); return; } }
        fault(FAULT_INTERNAL);
    }
#endif
    KAI_HD bool node_update_task(int n, int p) { if (!node_remove_task(n, p)) return false; return node_add_task(n, p); }  // :571-576

    // ------------------------------------------------------------------ proportion event handlers (plugins/proportion/proportion.go:443-489)
    KAI_HD void queue_event(int p, double sign) {
        if (!(cx().plugins & KAI_PLUGIN_PROPORTION)) return;
        if (!cx().p_accepted[p]) return;  // AcceptedResource is empty until the task was added to a node
        int j = cx().p_job[p]; bool np = !cx().j_preempt[j];
        for (int q = cx().j_queue[j]; q >= 0; q = qnp()[q].parent) {
            for (int k = 0; k < 3; k++) {
                QShare& s = cx().q_share[(size_t)q * 3 + k]; double v = k == KAI_Q_GPU ? pacc_gpu(p) : pquota(p, k);  // QuantifyResourceRequirements(AcceptedResource)
                s.allocated += sign * v;
                if (np) s.allocated_np += sign * v;
            }
            qnp()[q].flags &= ~QF_VALID;
        }
    }

    // ------------------------------------------------------------------ Statement (framework/statement.go)
    KAI_HD int checkpoint() const { return cx().st->ops_len; }
    KAI_HD bool push_op(const StmtOp& o) { if (cx().st->ops_len >= cx().ops_cap) { fault(FAULT_OPS_CAP); return false; } cx().ops[cx().st->ops_len++] = o; return true; }
    KAI_HD int op_undo_of(int target) const {  // the first OP_UNDO on the log that names `target` (the reference scans for it, statement.go:652-663), -1 = none: through the link undo_operation left
        const int u = cx().ops[target].undo_link - 1;
        return (u > target && u < cx().st->ops_len && cx().ops[u].name == OP_UNDO && cx().ops[u].op_index == target) ? u : -1;
    }
    KAI_HD bool op_valid(int i) const {  // :652-663 — valid(i) = no undo of i, or that undo is itself undone (iterative form)
        if (cx().st->n_undo == 0) return true;
        bool valid = true; int target = i;
        for (;;) {
            const int u = op_undo_of(target);
            if (u < 0) return valid;
            valid = !valid; target = u;
        }
    }
    KAI_HD bool stmt_allocate(int p, int n) {  // :297-358
        update_task_status(p, KAI_POD_ALLOCATED);
        cx().p_node[p] = n;
        if (!node_add_task(n, p)) return false;
        queue_event(p, 1.0);
        StmtOp o{}; o.name = OP_ALLOCATE; o.pod = p; o.next_node = n; o.prev_virtual = cx().p_virtual[p]; o.op_index = -1;
        if (!push_op(o)) return false;
        cx().p_virtual[p] = 1;
        return true;
    }
    KAI_HD bool stmt_pipeline(int p, int n, bool update_if_exists) {  // :197-295
        bool found_on_node = on_node(p, n);
#ifdef KAI_SHARED_GPUS
        // a shared-GPU task that was evicted from this node and comes back on ANOTHER GPU of it (:208-213)
        const bool shared_task = pod_shared(p);
        const int grp_there = (shared_task && found_on_node) ? group_on_node(p, n) : -1;
        const bool is_move = shared_task && found_on_node && cx().p_group[p] >= 0 && cx().p_group[p] != grp_there;
        if (found_on_node && !update_if_exists && !is_move) { if (shared_task) cx().p_group[p] = grp_there; return stmt_unevict_earliest(p); }
        int prev_group = shared_task ? cx().p_group[p] : -1;
#else
        if (found_on_node && !update_if_exists) return stmt_unevict_earliest(p);
#endif
        int prev_status = cx().p_status[p];
        update_task_status(p, KAI_POD_PIPELINED);
        int prev_node = cx().p_node[p]; cx().p_node[p] = n; int prev_virtual = cx().p_virtual[p];
#ifdef KAI_SHARED_GPUS
        if (is_move) { prev_group = grp_there; node_consolidate_shared(n, p); } else
#endif
        if (found_on_node) node_update_task(n, p); else if (!node_add_task(n, p)) return false;
        queue_event(p, 1.0);
        StmtOp o{}; o.name = OP_PIPELINE; o.pod = p; o.prev_status = prev_status; o.prev_node = prev_node; o.next_node = n; o.prev_virtual = prev_virtual; o.op_index = -1;
#ifdef KAI_SHARED_GPUS
        o.pad = prev_group;
#endif
        if (!push_op(o)) return false;
        cx().p_virtual[p] = 1;
        return true;
    }
    KAI_HD bool stmt_evict(int p) {  // :63-126
        int n = cx().p_node[p]; if (n < 0) return false;
        int prev_status = cx().p_status[p], prev_virtual = cx().p_virtual[p];
        update_task_status(p, KAI_POD_RELEASING);
        if (!node_update_task(n, p)) return false;
        queue_event(p, -1.0);
        StmtOp o{}; o.name = OP_EVICT; o.pod = p; o.prev_status = prev_status; o.prev_node = n; o.prev_virtual = prev_virtual; o.op_index = -1;
#ifdef KAI_SHARED_GPUS
        o.pad = cx().shared_on ? cx().p_group[p] : -1;
#endif
        if (!push_op(o)) return false;
        cx().p_virtual[p] = 1;
        return true;
    }
    KAI_HD void unallocate(int p, int prev_virtual) {  // :391-425
        update_task_status(p, KAI_POD_PENDING);
        int n = cx().p_node[p];
        if (n >= 0) node_remove_task(n, p);
        cx().p_node[p] = -1; cx().p_virtual[p] = (uint8_t)prev_virtual;
        queue_event(p, -1.0);
    }
    KAI_HD void unpipeline(int p, int prev_node, int prev_status, int prev_virtual, int prev_group = -1) {  // :431-476
        update_task_status(p, prev_status);
#ifdef KAI_SHARED_GPUS
        if (pod_shared(p)) cx().p_group[p] = prev_group;  // :452
#else
        (void)prev_group;
#endif
        int host = cx().p_node[p]; cx().p_node[p] = prev_node; cx().p_virtual[p] = (uint8_t)prev_virtual;
        if (host >= 0) node_remove_task(host, p);
        queue_event(p, -1.0);
    }
    KAI_HD void unevict(int p, int prev_status, int n, int prev_virtual, int prev_group = -1) {  // :152-195
        update_task_status(p, prev_status);
#ifdef KAI_SHARED_GPUS
        if (pod_shared(p)) cx().p_group[p] = prev_group;  // :167
#else
        (void)prev_group;
#endif
        cx().p_virtual[p] = (uint8_t)prev_virtual;
        if (n >= 0) { if (on_node(p, n)) node_update_task(n, p); else node_add_task(n, p); }
        queue_event(p, 1.0);
    }
    KAI_HD bool stmt_unevict_earliest(int p) {  // Unevict → undoEarliestValidOperation :478-481,578-600
        for (int i = 0; i < cx().st->ops_len; i++) {
            if (!op_valid(i)) continue;
            if (cx().ops[i].name != OP_EVICT || cx().ops[i].pod != p) continue;
            undo_operation(i); return true;
        }
        return false;
    }
    KAI_HD void undo_operation(int index) {  // :602-643
        if (!op_valid(index)) return;
        if constexpr (kVictim) el().sim_dead_n = 0;  // resources come back: a request that found no node may find one again
        StmtOp op = cx().ops[index];
        switch (op.name) {
            case OP_EVICT: unevict(op.pod, op.prev_status, op.prev_node, op.prev_virtual, op.pad); break;
            case OP_PIPELINE: unpipeline(op.pod, op.prev_node, op.prev_status, op.prev_virtual, op.pad); break;
            case OP_ALLOCATE: unallocate(op.pod, op.prev_virtual); break;
            default: {  // undo of an undo = redo the original operation (:606-623)
                if constexpr (kVictim) {
                    StmtOp orig = cx().ops[op.op_index];
                    if (orig.name == OP_EVICT) stmt_evict(orig.pod); else if (orig.name == OP_PIPELINE) stmt_pipeline(orig.pod, orig.next_node, true);
                    else if (orig.name == OP_ALLOCATE) stmt_allocate(orig.pod, orig.next_node); else undo_operation(orig.op_index);
                    break;
                }
                fault(FAULT_INTERNAL); return;
            }
        }
        StmtOp u{}; u.name = OP_UNDO; u.pod = -1; u.op_index = index;
        const bool linked = op_undo_of(index) >= 0;  // (an operation undone, redone and undone again: the scan of the reference stops at its FIRST undo, so that link stays)
        if (push_op(u)) { cx().st->n_undo++; if (!linked) cx().ops[index].undo_link = cx().st->ops_len; }
    }
    KAI_HD void truncate_ops(int cp) {
        for (int i = cp; i < cx().st->ops_len; i++) if (cx().ops[i].name == OP_UNDO) cx().st->n_undo--;
        cx().st->ops_len = cp;
    }
    KAI_HD void rollback(int cp) {  // :48-61
        for (int i = cx().st->ops_len - 1; i >= cp; i--) undo_operation(i);
        truncate_ops(cp); el().h.rollbacks++;
    }
    KAI_HD void discard() { for (int i = cx().st->ops_len - 1; i >= 0; i--) undo_operation(i); truncate_ops(0); }  // :522-534
    KAI_HD bool convert_all_allocated_to_pipelined(int job) {  // :483-516
        int n0 = cx().st->ops_len;
        for (int i = 0; i < n0; i++) {
            StmtOp op = cx().ops[i];
            if (op.name != OP_ALLOCATE || cx().p_job[op.pod] != job) continue;
            int node = cx().p_node[op.pod];
            unallocate(op.pod, 1);
            if (!stmt_pipeline(op.pod, node, true)) return false;
        }
        int w = 0;
        for (int i = 0; i < cx().st->ops_len; i++) {
            StmtOp op = cx().ops[i];
            if (op.name == OP_ALLOCATE && cx().p_job[op.pod] == job) continue;
            cx().ops[w++] = op;
        }
        cx().st->ops_len = w;
        return true;
    }
    KAI_HD void commit() {  // :536-575 — the cache side effects are replayed by the caller from out_ops
        const int64_t len0 = el().h.out_len;
        for (int i = 0; i < cx().st->ops_len; i++) {
            if (!op_valid(i)) continue;
            StmtOp op = cx().ops[i]; if (op.name == OP_UNDO) continue;
            if (el().h.out_len >= cx().out_cap) { fault(FAULT_OUT_CAP); break; }
            kai_op o; o.seq = el().h.out_len; o.pod = op.pod; o.job = cx().p_job[op.pod]; o.node = cx().p_node[op.pod]; o.stmt = (int32_t)el().h.stmts; o.pad = 0;
            if (op.name == OP_EVICT) { o.kind = KAI_OP_EVICT; o.node = op.prev_node; cx().p_virtual[op.pod] = 0; cx().st->non_allocate_commits++; }
            else if (op.name == OP_PIPELINE) { o.kind = KAI_OP_PIPELINE; cx().st->non_allocate_commits++; }
            else { o.kind = KAI_OP_ALLOCATE; update_task_status(op.pod, KAI_POD_BINDING); }  // ssn.BindPod (framework/session.go:111-126)
            cx().out_ops[el().h.out_len++] = o;
        }
        if (el().h.out_len > len0) el().h.stmts++;
        truncate_ops(0);
    }

    // ------------------------------------------------------------------ order functions
    // pod-set counters as the job object at hand holds them: the session's live counters, except for the job the victim search is
    // solving, which is stood in for by its partial representative (CloneWithTasks of the pending tasks: nothing allocated yet,
    // minAvailable = number of its tasks in the pod-set; job_solver.go:128-151)
    KAI_HD int ps_min(int s) const { if constexpr (kVictim) { if (cx().s_job[s] == el().ov_job) return sx().s_ov_min[s]; } return cx().s_min[s]; }
    KAI_HD int ps_act(int s) const { if constexpr (kVictim) { if (cx().s_job[s] == el().ov_job) return 0; } return cx().s_active_alloc[s]; }
    KAI_HD bool job_has_pod(int j, int p) const { if constexpr (kVictim) { if (j == el().ov_job) return sx().p_partial[p] != 0; } return true; }
    KAI_HD int min_available_state(int j) const {  // plugins/elastic/elastic.go:53-65 → 0 below, 1 exactly, 2 above
        bool exactly = true;
        for (int k = 0; k < cx().j_n_ps[j]; k++) {
            int s = cx().j_first_ps[j] + k; int n = ps_act(s), m = ps_min(s);
            if (n < m) return 0;
            if (n > m) exactly = false;
        }
        return exactly ? 1 : 2;
    }
    KAI_HD bool job_order(int l, int r) const {  // framework/session_plugins.go:227-242
        if (cx().plugins & KAI_PLUGIN_PRIORITY) {
            if (cx().j_prio[l] > cx().j_prio[r]) return true;
            if (cx().j_prio[l] < cx().j_prio[r]) return false;
        }
        if (cx().plugins & KAI_PLUGIN_ELASTIC) {  // plugins/elastic/elastic.go:25-51 — an order on (below < exactly < above)
            int ls = min_available_state(l), rs = min_available_state(r);
            if (ls == 0 && rs != 0) return true;
            if (ls == 1 && rs == 2) return true;
            if (ls != 0 && rs == 0) return false;
            if (ls == 2 && rs == 1) return false;
        }
        if (cx().j_created[l] == cx().j_created[r]) return cx().j_uid_rank[l] < cx().j_uid_rank[r];
        return cx().j_created[l] < cx().j_created[r];
    }
    // order of a leaf's job heap (job_order_by_queue.go:249-262): JobOrderFn, negated for a victims queue
    KAI_HD bool job_less(int a, int b) const { if constexpr (kVictim) { if (el().jo_kind) return !job_order_view(a, b); } return job_order(a, b); }
    KAI_HD bool podset_order(int l, int r) const {  // session_plugins.go:262-271 + plugins/subgrouporder/subgroup_order.go:31-62
        if (cx().plugins & KAI_PLUGIN_SUBGROUPORDER) {
            int ln = ps_act(l), rn = ps_act(r), lm = ps_min(l), rm = ps_min(r);
            bool ls = ln >= lm, rs = rn >= rm;
            if (!ls && !rs) return cx().s_name_rank[l] < cx().s_name_rank[r];
            if (!ls) return true;
            if (!rs) return false;
            double lr = (double)ln / (double)lm, rr = (double)rn / (double)rm;
            if (lr < rr) return true;
            if (rr < lr) return false;
        }
        return cx().s_name_rank[l] < cx().s_name_rank[r];
    }

    // ------------------------------------------------------------------ tasks to allocate (api/podgroup_info/allocation_info.go:27-177)
    // The tasks-to-allocate cache of job j is about to change (invalidated by a statement operation, rebuilt by ensure_tta) inside a SPECULATIVE simulation of a wave
    // (kai_engine_solver.inc solve_partial_multi): set aside, once per wave, what it held when the wave started — mw_caches_restore puts it back after the simulation.
    KAI_HD void mw_touch(int j) {
        if constexpr (kVictim) {
            const int st = el().mw_cow;
            if (!st || (sx().mw_sv[j] >> 1) == st) return;
            const int first = cx().j_first_pod[j], cnt = cx().j_tta_valid[j] ? cx().j_tta_n[j] : 0;
            sx().mw_sv[j] = (st << 1) | (cx().j_tta_valid[j] ? 1 : 0); sx().mw_sn[j] = cx().j_tta_n[j]; sx().mw_sj[el().mw_nsaved++] = j;
            for (int k = 0; k < 4; k++) sx().mw_sres[(size_t)j * 4 + k] = cx().j_tta_res[(size_t)j * 4 + k];
            for (int k = 0; k < cnt; k++) sx().mw_stta[first + k] = cx().tta[first + k];
        } else (void)j;
    }
    KAI_HD void ensure_tta(int j, bool real) {
        if (cx().j_tta_valid[j]) return;
        mw_touch(j);
        int first = cx().j_first_pod[j], np = cx().j_n_pods[j], nps = cx().j_n_ps[j], ps0 = cx().j_first_ps[j];
        int out = 0;
        if (nps == 1) {  // one pod-set: the chunk is the first max_tasks allocatable pods in task order
            int s = ps0, act = ps_act(s), mn = ps_min(s);
            int max_tasks = act >= mn ? 1 : mn - act;  // getNumTasksToAllocate :145-153
            for (int i = 0; i < np && out < max_tasks; i++) { int p = cx().j_pods_sorted[first + i]; if (should_allocate(p, real) && job_has_pod(j, p)) cx().tta[first + out++] = p; }
        } else {
            int unsat = 0; for (int k = 0; k < nps; k++) if (ps_act(ps0 + k) < ps_min(ps0 + k)) unsat++;
            int max_sg = unsat > 0 ? unsat : 1, n_sg = 0;
            // pod-sets leave the priority queue in PodSetOrderFn order: repeated arg-min, podsets per job are few
            uint64_t taken_lo = 0;  // bitmap for up to 64 pod-sets; larger jobs fall back to the scratch array
            for (int round = 0; round < nps && n_sg < max_sg; round++) {
                int best = -1;
                for (int k = 0; k < nps; k++) {
                    bool taken = k < 64 ? ((taken_lo >> k) & 1) : (cx().scratch[first + (k - 64)] != 0);
                    if (taken) continue;
                    const int cur = best < 0 ? k : best;  // branch-free, see allocate_job
                    const bool take = podset_order(ps0 + k, ps0 + cur) | (best < 0);
                    best = take ? k : best;
                }
                if (best < 0) break;
                if (best < 64) taken_lo |= (1ull << best); else cx().scratch[first + (best - 64)] = 1;
                int s = ps0 + best;
                int avail = 0; for (int i = 0; i < np; i++) { int p = cx().j_pods_sorted[first + i]; if (cx().p_podset[p] == s && should_allocate(p, real) && job_has_pod(j, p)) avail++; }
                if (avail == 0) continue;
                int max_tasks = ps_act(s) >= ps_min(s) ? (avail < 1 ? avail : 1) : (ps_min(s) - ps_act(s));  // getNumTasksToAllocate :145-153
                int got = 0;
                for (int i = 0; i < np && got < max_tasks; i++) { int p = cx().j_pods_sorted[first + i]; if (cx().p_podset[p] == s && should_allocate(p, real) && job_has_pod(j, p)) { cx().tta[first + out++] = p; got++; } }
                n_sg++;
            }
            if (nps > 64) for (int k = 64; k < nps; k++) cx().scratch[first + (k - 64)] = 0;
        }
        cx().j_tta_n[j] = out;
        double res[3] = {0, 0, 0};  // GetTasksToAllocateInitResource :88-113
        for (int i = 0; i < out; i++) { int p = cx().tta[first + i]; for (int k = 0; k < 3; k++) res[k] += k == KAI_Q_GPU ? ppend_gpu(p) : pquota(p, k); }  // GetTasksToAllocateInitResource :88-113
        for (int k = 0; k < 3; k++) cx().j_tta_res[(size_t)j * 4 + k] = res[k];
        cx().j_tta_valid[j] = 1;
    }

    // ------------------------------------------------------------------ proportion: queue order + capacity
    KAI_HD double dominant_share(int q, const double* add) const {  // resource_share/queue_resource_share.go:142-166
        double dom = 0.0;
        for (int k = 0; k < 3; k++) {
            const QShare& s = cx().q_share[(size_t)q * 3 + k];
            double allocatable = qs_allocatable(s);
            if (allocatable == KAI_UNLIMITED) allocatable = k == 0 ? el().total0 : k == 1 ? el().total1 : el().total2;
            double allocated = s.allocated; if (add) allocated += add[k];
            double v = allocatable == 0 ? allocated * 1000 : allocated / allocatable;
            dom = kmax(dom, v);
        }
        return dom;
    }
    KAI_HD double dominant_share_l(const QShare* L, const double* add) const {  // the same on shares already in registers
        double dom = 0.0;
        for (int k = 0; k < 3; k++) {
            double allocatable = qs_allocatable(L[k]);
            if (allocatable == KAI_UNLIMITED) allocatable = k == 0 ? el().total0 : k == 1 ? el().total1 : el().total2;
            double allocated = L[k].allocated; if (add) allocated += add[k];
            double v = allocatable == 0 ? allocated * 1000 : allocated / allocatable;
            dom = kmax(dom, v);
        }
        return dom;
    }
    KAI_HD static int cmp_q(double a, double b) {  // resource_quantities.go:80-97
        if (a == KAI_UNLIMITED) return b == KAI_UNLIMITED ? 0 : 1;
        if (b == KAI_UNLIMITED) return -1;
        return a > b ? 1 : a < b ? -1 : 0;
    }
    // getBestJobFromNode :309-318 (pending ordering); a node whose key is valid already knows the best job of its subtree
    KAI_HD int best_job_from_node(int q) {
        for (;;) {
            const QNode& n = qnp()[q];
            if (n.flags & (QF_VALID | QF_TOP)) return n.best_job;
            if (n.flags & QF_LEAF) { int t = leaf_top(q); qnp()[q].best_job = t; qnp()[q].flags |= QF_TOP; return t; }
            if (n.len == 0) return -1;
            q = qheapp()[n.heap_off];
        }
    }
    // operands of queue_order.GetQueueOrderResult for queue q with the best job of its subtree, cached until q's shares or best job change
    KAI_HD void queue_key(int q) {
        if (qnp()[q].flags & QF_VALID) return;
#ifdef KAI_PROF_POP
        int64_t tk0 = be.clock(); el().h.prof[8]++;
#endif
#ifdef KAI_PROF_VPOP
        const int64_t tvk0 = be.clock(); el().h.prof[1]++;  // (victim-search profile of the queue pops: keys recomputed / their cycles in slot 6)
        struct TVK { Engine* e; int64_t t0; KAI_HD ~TVK() { e->el().h.prof[6] += e->be.clock() - t0; } } tvk{this, tvk0};
#endif
        const QShare L[3] = {cx().q_share[(size_t)q * 3], cx().q_share[(size_t)q * 3 + 1], cx().q_share[(size_t)q * 3 + 2]};  // loaded before the best-job chain: the two overlap
        int bj = best_job_from_node(q);
        double req[3] = {0, 0, 0};
        double sub[3] = {0, 0, 0}; bool victims = false;
        if constexpr (kVictim) if (el().jo_kind && bj >= 0) {  // getVictimsForQueue :336-346: the jobs popped from the leaf so far plus its next job
            victims = true; int leaf = cx().j_queue[bj]; double a[3]; view_allocated(bj, a);
            for (int k = 0; k < 3; k++) sub[k] = sx().vq_pop[leaf * 3 + k] + a[k];
        }
        if (bj >= 0 && !victims) {
            // cached chunk: flag and sums in one batch of loads (the sums are re-read only if the chunk had to be rebuilt; the victim search's queues pop hundreds of bystander
            // jobs per simulation, all with a valid chunk — the out-of-line ensure_tta only for the few that were touched)
            const int tv = cx().j_tta_valid[bj];
            req[0] = cx().j_tta_res[(size_t)bj * 4 + 0]; req[1] = cx().j_tta_res[(size_t)bj * 4 + 1]; req[2] = cx().j_tta_res[(size_t)bj * 4 + 2];
            if (!tv) { ensure_tta(bj, false); for (int k = 0; k < 3; k++) req[k] = cx().j_tta_res[(size_t)bj * 4 + k]; }
        }
        uint32_t bits = 0;
        bool over = true, starved = true, viol = false;
        for (int k = 0; k < 3; k++) {
            if (L[k].fair >= L[k].allocated) over = false;                    // prioritizeUnderUtilized :87-98 — FairShare.Less(Allocated) in all three
            double with_job = L[k].allocated + req[k];
            if (cmp_q(with_job, L[k].deserved) > 0) starved = false;          // prioritizeUnderQuotaWithJob :100-125
            if (qs_allocatable(L[k]) == 0 && with_job > 0) viol = true;       // penalizeZeroShareWithJob :127-176
        }
        if (over) bits |= QF_OVER; if (starved) bits |= QF_STARVED; if (viol) bits |= QF_VIOL;
        double dwj = dominant_share_l(L, req);  // :178-196, 242-273; the share without the job (:198-212) is computed on demand
        if constexpr (kVictim) if (victims) dwj = dominant_share_x(q, nullptr, sub);
        QNode& n = qnp()[q];
        n.best_job = bj; n.dom_with_job = dwj;
        n.flags = (n.flags & ~(QF_OVER | QF_STARVED | QF_VIOL | QF_DNJ)) | bits | QF_VALID;
#ifdef KAI_PROF_POP
        el().h.prof[9] += be.clock() - tk0;
#endif
    }
    KAI_HD double dom_no_job(int q) {
        QNode& n = qnp()[q];
        if (!(n.flags & QF_DNJ)) { n.dom_no_job = dominant_share(q, nullptr); n.flags |= QF_DNJ; }
        return n.dom_no_job;
    }
    // plugins/proportion/queue_order/queue_order.go:19-73 (allocate ordering: no victims)
    KAI_HD int queue_order(int lq, int rq) {
        queue_key(lq); queue_key(rq);
        const QNode kl = qnp()[lq]; const QNode kr = qnp()[rq];
        { bool lo = kl.flags & QF_OVER, ro = kr.flags & QF_OVER; if (!lo && ro) return -1; if (lo && !ro) return 1; }
        { bool ls = kl.flags & QF_STARVED, rs = kr.flags & QF_STARVED; if (ls && !rs) return -1; if (rs && !ls) return 1; }
        if (kl.prio > kr.prio) return -1;  // prioritizePrioritized :76-85
        if (kl.prio < kr.prio) return 1;
        { bool lv = kl.flags & QF_VIOL, rv = kr.flags & QF_VIOL; if (lv && !rv) return 1; if (!lv && rv) return -1; }
        if (kl.dom_with_job < kr.dom_with_job) return -1; if (kl.dom_with_job > kr.dom_with_job) return 1;
        {   // prioritizeSmallerResourceShareWithoutTask :198-212 — only reached on an exact tie above
            double l = dom_no_job(lq), r = dom_no_job(rq);
            if (l < r) return -1; if (l > r) return 1;
        }
        {   // prioritizeBasedOnAllocatableShare :214-224
            const QShare* L = &cx().q_share[(size_t)lq * 3]; const QShare* Rr = &cx().q_share[(size_t)rq * 3];
            bool l_le = true, r_le = true;
            for (int k = 0; k < 3; k++) { int cmp = cmp_q(qs_allocatable(L[k]), qs_allocatable(Rr[k])); if (cmp > 0) l_le = false; if (cmp < 0) r_le = false; }
            if (!r_le && l_le) return -1;  // l.LessInAtLeastOne(r) == !r.LessEqual(l)
            if (!l_le && r_le) return 1;
        }
        if (cx().q_created[lq] < cx().q_created[rq]) return -1;  // :235-240
        return 1;
    }
    KAI_HD bool queue_order_fn(int lq, int rq) {  // framework/session_plugins.go:283-299
        if (cx().plugins & KAI_PLUGIN_PROPORTION) { int v = queue_order(lq, rq); if (v != 0) return v < 0; }
        if (cx().q_created[lq] == cx().q_created[rq]) return cx().q_uid_rank[lq] < cx().q_uid_rank[rq];
        return cx().q_created[lq] < cx().q_created[rq];
    }
    KAI_HD bool over_limit(int j, const double* req) const {  // capacity_policy/max_allowed_check.go:20-66
        for (int q = cx().j_queue[j]; q >= 0; q = qnp()[q].parent) for (int k = 0; k < 3; k++) {
            const QShare& s = cx().q_share[(size_t)q * 3 + k];
            if (s.max_allowed == KAI_UNLIMITED) continue;
            if (req[k] == 0) continue;
            if (s.max_allowed < s.allocated + req[k]) return true;
        }
        return false;
    }
    KAI_HD bool np_over_quota(int j, const double* req) const {  // capacity_policy/quota_check.go:27-77
        if (cx().j_preempt[j]) return false;
        for (int q = cx().j_queue[j]; q >= 0; q = qnp()[q].parent) for (int k = 0; k < 3; k++) {
            const QShare& s = cx().q_share[(size_t)q * 3 + k];
            if (s.deserved == KAI_UNLIMITED) continue;
            if (req[k] == 0) continue;
            if (s.deserved < s.allocated_np + req[k]) return true;
        }
        return false;
    }
    KAI_HD bool job_over_queue_capacity(int j) const {  // capacity_policy.go:26-36,76-84
        if (!(cx().plugins & KAI_PLUGIN_PROPORTION)) return false;
        double req[3] = {0, 0, 0}; int first = cx().j_first_pod[j];
        for (int i = 0; i < cx().j_tta_n[j]; i++) { int p = cx().tta[first + i]; req[KAI_Q_GPU] += pquota(p, KAI_Q_GPU); req[KAI_Q_CPU] += pquota(p, KAI_Q_CPU); req[KAI_Q_MEM] += pquota(p, KAI_Q_MEM); }
        return over_limit(j, req) || np_over_quota(j, req);
    }
    KAI_HD bool task_over_capacity(int p) const {  // capacity_policy.go:51-61 with NodeInfo.GetRequiredInitQuota (node_info.go:734-744):
        if (!(cx().plugins & KAI_PLUGIN_PROPORTION)) return false;  // the GPU term is 1 for ANY whole-GPU request, 0 for CPU-only
        double req[3] = {preq(p, KAI_RES_CPU), preq(p, KAI_RES_MEM), preq(p, KAI_RES_GPU) >= 1 ? 1.0 : 0.0};
#ifdef KAI_SHARED_GPUS
        // a fraction: ceil(int64(portion * mem) / mem * 100) / 100 of a device; node independent because the host admits shared GPUs only when
        // every node has the same MemoryOfEveryGpuOnNode
        if (pod_shared(p) && cx().N > 0) { SgNode g{cx(), 0}; req[2] = g.frac_of(cx().p_mem[p]); }
        if (pod_kind(p) & KAI_KIND_MIG) req[2] = pquota_gpu(p);  // node_info.go:736-737
#endif
        int j = cx().p_job[p];
        return over_limit(j, req) || np_over_quota(j, req);
    }

    // ------------------------------------------------------------------ job-order tree (actions/utils/job_order_by_queue.go)
    // Leaf queues hold jobs under JobOrderFn, a strict total order (uid ranks are unique), so the pop sequence of the reference's
    // binary heap is the sorted sequence: a leaf is a sorted region with a cursor (built by k_leaf_init) plus a small array heap
    // for jobs that come back with a changed key (allocate.go:69-72).  Inner nodes order their children with the proportion
    // comparator, which is not a total order in every corner, so they stay array heaps with container/heap's exact sift rules
    // (scheduler_util/priority_queue.go) and the lazy needsReorder protocol.
    // leaf storage of the job-order tree: the context's arrays, or (victim search) those of the active instance
    KAI_HD auto lq_sorted() const { if constexpr (kVictim) return el().i_sorted; else return cx().lq_sorted; }
    KAI_HD auto lq_cur() const { if constexpr (kVictim) return el().i_cur; else return cx().lq_cur; }
    KAI_HD auto lq_end() const { if constexpr (kVictim) return el().i_end; else return cx().lq_end; }
    KAI_HD auto lq_side() const { if constexpr (kVictim) return el().i_side; else return cx().lq_side; }
    KAI_HD auto lq_side_len() const { if constexpr (kVictim) return el().i_side_len; else return cx().lq_side_len; }
    KAI_HD bool q_is_leaf(int q) const { return qnp()[q].flags & QF_LEAF; }
    KAI_HD int leaf_len_mem(int q) const { return (lq_end()[q] - lq_cur()[q]) + lq_side_len()[q]; }
    // simulation queues (instance 2): a leaf's sorted region is the shared template; jobs taken out of it for this simulation are stepped over
    KAI_HD void leaf_skip(int q) const {
        if constexpr (kVictim) if (el().cur_inst == 2) { const int off = cx().q_job_off[q]; while (lq_cur()[q] < lq_end()[q] && sx().ja_skip[lq_sorted()[off + lq_cur()[q]]]) lq_cur()[q]++; }
    }
    KAI_HD int leaf_top(int q) const {
        leaf_skip(q);
        int a = lq_cur()[q] < lq_end()[q] ? lq_sorted()[cx().q_job_off[q] + lq_cur()[q]] : -1;
        int b = lq_side_len()[q] > 0 ? lq_side()[cx().q_job_off[q]] : -1;
        if (a < 0) return b;
        if (b < 0) return a;
        return job_less(b, a) ? b : a;
    }
    struct JobLess { const Engine* e; KAI_HD bool operator()(int a, int b) const { return e->job_less(a, b); } };
    struct NodeLess { Engine* e; KAI_HD bool operator()(int a, int b) const { return e->node_less(a, b); } };
    template <class Less> KAI_HD void heap_up(int32_t* h, int j, Less less) {
        for (;;) { int i = (j - 1) / 2; if (i == j || !less(h[j], h[i])) break; int t = h[i]; h[i] = h[j]; h[j] = t; j = i; }
    }
    template <class Less> KAI_HD bool heap_down(int32_t* h, int i0, int n, Less less) {
        int i = i0;
        for (;;) {
            int j1 = 2 * i + 1; if (j1 >= n || j1 < 0) break;
            int j = j1, j2 = j1 + 1;
            if (j2 < n && less(h[j2], h[j1])) j = j2;
            if (!less(h[j], h[i])) break;
            int t = h[i]; h[i] = h[j]; h[j] = t; i = j;
        }
        return i > i0;
    }
    KAI_HD int leaf_pop(int q) {
#ifdef KAI_PROF_VPOP
        struct TVL { Engine* e; int64_t t0; KAI_HD ~TVL() { e->el().h.prof[23] += e->be.clock() - t0; } } tvl{this, be.clock()};
#endif
        leaf_skip(q);
        const int cur = lq_cur()[q], end = lq_end()[q], off = cx().q_job_off[q], nside = lq_side_len()[q];
        int a = cur < end ? lq_sorted()[off + cur] : -1;
        int a2 = cur + 1 < end ? lq_sorted()[off + cur + 1] : -1;  // the job behind it: same latency (usually the same line) as the head itself
        int b = nside > 0 ? lq_side()[off] : -1;
        qnp()[q].len--; qnp()[q].flags &= ~QF_TOP;
        if (b < 0 || (a >= 0 && !job_less(b, a))) {
            lq_cur()[q] = cur + 1;
            if constexpr (!kVictim) { if (b < 0) { qnp()[q].best_job = a2; qnp()[q].flags |= QF_TOP; } }  // no side heap: the next head is known now (leaf_top would fetch exactly this)
            return a;
        }
        int32_t* h = lq_side() + cx().q_job_off[q]; int n = lq_side_len()[q] - 1;
        int t = h[0]; h[0] = h[n]; h[n] = t; heap_down(h, 0, n, JobLess{this}); lq_side_len()[q] = n;
        return b;
    }
    KAI_HD void leaf_push(int q, int j) {
        int32_t* h = lq_side() + cx().q_job_off[q]; int n = lq_side_len()[q];
        if (n >= cx().q_job_off[q + 1] - cx().q_job_off[q]) { fault(FAULT_HEAP); return; }
        h[n] = j; lq_side_len()[q] = n + 1; heap_up(h, n, JobLess{this});
        qnp()[q].len++; qnp()[q].flags &= ~QF_TOP;
    }
    KAI_HD bool node_less(int l, int r) {  // buildNodeOrderFn :280-305
        if constexpr (kVictim) if (el().jo_kind) {  // reverseOrder
            if (qnp()[l].len == 0) return false;
            if (qnp()[r].len == 0) return true;
            return !queue_order_fn(l, r);
        }
        // both records in one batch of LDS reads; with both keys cached (the usual case inside a sift: only the node whose subtree was popped is
        // recomputed) the first five rules of queue_order.go:19-73 — over fair share, starved, priority, zero-share violation, dominant share with
        // the job — are one integer and one f64 compare on registers.  Ties, and keys still to be computed, take the full comparator.
        const QNode a = qnp()[l], b = qnp()[r];
        if (a.len == 0) return true;
        if (b.len == 0) return false;
        if ((cx().plugins & KAI_PLUGIN_PROPORTION) && (a.flags & b.flags & QF_VALID)) {
            const uint64_t ka = order_bits(a), kb = order_bits(b);
            if (ka != kb) return ka < kb;
            if (a.dom_with_job < b.dom_with_job) return true;
            if (a.dom_with_job > b.dom_with_job) return false;
        }
        return queue_order_fn(l, r);
    }
    KAI_HD static uint64_t order_bits(const QNode& n) {  // smaller sorts first: not over fair share, starved, higher priority, no violation
        return ((uint64_t)((n.flags & QF_OVER) != 0) << 34) | ((uint64_t)((n.flags & QF_STARVED) == 0) << 33) |
               ((uint64_t)(uint32_t)((int64_t)0x7fffffff - (int64_t)n.prio) << 1) | (uint64_t)((n.flags & QF_VIOL) != 0);
    }
    KAI_HD int32_t* node_heap(int parent) { return parent < 0 ? rootheapp() : qheapp() + qnp()[parent].heap_off; }
    KAI_HD int node_heap_len(int parent) const { return parent < 0 ? el().root_len : qnp()[parent].len; }
    KAI_HD void set_heap_len(int parent, int n) { if (parent < 0) el().root_len = n; else qnp()[parent].len = n; }
    KAI_HD void node_heap_push(int parent, int q) { int32_t* h = node_heap(parent); int n = node_heap_len(parent); h[n] = q; set_heap_len(parent, n + 1); heap_up(h, n, NodeLess{this}); if (parent >= 0) invalidate_path(parent); }
    KAI_HD void node_heap_pop(int parent) { int32_t* h = node_heap(parent); int n = node_heap_len(parent) - 1; set_heap_len(parent, n); int t = h[0]; h[0] = h[n]; h[n] = t; heap_down(h, 0, n, NodeLess{this}); if (parent >= 0) invalidate_path(parent); }
    KAI_HD void node_heap_fix0(int parent) {
#ifdef KAI_PROF_VPOP
        struct TVF { Engine* e; int64_t t0; KAI_HD ~TVF() { e->el().h.prof[12] += e->be.clock() - t0; e->el().h.prof[15]++; } } tvf{this, be.clock()};  // (heap fixes: cycles incl. the keys computed inside / count)
#endif
#ifdef KAI_PROF_POP
        int64_t tf0 = be.clock(); el().h.prof[11]++;
#endif
        int32_t* h = node_heap(parent); int n = node_heap_len(parent); if (!heap_down(h, 0, n, NodeLess{this})) heap_up(h, 0, NodeLess{this}); if (parent >= 0) invalidate_path(parent);
#ifdef KAI_PROF_POP
        el().h.prof[10] += be.clock() - tf0;
#endif
    }

    KAI_HD void ensure_ancestor_chain(int child) {  // :134-176
        for (;;) {
            int parent = qnp()[child].parent;
            if (parent < 0) { if (!(qnp()[child].flags & QF_LINKED)) { el().root_init = 1; qnp()[child].flags |= QF_LINKED; node_heap_push(-1, child); } return; }
            bool parent_new = !(qnp()[parent].flags & QF_EXISTS);
            if (parent_new) { qnp()[parent].flags = (qnp()[parent].flags & ~(QF_REORDER | QF_LINKED)) | QF_EXISTS; qnp()[parent].len = 0; }
            if (!(qnp()[child].flags & QF_LINKED)) { qnp()[child].flags |= QF_LINKED; node_heap_push(parent, child); }
            if (!parent_new) return;
            child = parent;
        }
    }
    KAI_HD void push_job(int j) {  // :91-120
        int q = cx().j_queue[j];
        if (!q_is_leaf(q)) return;
        bool needs_linking = !(qnp()[q].flags & QF_EXISTS);
        if (needs_linking) qnp()[q].flags = (qnp()[q].flags & ~(QF_REORDER | QF_LINKED)) | QF_EXISTS;
        leaf_push(q, j);
        invalidate_path(q);
        if (needs_linking) ensure_ancestor_chain(q);
        for (int x = q; x >= 0; x = qnp()[x].parent) qnp()[x].flags |= QF_REORDER;  // markAncestorsForReorder: parent pointers follow the queue tree
    }
    KAI_HD int next_node(int parent) {  // getNextNode :194-217
        for (;;) {
            if (node_heap_len(parent) == 0) return -1;
            int q = node_heap(parent)[0];
            if (qnp()[q].flags & QF_REORDER) { node_heap_fix0(parent); qnp()[q].flags &= ~QF_REORDER; continue; }
            if (qnp()[q].len == 0) return -1;
            return q;
        }
    }
    KAI_HD void handle_pop_from_node(int q) {  // :221-245
        for (;;) {
            if (qnp()[q].len != 0) { for (int x = q; x >= 0; x = qnp()[x].parent) qnp()[x].flags |= QF_REORDER; return; }
            int parent = qnp()[q].parent;
            node_heap_pop(parent);  // removeNodeFromParent: the node is at the top of its parent's heap
            qnp()[q].flags &= ~(QF_EXISTS | QF_LINKED);
            if (parent < 0) return;
            q = parent;
        }
    }
    KAI_HD int pop_next_job() {  // :61-89
        if (!el().root_init || el().root_len == 0) return -1;
        int parent = -1, q;
        for (;;) { q = next_node(parent); if (q < 0) return -1; if (q_is_leaf(q)) break; parent = q; }  // traverseToLeaf :179-191
        int job = leaf_pop(q);
        el().pop_leaf = q;
        if constexpr (!kVictim) { if (cx().use_index && cx().fast_ok) be.stage_async(cx(), job); }  // gathered by a service wave while the pop is finished
        invalidate_path(q);
        handle_pop_from_node(q);
        return job;
    }
    // InitializeWithJobs (actions/utils/input_jobs.go:21-68) in the canonical best-first order (see the oracle and DESIGN.md):
    // the leaves were filled by k_job_init / k_leaf_init; here every inner node links its best child first, then the others
    // in index order, deepest nodes first, so every node enters its parent's heap with its final key.
    KAI_HD void link_children(int x) {  // x = queue index, or Q for the virtual root
        int parent = x == cx().Q ? -1 : x;
        int b0 = cx().q_child_off[x], b1 = cx().q_child_off[x + 1], best = -1, live = 0;
        for (int i = b0; i < b1; i++) {
            int k = cx().q_children[i];
            if (qnp()[k].len == 0) continue;
            live++;
            if (best < 0 || node_less(k, best)) best = k;
        }
        if (!live) return;
        if (parent >= 0) qnp()[parent].flags |= QF_EXISTS | QF_REORDER; else el().root_init = 1;
        qnp()[best].flags |= QF_EXISTS | QF_LINKED | QF_REORDER; node_heap_push(parent, best);
        for (int i = b0; i < b1; i++) {
            int k = cx().q_children[i]; if (k == best) continue;
            if (qnp()[k].len == 0) continue;
            qnp()[k].flags |= QF_EXISTS | QF_LINKED | QF_REORDER;
            // the heap length of an inner node counts linked children only: children are appended as they are pushed
            node_heap_push(parent, k);
        }
    }
    KAI_HD void truncate_leaf(int q, int depth) {  // PriorityQueue.Push with a finite maxQueueSize under sorted pushes keeps the best `depth` jobs
        while (leaf_len_mem(q) > depth) {
            int a = lq_cur()[q] < lq_end()[q] ? lq_sorted()[cx().q_job_off[q] + lq_end()[q] - 1] : -1;
            int32_t* h = lq_side() + cx().q_job_off[q]; int n = lq_side_len()[q], wi = -1;
            for (int i = 0; i < n; i++) if (wi < 0 || job_order(h[wi], h[i])) wi = i;
            if (wi < 0 || (a >= 0 && job_order(h[wi], a))) { lq_end()[q]--; continue; }
            h[wi] = h[n - 1]; lq_side_len()[q] = n - 1;
            for (int i = (n - 1) / 2; i >= 0; i--) heap_down(h, i, n - 1, JobLess{this});
        }
        qnp()[q].len = leaf_len_mem(q); qnp()[q].flags &= ~QF_TOP;
    }
    KAI_HD void init_jobs_order() {  // qnp()[] comes from k_leaf_init: static fields, flags = QF_LEAF or 0, len = queued jobs of a leaf, 0 for inner nodes
        el().root_len = 0; el().root_init = 0;
        if (cx().queue_depth > 0) for (int q = 0; q < cx().Q; q++) if (q_is_leaf(q)) truncate_leaf(q, cx().queue_depth);
        for (int i = 0; i < cx().Q; i++) { int x = cx().q_depth_order[i]; if (!q_is_leaf(x)) link_children(x); }
        link_children(cx().Q);
    }

    // ------------------------------------------------------------------ actions/common/allocate.go
    KAI_HD void fill_req(ScanReq& q, int p) {
        q.pod = p; q.cpu_only = pod_cpu_only(p); q.best_effort = pod_best_effort(p); q.pod_class = cx().p_class[p]; q.nominated = cx().p_nominated[p];
        q.r_place = q.cpu_only ? KAI_RES_CPU : KAI_RES_GPU; q.strategy = q.cpu_only ? cx().cpu_strategy : cx().gpu_strategy;
        for (int r = 0; r < KAI_MAX_RES; r++) q.req[r] = r < cx().R ? preq(p, r) : 0.0;
        q.min_a = 0; q.max_a = 0;
#ifdef KAI_SHARED_GPUS
        q.portion = cx().shared_on ? (cx().p_portion[p] > 0 ? cx().p_portion[p] : (preq(p, KAI_RES_GPU) >= 1 ? 1.0 : 0.0)) : 0.0;  // GpuFractionalPortion()
        q.shared = pod_shared(p) ? 1 : 0; q.kind = pod_kind(p); q.gmem = q.shared ? cx().p_mem[p] : -1;  // GetResourceGpuMemory of a shared request (a gpu-memory request: portion 0, its own MiB)
#endif
    }
    // A decision without a class index (shared GPUs) is a pass over every node.  Between two decisions with the same request only a node or two changed, so the answer is kept
    // and patched: the arg-max over the UNCHANGED nodes is still the kept node (if it is unchanged) — or the kept node itself when it changed but its score did not drop, the
    // bin-packing case: a fuller node scores higher — and the changed nodes are evaluated one by one against it (score descending, index ascending, as the pass breaks ties).
    // A kept node whose score dropped leaves the others unknown: then the pass runs.  Same scores from the same function (scan_node_score), so the same node.
    KAI_HD static bool sim_dead_same(const ScanReq& a, const ScanReq& b) {  // everything the FILTERS of a pass read (the pre-order range only scales scores)
        if (a.cpu_only != b.cpu_only || a.best_effort != b.best_effort || a.pod_class != b.pod_class || a.nominated != b.nominated || a.r_place != b.r_place || a.strategy != b.strategy) return false;
        for (int r = 0; r < KAI_MAX_RES; r++) if (a.req[r] != b.req[r]) return false;
#ifdef KAI_SHARED_GPUS
        if (a.portion != b.portion || a.gmem != b.gmem || a.shared != b.shared || a.kind != b.kind) return false;
#endif
        return true;
    }
    KAI_HD static bool bn_same(const ScanReq& a, const ScanReq& b) {
        if (a.cpu_only != b.cpu_only || a.best_effort != b.best_effort || a.pod_class != b.pod_class || a.nominated != b.nominated || a.r_place != b.r_place || a.strategy != b.strategy) return false;
        for (int r = 0; r < KAI_MAX_RES; r++) if (a.req[r] != b.req[r]) return false;
        if (a.min_a != b.min_a || a.max_a != b.max_a) return false;
#ifdef KAI_SHARED_GPUS
        if (a.portion != b.portion || a.gmem != b.gmem || a.shared != b.shared || a.kind != b.kind) return false;
#endif
        return true;
    }
    KAI_HD static uint64_t bn_hash(const ScanReq& q) {  // any mix will do: equal requests have equal hashes, bn_same decides
        uint64_t h = 0x9E3779B97F4A7C15ull;
        auto mix = [&](uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); };
        mix((uint64_t)(uint32_t)q.cpu_only | ((uint64_t)(uint32_t)q.best_effort << 1) | ((uint64_t)(uint32_t)q.r_place << 2) | ((uint64_t)(uint32_t)q.strategy << 6) | ((uint64_t)(uint32_t)q.pod_class << 16) | ((uint64_t)(uint32_t)q.nominated << 40));
        for (int r = 0; r < KAI_MAX_RES; r++) { uint64_t b; const double v = q.req[r]; __builtin_memcpy(&b, &v, 8); mix(b); }
        { uint64_t b; double v = q.min_a; __builtin_memcpy(&b, &v, 8); mix(b); v = q.max_a; __builtin_memcpy(&b, &v, 8); mix(b); }
#ifdef KAI_SHARED_GPUS
        { uint64_t b; const double v = q.portion; __builtin_memcpy(&b, &v, 8); mix(b); mix((uint64_t)q.gmem); mix((uint64_t)(uint32_t)q.shared | ((uint64_t)(uint32_t)q.kind << 8)); }
#endif
        return h;
    }
    KAI_HD int best_node_kept(const ScanReq& q) {
        const uint64_t h = bn_hash(q);
        BnEnt* e = nullptr; int lru = 0;
        { BnEnt& x = eb().bn[el().bn_last]; if (x.valid && x.h == h && bn_same(x.q, q)) e = &x; }  // the tasks of a gang ask one after the other
        if (!e) for (int i = 0; i < KAI_BN_ENT; i++) {
            BnEnt& x = eb().bn[i];
            if (x.valid && x.h == h && bn_same(x.q, q)) { e = &x; el().bn_last = i; break; }
            if (!x.valid || (eb().bn[lru].valid && x.used < eb().bn[lru].used)) lru = i;
        }
        if (e && el().bn_seq - e->sync <= (uint32_t)KAI_BN_LOG) {
            int32_t* m = eb().bn_m; double* sc = eb().bn_sc; uint8_t* ok = eb().bn_ok; int nm = 0;
            for (uint32_t i = e->sync; i != el().bn_seq; i++) { const int x = eb().bn_log[i % KAI_BN_LOG]; bool seen = false; for (int k = 0; k < nm; k++) if (m[k] == x) seen = true; if (!seen) m[nm++] = x; }
            if (nm == 0 || be.eval_nodes(cx(), q, m, nm, sc, ok)) {
                int best = e->node; double bs = e->score; bool keep = true;
                for (int k = 0; k < nm; k++) if (m[k] == e->node) { if (ok[k] && sc[k] >= e->score) bs = sc[k]; else keep = false; }
                if (keep) {
                    for (int k = 0; k < nm; k++) if (m[k] != e->node && ok[k] && (best < 0 || sc[k] > bs || (sc[k] == bs && m[k] < best))) { best = m[k]; bs = sc[k]; }
                    e->node = best; e->score = bs; e->sync = el().bn_seq; e->used = ++el().bn_tick;
                    el().h.index_queries++;
                    return best;
                }
            }
        }
        if (!e) { e = &eb().bn[lru]; el().bn_last = lru; }
        double s = 0; const int n = be.best_node(cx(), q, &s);
        cx().st->node_scans++; cx().st->nodes_scanned += cx().N;
        e->q = q; e->h = h; e->node = n; e->score = s; e->valid = 1; e->sync = el().bn_seq; e->used = ++el().bn_tick;
        return n;
    }
    // OrderedNodesByTask + FittingNode for one task (framework/session.go:201-264): the first fitting node in score order, or -1
    KAI_HD int find_node(int p, bool& allocatable) {
        int k = (cx().use_index && !el().scope_bits && el().scope_row < 0) ? cx().p_scls[p] : -1;
        if (k >= 0) {
            const ClassRec& cr = cx().cls[k];
            int n = -1; uint64_t key = 0;
            int nom = (cx().plugins & KAI_PLUGIN_NOMINATEDNODE) ? cx().p_nominated[p] : -1;
            if (nom >= 0) { key = class_key(cx(), cr, nom); if (key) n = nom; }  // +1e6 outranks every other sum (plugins/nominatednode/nominatednode.go:29-41)
            if (n < 0) { flush_index(); be.class_top(cx(), k, key, n); el().h.index_queries++; if (!key) n = -1; }
            if (n >= 0) allocatable = (cx().plugins & KAI_PLUGIN_NODEAVAILABILITY) ? ((key >> key_avail_bit(cx())) & 1) != 0 : (cr.best_effort || fits(cx(), cr.req, n, false));
            return n;
        }
        ScanReq q; fill_req(q, p);
        bool sim_scope = false;
        if constexpr (kVictim) {  // a simulation's feasible nodes, no narrower node set on top: has this request already found nothing since the last rollback?
            sim_scope = el().sim_dead_on && el().scope_bits && el().scope_bits == el().base_bits && el().scope_row < 0;
            if (sim_scope) for (int i = 0; i < el().sim_dead_n; i++) if (sim_dead_same(eb().sim_dead[i], q)) { cx().st->node_scans++; cx().st->nodes_scanned += cx().N; return -1; }  // (counted like the pass it stands for)
        }
        if ((cx().plugins & KAI_PLUGIN_NODEPLACEMENT) && q.strategy == KAI_BINPACK) preorder_range(q.r_place, q.min_a, q.max_a);  // NodePreOrderFn
        int n;
        if (cx().action == KAI_ACTION_ALLOCATE && !el().scope_bits && el().scope_row < 0) n = best_node_kept(q);  // over all nodes: the answer of the last decision with this request, patched
        else { n = be.best_node(cx(), q, nullptr); cx().st->node_scans++; cx().st->nodes_scanned += cx().N; }
        if constexpr (kVictim) if (sim_scope && n < 0) { const int k = el().sim_dead_n < 4 ? el().sim_dead_n++ : 3; eb().sim_dead[k] = q; }
        if (n >= 0) allocatable = q.best_effort || fits(cx(), q.req, n, false);  // NodeInfo.IsTaskAllocatable (node_info.go:168-188)
#ifdef KAI_SHARED_GPUS
        if (n >= 0 && pod_shared(p)) allocatable = q.best_effort || fits_shared(cx(), q, n, false);
#endif
        return n;
    }
    KAI_HD bool allocate_task(int p, bool pipeline_only) {  // :121-163
        el().h.decisions++;
        int64_t t0 = be.clock();
        // predicates step 1 is node independent on this path (capacity_policy.go:51-61): evaluate it once
        bool over = (cx().plugins & KAI_PLUGIN_PREDICATES) && task_over_capacity(p);
        int64_t t1 = be.clock(); el().h.prof[PF_TASKCAP] += t1 - t0;
        if (over) return false;
        bool allocatable = false;
        int n = find_node(p, allocatable);
        int64_t t2 = be.clock(); el().h.prof[PF_FIND] += t2 - t1;
        if (n < 0) { el().fail_no_node = true; return false; }
        // allocateTaskToNode :165-174
        bool ok;
#ifdef KAI_SHARED_GPUS
        if (pod_shared(p)) {  // gpu_sharing.AllocateFractionalGPUTaskToNode (gpuSharing.go:20-37, 85-103)
            SgNode g{cx(), n}; int slot = 0; bool releasing = false;
            if (!g.preferable(cx().p_mem[p], pipeline_only, allocatable, slot, releasing)) { fault(FAULT_INTERNAL); return false; }
            cx().p_group[p] = slot == KAI_WHOLE_GPU ? cx().next_new_group[0]++ : g.id(slot);
            const bool pipe = pipeline_only || releasing;
            ok = pipe ? stmt_pipeline(p, n, false) : stmt_allocate(p, n);
            if (!ok) cx().p_group[p] = -1;
        } else
#endif
        ok = (!pipeline_only && allocatable) ? stmt_allocate(p, n) : stmt_pipeline(p, n, !pipeline_only);
#if !defined(__HIP_DEVICE_COMPILE__) && defined(KAI_SOLVER_TRACE)
        std::fprintf(stderr, "[eng] task %d -> node %d pipe %d ok %d\n", p, n, (int)pipeline_only, (int)ok);
#endif
        el().h.prof[PF_STMT] += be.clock() - t2;
        return ok;
    }
    // ------------------------------------------------------------------ plugins/topology: SubsetNodesFn
    KAI_HD bool podset_in_group(int ps, int g) const { for (int x = cx().s_group[ps]; x >= 0; x = cx().g_parent[x]) if (x == g) return true; return false; }
    KAI_HD bool task_in_subgroup(int p, int grp, int ps) const { return ps >= 0 ? cx().p_podset[p] == ps : podset_in_group(cx().p_podset[p], grp); }
    KAI_HD static bool bits_has(KAI_GP(const uint32_t) bits, int n) { return !bits || ((bits[n >> 5] >> (n & 31)) & 1u); }
    KAI_HD int node_dom(int row, int n) const { return cx().node_domain[(size_t)row * cx().N + n]; }
    KAI_HD bool dom_in_subtree(int d, int root_dom) const {  // is d inside the sub-tree of root_dom (same topology)?
        int rl = cx().dom_level[root_dom];
        while (d >= 0 && cx().dom_level[d] > rl) d = cx().dom_parent[d];
        return d == root_dom;
    }
    KAI_HD double quantity_value_milli(double milli) const {  // resource.NewMilliQuantity(x).Value(): rounded up
        double q = milli / 1000.0, f = rd_floor_any(q);
        return f < q ? f + 1.0 : f;
    }
    KAI_HD static double rd_floor_any(double x) { double t = (double)(int64_t)x; return t > x ? t - 1.0 : t; }
    // getJobRatioToFreeResources (plugins/topology/job_filtering.go:491-524); tr = summed request of the tasks
    KAI_HD double job_ratio(const double* tr, int d) const {
        double dominant = 0.0;
        bool none = !(tr[KAI_RES_GPU] > 0) && !(tr[KAI_RES_CPU] > 0) && !(tr[KAI_RES_MEM] > 0);
        for (int r = KAI_RES_PODS; r < cx().R; r++) if (tr[r] > 0) none = false;   // a scalar the empty resource does not have
        if (none) return dominant;
        KAI_GP(const double) fr = cx().dom_free + (size_t)d * KAI_MAX_RES;
        if (tr[KAI_RES_GPU] > 0) dominant = kmax(dominant, tr[KAI_RES_GPU] / fr[KAI_RES_GPU]);
        {   double tv = quantity_value_milli((double)(int64_t)tr[KAI_RES_CPU]), fv = quantity_value_milli((double)(int64_t)fr[KAI_RES_CPU]);
            if (tv != 0) dominant = kmax(dominant, fv == 0 ? 1000.0 : tv / fv); }
        {   double tv = (double)(int64_t)tr[KAI_RES_MEM], fv = (double)(int64_t)fr[KAI_RES_MEM];
            if (tv != 0) dominant = kmax(dominant, fv == 0 ? 1000.0 : tv / fv); }
        for (int r = KAI_RES_PODS + 1; r < cx().R; r++) {  // the pods resource is ignored for bin-packing
            double tv = quantity_value_milli(tr[r]), fv = quantity_value_milli(fr[r]);
            if (tv != 0) dominant = kmax(dominant, fv == 0 ? 1000.0 : tv / fv);
        }
        return dominant;
    }
    KAI_HD bool domain_fits(int d, const double* tr, int count) const {  // checkJobDomainFit :362-379
        int ap = cx().dom_alloc_pods[d];
        if (ap != -1) return ap >= count;
        return !(job_ratio(tr, d) > 1.0);
    }
    // one domain of a TopoScan op 5..9 (the scan lanes of the action kernel run these; tests/host_sim runs them in a plain loop); returns what op 9 counts
    // is `p` a domain whose children sortTreeFromRoot re-orders?  The walk starts at t.domain (level t.dl) and does not descend below level t.lvl
    KAI_HD bool sort_visits(const TopoScan& t, int p) const {
        if (!dom_in_subtree(p, t.domain)) return false;
        return !(t.dl <= t.lvl && t.lvl < cx().dom_level[p]);
    }
    KAI_HD int topo_dom_body(const TopoScan& t, int d) const {
        const KaiCtx& c = cx();
        const int DT = c.D + c.T;
        if (t.op == 11 || t.op == 12) {  // d = a slot of the children table
            if (d >= c.dom_child_off[DT]) return 0;
            const int x = c.dom_children[d];
            if (c.dom_topo[x] != t.topo) return 0;
            const int p = c.dom_parent[x];
            if (t.op == 11) { if (sort_visits(t, p)) c.dom_children[d] = c.dom_tmp[d]; }  // the sorted order back (a slot only ever holds children of one parent)
            else c.dom_tmp[x] = d - c.dom_child_off[p];                                  // position among the siblings
            return 0;
        }
        if (c.dom_topo[d] != t.topo) { if (t.op == 13) c.dom_key[2 * (size_t)d + 1] = -100; return 0; }  // (op 14 counts over every entry of the table)
        switch (t.op) {
        case 5: c.dom_alloc_pods[d] = -1; c.dom_tmp[2 * DT + d] = 0; for (int r = 0; r < KAI_MAX_RES; r++) c.dom_free[(size_t)d * KAI_MAX_RES + r] = 0.0; return 0;
        case 16: {
            if (!dom_in_subtree(d, t.domain)) { for (int r = 0; r < KAI_MAX_RES; r++) c.dom_free[(size_t)d * KAI_MAX_RES + r] = 0.0; return 0; }
            // what the survey gathered is read coherently (other workgroups of a scan grid added to it behind this one's L2) and, for the sums, stored again for the plain loads that follow
            if (c.dom_level[d] == t.L - 1) for (int r = 0; r < KAI_MAX_RES; r++) c.dom_free[(size_t)d * KAI_MAX_RES + r] = be.coh_f64(&c.dom_free[(size_t)d * KAI_MAX_RES + r]);
            if (t.what & 2) c.dom_alloc_pods[d] = c.dom_level[d] == t.L - 1 ? be.coh_i32(&c.dom_tmp[2 * DT + d]) : 0;
            return 0; }
        case 6: {
            if (d >= c.D || c.dom_level[d] != t.lvl || !dom_in_subtree(d, t.domain)) return 0;
            const int par = c.dom_parent[d];
            if (t.what & 1) for (int r = 0; r < t.R; r++) be.add_f64((double*)&c.dom_free[(size_t)par * KAI_MAX_RES + r], c.dom_free[(size_t)d * KAI_MAX_RES + r]);  // integral amounts (exact_sums): any order of addition
            if (t.what & 2) be.add_i32((int32_t*)&c.dom_alloc_pods[par], c.dom_alloc_pods[d]);
            return 0; }
        case 7: if (dom_in_subtree(d, t.domain)) c.dom_alloc_pods[d] = 0; return 0;
        case 8: c.dom_ratio[d] = job_ratio(t.tr, d); return 0;
        case 9: {
            const int l = c.dom_level[d];
            c.dom_tmp[(c.D + c.T) + d] = 0;
            if (!((t.what >> (l + 1)) & 1)) return 0;
            if (!domain_fits(d, t.tr, t.tasks)) return 0;
            c.dom_tmp[(c.D + c.T) + d] = 1; return 1; }
        case 10: {  // sortTreeFromRoot: (ratio descending, ID ascending) is a total order on siblings, so a child's place is the number of siblings in front of it
            if (d >= c.D) return 0;
            const int p = c.dom_parent[d];
            if (!sort_visits(t, p)) return 0;
            const int b0 = c.dom_child_off[p], b1 = c.dom_child_off[p + 1];
            const double rx = c.dom_ratio[d]; const uint32_t ix = c.dom_id_rank[d];
            int r = 0;
            for (int i = b0; i < b1; i++) { const int y = c.dom_children[i]; if (y == d) continue; const double ry = c.dom_ratio[y]; if (ry > rx || (ry == rx && c.dom_id_rank[y] < ix)) r++; }
            c.dom_tmp[b0 + r] = d; return 0; }
        case 13: {  // the path from the root as a number with base DT digits: the order of the level-order walk inside one level; beside it the level of a chosen domain (else a level nothing has)
            int64_t key = 0, mul = 1;
            for (int a = d; a >= 0 && c.dom_level[a] >= 0; a = c.dom_parent[a]) { key += (int64_t)c.dom_tmp[a] * mul; mul *= DT; }
            c.dom_key[2 * (size_t)d] = key; c.dom_key[2 * (size_t)d + 1] = c.dom_tmp[DT + d] ? c.dom_level[d] : -100; return 0; }
        case 14: {  // sortDomainInfos: deepest level first, level-order inside a level — a chosen domain's place is the number of chosen ones in front of it
            if (!c.dom_tmp[DT + d]) return 0;
            const int64_t l = c.dom_level[d], k = c.dom_key[2 * (size_t)d];
            int r = 0;
            for (int e = 0; e < DT; e++) { const int64_t ke = c.dom_key[2 * (size_t)e], le = c.dom_key[2 * (size_t)e + 1]; r += (int)((le > l) | ((le == l) & (ke < k))); }
            ((KAI_GP(int32_t))t.out)[r] = d; return 0; }
        }
        return 0;
    }
    // subSetNodesFn (plugins/topology/job_filtering.go:34-112) for a sub-group (grp = SubGroupSet, or ps = pod-set) of job j over the node set
    // `parent`.  Writes the node sets in order to sets_out: a domain index, or -1 = "the parent set as it is".  Returns their number
    // (0 = no node set / configuration error).  Control-lane code over the HBM domain tables.
    KAI_HD int subset_nodes(int j, int key, int tc_topo, int tc_req, int tc_pref, int grp, int ps, const int32_t* chunk, int nt,
                            KAI_GP(const uint32_t) parent, KAI_GP(int32_t) sets_out) {
        const KaiCtx& c = cx();
        if (!(c.plugins & KAI_PLUGIN_TOPOLOGY)) { sets_out[0] = -1; return 1; }
        if (tc_topo == -2) return 0;  // "Requested topology does not exist"
        int tasks = 0; for (int i = 0; i < nt; i++) if (task_in_subgroup(chunk[i], grp, ps)) tasks++;
        if (tc_topo < 0 || tasks == 0) { sets_out[0] = -1; return 1; }
        const int t = tc_topo, row0 = c.topo_level_off[t], L = c.topo_level_off[t + 1] - row0, N = c.N, DT = c.D + c.T, root = c.D + t, R = c.R;
        // tasks: summed request, element-wise maximum, homogeneity (useRepresentorPodsAccounting :550-571)
        double tr[KAI_MAX_RES], mx[KAI_MAX_RES]; int scalar_users[KAI_MAX_RES], gpu_users = 0;
        for (int r = 0; r < KAI_MAX_RES; r++) { tr[r] = 0; mx[r] = 0; scalar_users[r] = 0; }
        for (int i = 0; i < nt; i++) {
            int p = chunk[i]; if (!task_in_subgroup(p, grp, ps)) continue;
            for (int r = 0; r < R; r++) { double v = preq(p, r); tr[r] += v; if (v > mx[r]) mx[r] = v; if (r >= KAI_RES_PODS && v != 0) scalar_users[r]++; }
            if (preq(p, KAI_RES_GPU) > 0) gpu_users++;
        }
        bool homogeneous = !(gpu_users != tasks && gpu_users != 0);
        for (int r = KAI_RES_PODS; r < R; r++) if (scalar_users[r] != 0 && scalar_users[r] != tasks) homogeneous = false;
        bool one_pod_only = !(mx[KAI_RES_CPU] > 0) && !(mx[KAI_RES_MEM] > 0) && !(mx[KAI_RES_GPU] > 0) && mx[KAI_RES_PODS] <= 1;
        for (int r = KAI_RES_PODS + 1; r < R; r++) if (mx[r] > 0) one_pod_only = false;
        // lowestCommonDomainID (common.go:17-67) over the nodes of `parent` that are part of the topology
        int domain = root, dl = -1;
        KAI_TCLK0
        TopoScan ts; ts.op = 1; ts.row0 = row0; ts.L = L; ts.domain = root; ts.root = root; ts.dl = 0; ts.R = R; ts.tasks = tasks; ts.one_pod = 0; ts.any = 0; ts.parent = parent; ts.out = nullptr;
        for (int r = 0; r < KAI_MAX_RES; r++) ts.mx[r] = 0;
        ts.topo = t; ts.lvl = 0; ts.what = 0; for (int r = 0; r < KAI_MAX_RES; r++) ts.tr[r] = 0;
        bool dom_lanes = false, surveyed = false;
        // the loops over the DOMAINS go to the scan lanes once there are enough of them (TopoScan ops 5..16, topo_dom_body); `dom_lanes` = they took the first one
        if (c.exact_sums && DT >= KAI_DOM_LANES_MIN) { ts.op = 5; dom_lanes = be.topo_scan(c, ts); }  // treeAllocatableCleanup :438-445
        if (dom_lanes && L > 0 && L <= KAI_TOPO_SCAN_LEVELS) {
            // one pass over the nodes for the three node loops of the function: the common domain of `parent`, Idle + Releasing per leaf domain, and (homogeneous tasks) the
            // pods every node accommodates per leaf domain — gathered for the whole topology, then cut down to the sub-tree of the common domain
            ts.op = 15; ts.what = homogeneous ? 2 : 0; ts.one_pod = one_pod_only ? 1 : 0; for (int r = 0; r < KAI_MAX_RES; r++) ts.mx[r] = mx[r];
            surveyed = be.topo_scan(c, ts);
        }
        if (surveyed) {
            for (int l = 0; l < L; l++) {
                if (!ts.any || ts.lvl_min[l] != ts.lvl_max[l]) break;
                domain = ts.lvl_min[l];
                if (tc_pref == l) break;
            }
            dl = c.dom_level[domain];
            KAI_TCLK(21)
            ts.op = 16; ts.domain = domain; ts.dl = dl; be.topo_scan(c, ts);
            for (int lvl = L - 1; lvl > dl; lvl--) { ts.op = 6; ts.lvl = lvl; ts.what = homogeneous ? 3 : 1; be.topo_scan(c, ts); }  // calcSubTreeFreeResources :192-211 + calcTreeAllocatable :138-190, one level at a time
            KAI_TCLK(22)
        } else {
        ts.op = 1;
        const bool scan_lanes = L <= KAI_TOPO_SCAN_LEVELS && c.exact_sums && be.topo_scan(c, ts);
        if (scan_lanes) {
            for (int l = 0; l < L; l++) {
                if (!ts.any || ts.lvl_min[l] != ts.lvl_max[l]) break;
                domain = ts.lvl_min[l];
                if (tc_pref == l) break;
            }
        } else for (int l = 0; l < L; l++) {
            int v = -2; bool all = true;
            for (int n = 0; n < N && all; n++) {
                if (!bits_has(parent, n) || node_dom(row0, n) < 0) continue;
                int dd = node_dom(row0 + l, n);
                if (v == -2) v = dd; else if (dd != v) all = false;
            }
            if (v == -2 || !all) break;
            domain = v;
            if (tc_pref == l) break;
        }
        dl = c.dom_level[domain];
        KAI_TCLK(21)
        // treeAllocatableCleanup :438-445 + calcSubTreeFreeResources :192-211 (leaf accumulation, then bottom-up inside the sub-tree)
        if (!dom_lanes) for (int d = 0; d < DT; d++) if (c.dom_topo[d] == t) { c.dom_alloc_pods[d] = -1; for (int r = 0; r < KAI_MAX_RES; r++) c.dom_free[(size_t)d * KAI_MAX_RES + r] = 0.0; }
        auto node_in_domain = [&](int n) { return L > 0 && (domain == root ? node_dom(row0, n) >= 0 : node_dom(row0 + dl, n) == domain); };
        ts.op = 2; ts.domain = domain; ts.dl = dl;
        if (!(scan_lanes && be.topo_scan(c, ts))) for (int n = 0; n < N; n++) {
            if (!node_in_domain(n)) continue;
            int leaf = node_dom(row0 + L - 1, n);
            for (int r = 0; r < R; r++) { size_t x = (size_t)leaf * KAI_MAX_RES + r; c.dom_free[x] += c.n_idle[(size_t)r * N + n]; c.dom_free[x] += c.n_rel[(size_t)r * N + n]; }
        }
        for (int lvl = L - 1; lvl > dl; lvl--) {
            if (dom_lanes) { ts.op = 6; ts.lvl = lvl; ts.what = 1; be.topo_scan(c, ts); continue; }
            for (int d = 0; d < c.D; d++) {
                if (c.dom_topo[d] != t || c.dom_level[d] != lvl || !dom_in_subtree(d, domain)) continue;
                int par = c.dom_parent[d];
                for (int r = 0; r < R; r++) c.dom_free[(size_t)par * KAI_MAX_RES + r] += c.dom_free[(size_t)d * KAI_MAX_RES + r];
            }
        }
        KAI_TCLK(22)
        if (homogeneous) {  // calcTreeAllocatable :138-190 with calcNodeAccommodation :213-246
            if (dom_lanes) { ts.op = 7; be.topo_scan(c, ts); }
            else for (int d = 0; d < DT; d++) if (c.dom_topo[d] == t && dom_in_subtree(d, domain)) c.dom_alloc_pods[d] = 0;
            ts.op = 3; ts.one_pod = one_pod_only ? 1 : 0; for (int r = 0; r < KAI_MAX_RES; r++) ts.mx[r] = mx[r];
            if (!(scan_lanes && be.topo_scan(c, ts))) for (int n = 0; n < N; n++) {
                if (!node_in_domain(n)) continue;
                int count = 0;
                if (one_pod_only) count = tasks;
                else {
                    double cur[KAI_MAX_RES]; for (int r = 0; r < KAI_MAX_RES; r++) cur[r] = mx[r];  // k-th test pod = k x the maximal pod, by repeated addition
                    for (;;) {
                        if (!fits(c, cur, n, true)) break;
                        count++;
                        for (int r = 0; r < R; r++) cur[r] += mx[r];
                    }
                }
                c.dom_alloc_pods[node_dom(row0 + L - 1, n)] += count;
            }
            for (int lvl = L - 1; lvl > dl; lvl--) {
                if (dom_lanes) { ts.op = 6; ts.lvl = lvl; ts.what = 2; be.topo_scan(c, ts); continue; }
                for (int d = 0; d < c.D; d++) {
                    if (c.dom_topo[d] != t || c.dom_level[d] != lvl || !dom_in_subtree(d, domain)) continue;
                    c.dom_alloc_pods[c.dom_parent[d]] += c.dom_alloc_pods[d];
                }
            }
        }
        }
        ts.domain = domain; ts.dl = dl;
        KAI_TCLK(23)
        if (!domain_fits(domain, tr, tasks)) return 0;
        // sortTreeFromRoot :447-486: children by ratio descending, then by domain ID, down to the preferred (else required) level
        const int max_depth = tc_pref >= 0 ? tc_pref : tc_req;
        KAI_GP(int32_t) stack = c.dom_tmp;
        if (max_depth >= 0) {
            bool ratios_ready = false;
            if (dom_lanes) { ts.op = 8; for (int r = 0; r < KAI_MAX_RES; r++) ts.tr[r] = tr[r]; ratios_ready = be.topo_scan(c, ts); }  // the ratio of every domain of the topology at once
            bool sorted = false;
            if (ratios_ready) {  // every child's place among its siblings at once, then the new order back into the table
                ts.op = 10; ts.lvl = max_depth;
                if (be.topo_scan(c, ts)) { ts.op = 11; be.topo_scan(c, ts); sorted = true; }
            }
            int sp = 0; if (!sorted) stack[sp++] = domain;
            while (sp > 0) {
                int d = stack[--sp];
                int b0 = c.dom_child_off[d], b1 = c.dom_child_off[d + 1];
                if (!ratios_ready) for (int i = b0; i < b1; i++) c.dom_ratio[c.dom_children[i]] = job_ratio(tr, c.dom_children[i]);
                for (int i = b0 + 1; i < b1; i++) {  // insertion sort: (ratio desc, ID asc) is a total order on siblings
                    int x = c.dom_children[i]; int k = i - 1;
                    while (k >= b0) {
                        int y = c.dom_children[k];
                        bool x_first = c.dom_ratio[x] > c.dom_ratio[y] || (c.dom_ratio[x] == c.dom_ratio[y] && c.dom_id_rank[x] < c.dom_id_rank[y]);
                        if (!x_first) break;
                        c.dom_children[k + 1] = y; k--;
                    }
                    c.dom_children[k + 1] = x;
                }
                if (c.dom_level[d] == max_depth) continue;
                for (int i = b0; i < b1; i++) stack[sp++] = c.dom_children[i];
            }
        }
        KAI_TCLK(24)
        if (tc_pref >= 0) {  // calculateNodeScores (node_scoring.go:37-69): i-th preferred-level domain in tree order scores floor((i+1)/n*10)*10000
            int slot = -1; for (int i = 0; i < el().n_keys; i++) if (c.sg_key[i] == key) slot = i;
            if (slot < 0) { if (el().n_keys >= KAI_TKEYS) { fault(FAULT_INTERNAL); return 0; } slot = el().n_keys++; c.sg_key[slot] = key; }
            c.sg_row[slot] = row0 + tc_pref;
            KAI_GP(double) sc = c.sg_score + (size_t)slot * DT;
            for (int d = 0; d < DT; d++) sc[d] = -1.0;
            for (int pass = 0, total = 0; pass < 2; pass++) {
                int sp = 0, idx = 0; stack[sp++] = domain;
                while (sp > 0) {
                    int d = stack[--sp];
                    if (c.dom_level[d] == tc_pref) { if (pass == 0) total++; else { double score = ((double)(idx + 1) / (double)total) * 10; sc[d] = rd_floor_any(score) * 10000.0; idx++; } continue; }
                    for (int i = c.dom_child_off[d + 1] - 1; i >= c.dom_child_off[d]; i--) stack[sp++] = c.dom_children[i];  // reverse push = in-order visit
                }
            }
        }
        KAI_TCLK(25)
        // getJobAllocatableDomains :265-310 over calculateRelevantDomainLevels :381-425 (from the preferred level up to the required one)
        if (tc_req < 0 && tc_pref < 0) return 0;
        if (tc_req >= L || tc_pref >= L) return 0;  // a level the topology does not have
        bool restrict_active = false;
        if (tc_req >= 0) {  // hasActiveAllocatedTasks && required level: only sub-trees that already hold an active pod of these pod-sets (:321-347)
            for (int k = 0; k < c.j_n_ps[j]; k++) { int s2 = c.j_first_ps[j] + k; if ((ps >= 0 ? s2 == ps : podset_in_group(s2, grp)) && ps_act(s2) > 0) restrict_active = true; }
        }
        KAI_GP(int32_t) chosen = c.dom_tmp + DT;      // 1 = allocatable domain of a relevant level
        int n_chosen = 0;
        bool found_pref = false, found_req = false;
        bool chosen_done = false;
        if (dom_lanes && !restrict_active) {  // the relevant levels as a mask, every domain of them judged on the scan lanes
            int mask = 0; bool fp = false, fq = false;
            for (int l = L - 1; l >= -1; l--) { if (l == tc_pref && l >= 0) fp = true; if (l == tc_req && l >= 0) fq = true; if (fp || fq) mask |= 1 << (l + 1); if (fq) break; }
            ts.op = 9; ts.what = mask; ts.any = 0; for (int r = 0; r < KAI_MAX_RES; r++) ts.tr[r] = tr[r];
            if (be.topo_scan(c, ts)) { n_chosen = ts.any; chosen_done = true; }
        }
        if (!chosen_done) for (int d = 0; d < DT; d++) chosen[d] = 0;
        if (!chosen_done) for (int l = L - 1; l >= -1; l--) {
            if (l == tc_pref && l >= 0) found_pref = true;
            if (l == tc_req && l >= 0) found_req = true;
            if (found_pref || found_req) for (int d = 0; d < DT; d++) {
                if (c.dom_topo[d] != t || c.dom_level[d] != l) continue;
                if (restrict_active) {
                    int anc = d; while (anc >= 0 && c.dom_level[anc] > tc_req) anc = c.dom_parent[anc];
                    bool has = false;
                    if (anc >= 0 && c.dom_level[anc] == tc_req) for (int i = 0; i < c.j_n_pods[j] && !has; i++) {
                        int p = c.j_first_pod[j] + i;
                        if (!st_active_allocated(c.p_status[p]) || c.p_node[p] < 0 || !task_in_subgroup(p, grp, ps)) continue;
                        if (node_dom(row0 + tc_req, c.p_node[p]) == anc) has = true;
                    }
                    if (!has) continue;
                }
                if (domain_fits(d, tr, tasks)) { chosen[d] = 1; n_chosen++; }
            }
            if (found_req) break;
        }
        KAI_TCLK(26)
        if (!n_chosen) return 0;
        // sortDomainInfos :526-542: bottom-up level order of the tree from the topology root
        if (chosen_done && DT <= 4096) {  // on the scan lanes: positions among siblings, paths as numbers, a chosen domain's place = the chosen ones in front of it
            bool fits64 = true; int64_t m = 1;
            for (int l = 0; l < L; l++) { if (m > (int64_t)0x7fffffffffffffffll / DT) { fits64 = false; break; } m *= DT; }
            if (fits64) {
                ts.op = 12; be.topo_scan(c, ts);
                ts.op = 13; be.topo_scan(c, ts);
                ts.op = 14; ts.out = (KAI_GP(uint32_t))sets_out; be.topo_scan(c, ts);
                KAI_TCLK(27)
                return n_chosen;
            }
        }
        KAI_GP(int32_t) ord = c.dom_tmp + 2 * DT;
        int n_ord = 0, lvl_start[KAI_MAX_RES * 4], n_lvls = 0;
        ord[n_ord++] = root; lvl_start[n_lvls++] = 0;
        for (int b = 0; b < n_ord;) {
            int e = n_ord;
            for (int i = b; i < e; i++) for (int k = c.dom_child_off[ord[i]]; k < c.dom_child_off[ord[i] + 1]; k++) ord[n_ord++] = c.dom_children[k];
            b = e;
            if (n_ord > e) { if (n_lvls >= KAI_MAX_RES * 4) { fault(FAULT_INTERNAL); return 0; } lvl_start[n_lvls++] = e; }
        }
        int n_sets = 0;
        for (int lv = n_lvls - 1; lv >= 0; lv--) {
            int b = lvl_start[lv], e = lv + 1 < n_lvls ? lvl_start[lv + 1] : n_ord;
            for (int i = b; i < e; i++) if (chosen[ord[i]]) sets_out[n_sets++] = ord[i];
        }
        KAI_TCLK(27)
        return n_sets;
    }
    // node set of a frame: parent ∩ nodes of the topology ∩ nodes of domain d   (d = -1: the parent set itself)
    KAI_HD void build_node_set(KAI_GP(uint32_t) out, KAI_GP(const uint32_t) parent, int d) {
        const KaiCtx& c = cx();
        int row0 = 0, dl = -1;
        if (d >= 0) { row0 = c.topo_level_off[c.dom_topo[d]]; dl = c.dom_level[d]; }
        {   // N iterations on the control lane cost milliseconds at 64k nodes: the scan lanes build the words
            TopoScan ts; ts.op = 4; ts.row0 = row0; ts.L = 1; ts.domain = d; ts.root = -1; ts.dl = dl; ts.R = 0; ts.tasks = 0; ts.one_pod = 0; ts.any = 0; ts.parent = parent; ts.out = out;
            for (int r = 0; r < KAI_MAX_RES; r++) ts.mx[r] = 0;
            if (c.N >= 64 && be.topo_scan(c, ts)) return;
        }
        for (int w = 0; w < c.W; w++) out[w] = 0;
        for (int n = 0; n < c.N; n++) {
            bool in = bits_has(parent, n);
            if (in && d >= 0) in = dl < 0 ? node_dom(row0, n) >= 0 : node_dom(row0 + dl, n) == d;
            if (in) out[n >> 5] |= 1u << (n & 31);
        }
    }
    // the preferred-level scores that apply to a task: its pod-set's, else the nearest ancestor's (node_scoring.go:88-99)
    KAI_HD void set_scan_scope(int p, KAI_GP(const uint32_t) bits) {
        const KaiCtx& c = cx();
        el().scope_bits = bits; el().scope_row = -1; el().scope_score = nullptr;
        if (!(c.plugins & KAI_PLUGIN_TOPOLOGY) || el().n_keys == 0) return;
        int key = -(c.p_podset[p] + 1);
        for (;;) {
            for (int i = 0; i < el().n_keys; i++) if (c.sg_key[i] == key) { el().scope_row = c.sg_row[i]; el().scope_score = c.sg_score + (size_t)i * (c.D + c.T); return; }
            int parent = key < 0 ? c.s_group[-key - 1] : c.g_parent[key];
            if (parent < 0) return;
            key = parent;
        }
    }

    // ------------------------------------------------------------------ actions/common/allocate.go: the sub-group DFS
    // AllocateJob :20-36 → allocateSubGroupSet :38-60 → allocateSubGroupSetOnNodes :62-81 → allocatePodSet :83-107 →
    // allocateTasksOnNodeSet :109-119, as an explicit stack of frames (one per SubGroupSet / pod-set on the current path): each frame
    // walks its node sets in order, under a checkpoint, and a failing child sends its parent to the parent's next node set.
    struct Frame { int32_t kind /* 0 SubGroupSet, 1 pod-set */, id, n_sets, cur, cp, last_rank; uint64_t done; };
    KAI_HD bool allocate_job(int j, bool pipeline_only) {
        const KaiCtx& c = cx();
        if (c.j_n_ps[j] > 64) { fault(FAULT_INTERNAL); return false; }  // engine limit: 64 pod-sets per job
        el().n_keys = 0; el().restricted = 0;  // ssn.PreJobAllocation → topology.preJobAllocationFn (topology_plugin.go:52-55)
        int64_t t0 = be.clock();
        ensure_tta(j, !pipeline_only);
        int64_t t1 = be.clock(); el().h.prof[PF_TTA] += t1 - t0;
        bool gated = job_over_queue_capacity(j);
        el().h.prof[PF_GATE] += be.clock() - t1;
        if (gated) return false;
        const int first = c.j_first_pod[j], nt = c.j_tta_n[j], nps = c.j_n_ps[j], ps0 = c.j_first_ps[j], DT1 = c.D + c.T + 1;
        // the cached chunk is consumed below while statuses change, so snapshot it (Go holds the slice it got)
        int32_t* chunk = (int32_t*)(c.scratch + first);
        for (int i = 0; i < nt; i++) chunk[i] = c.tta[first + i];
        Frame fr[KAI_TDEPTH]; int depth = 0;
        fr[0].kind = 0; fr[0].id = c.j_root_group[j]; fr[0].n_sets = -1; fr[0].cur = -1; fr[0].cp = 0; fr[0].last_rank = -1; fr[0].done = 0;
        int ret = -1;  // result handed up by a finished child: -1 none, 0 failed, 1 succeeded
        bool result = false;
        for (;;) {
            Frame& f = fr[depth];
            KAI_GP(int32_t) sets = c.ns_sets + (size_t)depth * DT1;
            if (f.n_sets < 0) {  // first visit: SubsetNodesFn for this sub-group
                int topo = f.kind ? c.s_topo[f.id] : c.g_topo[f.id], req = f.kind ? c.s_req[f.id] : c.g_req[f.id], pref = f.kind ? c.s_pref[f.id] : c.g_pref[f.id];
#ifdef KAI_PROF_VICTIM
                const int64_t tsn0 = be.clock();
#endif
                f.n_sets = subset_nodes(j, f.kind ? -(f.id + 1) : f.id, topo, req, pref, f.kind ? -1 : f.id, f.kind ? f.id : -1, chunk, nt, el_parent_bits(depth), sets);
#ifdef KAI_PROF_VICTIM
                el().h.prof[20] += be.clock() - tsn0;
#endif
                f.cur = -1; ret = -1;
            }
            if (ret == 1) { ret = -1; }                       // a child succeeded: continue with the next child of this frame
            else {                                            // first visit, or a child failed: (roll back and) move to the next node set
                if (f.cur >= 0) { rollback(f.cp); }
                ret = -1;
                f.cur++;
                if (f.cur >= f.n_sets) { if (depth == 0) { result = false; break; } depth--; ret = 0; continue; }
                f.cp = checkpoint(); f.last_rank = -1; f.done = 0;
                { KAI_TCLK0 set_frame_bits(depth, sets[f.cur]); KAI_TCLK(28) }
            }
            if (f.kind == 1) {  // allocateTasksOnNodeSet :109-119
                bool ok = true;
                KAI_TCLK0
                for (int i = 0; i < nt; i++) {
                    int p = chunk[i]; if (c.p_podset[p] != f.id) continue;
                    set_scan_scope(p, frame_bits(depth));
                    if (!allocate_task(p, pipeline_only)) { ok = false; break; }
                }
                el().scope_bits = nullptr; el().scope_row = -1; el().scope_score = nullptr;
                KAI_TCLK(el_restricted(depth) ? 29 : 30)
                if (ok) { if (depth == 0) { result = true; break; } depth--; ret = 1; }
                else ret = 0;  // stay on this frame: next node set
                continue;
            }
            // SubGroupSet: child SubGroupSets by name (SubGroupSetOrderFn, session_plugins.go:273-282), then child pod-sets by PodSetOrderFn
            int next_kind = -1, next_id = -1;
            {   int best = -1; uint32_t best_rank = 0;
                for (int k = c.g_child_off[f.id]; k < c.g_child_off[f.id + 1]; k++) {
                    int g = c.g_children[k]; uint32_t rk = c.g_name_rank[g];
                    if ((int64_t)rk <= (int64_t)f.last_rank) continue;
                    if (best < 0 || rk < best_rank) { best = g; best_rank = rk; }
                }
                if (best >= 0) { next_kind = 0; next_id = best; f.last_rank = (int32_t)best_rank; } }
            if (next_kind < 0) {
                f.last_rank = 0x7fffffff;
                int best = -1;
                // (branch-free selection: hipcc 7.2 dropped the conditional `best = k` of the short-circuit form on gfx950)
                for (int k = 0; k < nps; k++) {
                    if ((f.done >> k) & 1) continue;
                    if (c.s_group[ps0 + k] != f.id) continue;
                    const int cur = best < 0 ? k : best;
                    const bool take = podset_order(ps0 + k, ps0 + cur) | (best < 0);
                    best = take ? k : best;
                }
                if (best >= 0) { f.done |= 1ull << best; next_kind = 1; next_id = ps0 + best; }
            }
            if (next_kind < 0) { if (depth == 0) { result = true; break; } depth--; ret = 1; continue; }  // every child placed
            if (depth + 1 >= KAI_TDEPTH) { fault(FAULT_INTERNAL); result = false; break; }
            depth++;
            fr[depth].kind = next_kind; fr[depth].id = next_id; fr[depth].n_sets = -1; fr[depth].cur = -1; fr[depth].cp = 0; fr[depth].last_rank = -1; fr[depth].done = 0;
            ret = -1;
        }
        el().scope_bits = nullptr; el().scope_row = -1; el().scope_score = nullptr;
        return result;
    }
    // node-set bitmaps of the DFS: slot d holds the set the frame at depth d is currently trying; `restricted[d]` tells whether any frame
    // up to depth d narrowed the set (otherwise the set is "every node" and scans may use the class index)
    KAI_HD KAI_GP(const uint32_t) frame_bits(int depth) { return el_restricted(depth) ? eb().fbits[depth] : (KAI_GP(const uint32_t))nullptr; }  // (the frame's own slot of ns_bits, or — a frame that narrows nothing — the set of the frame above it)
    KAI_HD KAI_GP(const uint32_t) el_parent_bits(int depth) { return depth == 0 ? el().base_bits : frame_bits(depth - 1); }
    KAI_HD bool el_restricted(int depth) const { return (el().restricted >> depth) & 1u; }
    KAI_HD void set_frame_bits(int depth, int dom) {
        bool parent_restricted = depth > 0 ? el_restricted(depth - 1) : (el().base_bits != nullptr);
        if (dom < 0 && !parent_restricted) { el().restricted &= ~(1u << depth); return; }  // still every node
        if (dom < 0) { eb().fbits[depth] = el_parent_bits(depth); el().restricted |= 1u << depth; return; }  // the parent's set as it is (a simulation's feasible nodes under every frame of every job it re-places): no copy
        build_node_set(cx().ns_bits + (size_t)depth * cx().W, el_parent_bits(depth), dom);
        eb().fbits[depth] = (KAI_GP(const uint32_t))(cx().ns_bits + (size_t)depth * cx().W);
        el().restricted |= 1u << depth;
    }
    // ------------------------------------------------------------------ staged job path
    // The same AllocateJob → allocateTask → Statement.Allocate → Commit / Rollback sequence as above for the overwhelmingly common
    // job shape — one pod-set, pending pods of indexed classes, nothing releasing or pipelined in the session (so every fitting
    // node is allocatable now and nothing is pipelined) — with the job's working set staged in LDS: the shares of the queue chain
    // and the chunk's request vectors are loaded once, every f64 operation of the reference is applied to the staged copy in the
    // reference's order (allocate handlers :443-465, rollback = the same operations with the opposite sign, Allocated→Binding on
    // commit), and the results are written back once.  Returns -1 when the job does not qualify (nothing touched).
    KAI_HD bool frame_over_limit(const FastFrame& f, const double* req) const {  // capacity_policy/max_allowed_check.go:20-66
        for (int i = 0; i < f.n_lim; i++) {  // an unlimited entry never trips (:38-40): only the finite ones were listed when the chain was loaded
            int l = f.lim_idx[i] / 3, k = f.lim_idx[i] % 3;
            if (req[k] == 0) continue;
            if (f.max_allowed[l][k] < f.alloc[l][k] + req[k]) return true;
        }
        return false;
    }
    KAI_HD bool frame_np_over_quota(const FastFrame& f, const double* req) const {  // capacity_policy/quota_check.go:27-77
        if (!f.np) return false;
        for (int i = 0; i < f.n_dsv; i++) {
            int l = f.dsv_idx[i] / 3, k = f.dsv_idx[i] % 3;
            if (req[k] == 0) continue;
            if (f.deserved[l][k] < f.alloc_np[l][k] + req[k]) return true;
        }
        return false;
    }
    KAI_HD void frame_list_constraints(FastFrame& f, int d) const {
        int nl = 0, nd = 0;
        for (int l = 0; l < d; l++) for (int k = 0; k < 3; k++) {
            if (f.max_allowed[l][k] != KAI_UNLIMITED) f.lim_idx[nl++] = (uint8_t)(l * 3 + k);
            if (f.deserved[l][k] != KAI_UNLIMITED) f.dsv_idx[nd++] = (uint8_t)(l * 3 + k);
        }
        f.n_lim = nl; f.n_dsv = nd;
    }
    KAI_HD static double frame_quota(const double* rq, int k) { return k == KAI_Q_CPU ? rq[KAI_RES_CPU] : k == KAI_Q_MEM ? rq[KAI_RES_MEM] : rq[KAI_RES_GPU]; }
    KAI_HD int allocate_job_fast(int j) {
        if (!cx().use_index || !cx().fast_ok) return -1;
        int s, first, jq, jpre, nt; double ja0, ja1, ja2;
        FastFrame& f = KAI_FRAME;
        // the shares of the queue chain first: they hang off the leaf the job was popped from, not off anything of the job, so their loads overlap the
        // staging of the job by the service wave (and, on the sequential path, the job's own first loads)
        const int chain_leaf = el().pop_leaf; int d = 0;
        for (int q = chain_leaf; q >= 0; q = qnp()[q].parent) { if (d == KAI_FDEPTH) return -1; f.q[d++] = q; }
        for (int l = 0; l < d; l++) for (int k = 0; k < 3; k++) {
            const QShare& sh = cx().q_share[(size_t)f.q[l] * 3 + k];
            f.alloc[l][k] = sh.allocated; f.alloc_np[l][k] = sh.allocated_np; f.max_allowed[l][k] = sh.max_allowed; f.deserved[l][k] = sh.deserved;
        }
        frame_list_constraints(f, d);
        if (be.staged(j)) {  // a service wave gathered the job while the pop was being finished (JobPf)
            const JobPf& pf = KAI_JOBPF;
            s = pf.s; first = pf.first; jq = pf.jq; jpre = pf.jpre; nt = pf.tta_n; ja0 = pf.ja[0]; ja1 = pf.ja[1]; ja2 = pf.ja[2];
        } else {
            // every job-level field first: independent loads, one memory latency (the early exits below would serialise them)
            const int n_ps = cx().j_n_ps[j], has_topo = cx().j_has_topology[j];
            s = cx().j_first_ps[j]; first = cx().j_first_pod[j];
            const int tta_valid = cx().j_tta_valid[j], tta_n = cx().j_tta_n[j];
            jq = cx().j_queue[j]; jpre = cx().j_preempt[j];
            ja0 = cx().j_allocated[(size_t)j * 4 + 0]; ja1 = cx().j_allocated[(size_t)j * 4 + 1]; ja2 = cx().j_allocated[(size_t)j * 4 + 2];
            if (n_ps != 1 || has_topo) return -1;
            if (cx().s_pipelined[s] != 0) return -1;
            if (!tta_valid) ensure_tta(j, true);
            nt = tta_valid ? tta_n : cx().j_tta_n[j];
            if (nt <= 0 || nt > KAI_FMAX) return -1;
            const bool nominated = cx().plugins & KAI_PLUGIN_NOMINATEDNODE;
            for (int i = 0; i < nt; i++) f.p[i] = cx().tta[first + i];
            int bad = 0;
            for (int i = 0; i < nt; i++) {  // independent loads, checked afterwards: nothing here is conditional on an earlier load
                int p = f.p[i];
                int k = cx().p_scls[p], st = cx().p_status[p], on = cx().p_on_node[p], nom = cx().p_nominated[p];
                f.cls[i] = k;
                bad |= (k < 0) | (st != KAI_POD_PENDING) | (on >= 0) | (nominated && nom >= 0);
                for (int r = 0; r < KAI_MAX_RES; r++) f.req[i][r] = r < cx().R ? preq(p, r) : 0.0;
            }
            if (bad) return -1;
        }
#ifdef KAI_PROF_POP
        int64_t ts0 = be.clock();
#endif
        const bool prop = cx().plugins & KAI_PLUGIN_PROPORTION;
        if (jq != chain_leaf) {  // not reached from the allocate loop (a job is popped from its own queue); kept for any other caller
            d = 0;
            for (int q = jq; q >= 0; q = qnp()[q].parent) { if (d == KAI_FDEPTH) return -1; f.q[d++] = q; }
            for (int l = 0; l < d; l++) for (int k = 0; k < 3; k++) {
                const QShare& sh = cx().q_share[(size_t)f.q[l] * 3 + k];
                f.alloc[l][k] = sh.allocated; f.alloc_np[l][k] = sh.allocated_np; f.max_allowed[l][k] = sh.max_allowed; f.deserved[l][k] = sh.deserved;
            }
            frame_list_constraints(f, d);
        }
        f.depth = d; f.np = !jpre;
        if (prop) {  // IsJobOverQueueCapacityFn (capacity_policy.go:26-36,76-84)
            double req[3] = {0, 0, 0};
            for (int i = 0; i < nt; i++) { req[KAI_Q_GPU] += frame_quota(f.req[i], KAI_Q_GPU); req[KAI_Q_CPU] += frame_quota(f.req[i], KAI_Q_CPU); req[KAI_Q_MEM] += frame_quota(f.req[i], KAI_Q_MEM); }
            if (frame_over_limit(f, req) || frame_np_over_quota(f, req)) return 0;
        }
#ifdef KAI_PROF_POP
        el().h.prof[13] += be.clock() - ts0;  // frame: queue chain + gate
        int64_t ts1 = be.clock();
#endif
        double ja[3] = {ja0, ja1, ja2};
        const bool preds = cx().plugins & KAI_PLUGIN_PREDICATES;
        int done = 0; bool ok = true;
        int run_n = -1; uint64_t run_key = 0;
        for (int i = 0; i < nt; i++) {  // allocateTask :121-163
#ifdef KAI_PROF_LOOP
            int64_t L0 = be.clock();
#endif
            el().h.decisions++;
            const double* rq = f.req[i];
            if (preds && prop) {  // predicates step 1 (capacity_policy.go:51-61, node_info.go:734-744: 1 GPU for any whole-GPU request)
                double r3[3] = {rq[KAI_RES_CPU], rq[KAI_RES_MEM], rq[KAI_RES_GPU] >= 1 ? 1.0 : 0.0};
                if (frame_over_limit(f, r3) || frame_np_over_quota(f, r3)) { ok = false; break; }
            }
#ifdef KAI_PROF_LOOP
            int64_t L1 = be.clock(); el().h.prof[8] += L1 - L0;
#endif
            // A run of tasks of one class (round 6): the task before went to node run_n, the class's arg-max with key run_key, and since then only that node changed — by this
            // lane.  If its key for the class did not drop (bin-packing: a fuller node scores higher) and it still fits, no other node can have overtaken it (they are unchanged, and
            // among equal keys run_n was the lowest index): the task follows without waiting for the index, whose refresh is published once per run (k_fill's lazy follow, kai_batch_kernels.hpp).
            uint64_t key = 0; int n = -1;
            if (run_n >= 0 && f.cls[i] == f.cls[i - 1]) { const uint64_t kap = class_key(cx(), cx().cls[f.cls[i]], run_n); if (kap != 0 && kap >= run_key) { key = kap; n = run_n; } }
            if (n < 0) { flush_index(); be.class_top(cx(), f.cls[i], key, n); el().h.index_queries++; }
#ifdef KAI_PROF_LOOP
            int64_t L2 = be.clock(); el().h.prof[9] += L2 - L1;
#endif
            if (!key) { el().fail_no_node = true; ok = false; break; }
            // Statement.Allocate :297-358 → NodeInfo.AddTask → addTaskResources (node_info.go:457-493)
            bool next_fits = true;  // would one more task with this request still fit on the node (a hint for the follow step below; the class key decides)
            {   // all loads first (independent, one latency), then the stores: same values, same operations
                double u[KAI_MAX_RES], id[KAI_MAX_RES];
                for (int r = 0; r < KAI_MAX_RES; r++) if (r < cx().R) { size_t x = (size_t)r * cx().N + n; u[r] = cx().n_used[x]; id[r] = cx().n_idle[x]; }
                for (int r = 0; r < KAI_MAX_RES; r++) if (r < cx().R) {
                    double v = rq[r]; if (v == 0) continue;
                    size_t x = (size_t)r * cx().N + n;
                    cx().n_used[x] = u[r] + v; cx().n_idle[x] = id[r] - v;
                    if (v > id[r] - v) next_fits = false;
                }
            }
#ifdef KAI_PROF_LOOP
            int64_t L3 = be.clock(); el().h.prof[10] += L3 - L2;
#endif
            mark_dirty(n);
            // the refresh is published now — awaited by the next reader of the index, it overlaps the bookkeeping below, the commit and the next pop — unless the next task is
            // expected to follow onto this node (same class, the node has room for it, a strategy under which a fuller node does not score lower): then once, when the run ends or breaks
            const bool defer = i + 1 < nt && f.cls[i + 1] == f.cls[i] && next_fits && cx().cls[f.cls[i]].strategy != KAI_SPREAD;
            run_n = defer ? n : -1; run_key = key;
            if (!defer) flush_index();
#ifdef KAI_PROF_LOOP
            int64_t L4 = be.clock(); el().h.prof[11] += L4 - L3;
#endif
            if (prop) {  // proportion allocate handler :443-465 — unrolled over the levels: the LDS reads of one pass go out together
                const double v0 = frame_quota(rq, 0), v1 = frame_quota(rq, 1), v2 = frame_quota(rq, 2); const bool np = f.np;
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (int l = 0; l < KAI_FDEPTH; l++) {
                    if (l >= d) break;
                    f.alloc[l][0] += v0; f.alloc[l][1] += v1; f.alloc[l][2] += v2;
                    if (np) { f.alloc_np[l][0] += v0; f.alloc_np[l][1] += v1; f.alloc_np[l][2] += v2; }
                }
            }
            for (int k = 0; k < 3; k++) ja[k] += frame_quota(rq, k);  // PodGroupInfo.Allocated (job_info.go:208-226)
            f.node[i] = n; done++;
#ifdef KAI_PROF_LOOP
            el().h.prof[13] += be.clock() - L4;
#endif
        }
#ifdef KAI_PROF_POP
        el().h.prof[14] += be.clock() - ts1;  // task loop
#endif
        if (ok) {  // Statement.Commit :536-575 + ssn.BindPod: nt Allocate operations in task order
            for (int i = 0; i < nt; i++) {
                int p = f.p[i], n = f.node[i];
                if (el().h.out_len >= cx().out_cap) { fault(FAULT_OUT_CAP); break; }
                kai_op o; o.seq = el().h.out_len; o.kind = KAI_OP_ALLOCATE; o.pod = p; o.node = n; o.job = j; o.stmt = (int32_t)el().h.stmts; o.pad = 0;
                cx().out_ops[el().h.out_len++] = o;
                for (int k = 0; k < 3; k++) { double v = frame_quota(f.req[i], k); ja[k] -= v; ja[k] += v; }  // Allocated → Binding (job_info.go:228-287)
                cx().p_status[p] = KAI_POD_BINDING; cx().p_node[p] = n; cx().p_on_node[p] = n; cx().p_on_node_status[p] = KAI_POD_ALLOCATED; cx().p_accepted[p] = 1; cx().p_virtual[p] = 1;
            }
            cx().s_active_alloc[s] += nt; cx().s_active_used[s] += nt; cx().j_n_pending[j] -= nt;
            if (nt > 0) el().h.stmts++;
        } else {  // Statement.Rollback :48-61: the undone operations in reverse order, same arithmetic with the opposite sign
            for (int i = done - 1; i >= 0; i--) {
                const double* rq = f.req[i]; int n = f.node[i];
                for (int k = 0; k < 3; k++) ja[k] -= frame_quota(rq, k);
                {   double u[KAI_MAX_RES], id[KAI_MAX_RES];
                    for (int r = 0; r < KAI_MAX_RES; r++) if (r < cx().R) { size_t x = (size_t)r * cx().N + n; u[r] = cx().n_used[x]; id[r] = cx().n_idle[x]; }
                    for (int r = 0; r < KAI_MAX_RES; r++) if (r < cx().R) {
                        double v = rq[r]; if (v == 0) continue;
                        size_t x = (size_t)r * cx().N + n;
                        cx().n_used[x] = u[r] + -1.0 * v; cx().n_idle[x] = id[r] - -1.0 * v;
                    }
                }
                mark_dirty(n);
                cx().p_accepted[f.p[i]] = 1;  // AcceptedResource stays set after unallocate (node_info.go:746-766)
                if (prop) for (int l = 0; l < d; l++) for (int k = 0; k < 3; k++) {
                    double v = frame_quota(rq, k);
                    f.alloc[l][k] += -1.0 * v;
                    if (f.np) f.alloc_np[l][k] += -1.0 * v;
                }
            }
            el().h.rollbacks += 2;
        }
        if (done > 0) {
            cx().j_tta_valid[j] = 0;
            for (int k = 0; k < 3; k++) cx().j_allocated[(size_t)j * 4 + k] = ja[k];
            if (prop) for (int l = 0; l < d; l++) {
                for (int k = 0; k < 3; k++) { QShare& sh = cx().q_share[(size_t)f.q[l] * 3 + k]; sh.allocated = f.alloc[l][k]; sh.allocated_np = f.alloc_np[l][k]; }
                qnp()[f.q[l]].flags &= ~QF_VALID;
            }
        }
        return ok ? 1 : 0;
    }
    // What the allocate loop does with a job once no class has a fitting node (used by k_drain and the host twin): the job is
    // attempted, passes or fails the queue-capacity gate, and its first task finds no node.  Touches only job j's own cache.
    KAI_HD void drain_job(int j, int64_t& attempted, int64_t& decisions, int64_t& rollbacks) {
        attempted++;
        ensure_tta(j, true);
        if (job_over_queue_capacity(j)) return;
        if (cx().j_tta_n[j] == 0) return;
        decisions++; rollbacks += 2;
    }
    KAI_HD void hot_begin() {
        EngineHot& h = el().h; const EngineState& st = *cx().st;
        h.decisions = st.decisions; h.index_queries = st.index_queries; h.index_refreshes = st.index_refreshes; h.rollbacks = st.rollbacks;
        h.jobs_attempted = st.jobs_attempted; h.jobs_committed = st.jobs_committed; h.out_len = st.out_len; h.stmts = st.stmts;
        for (int i = 0; i < KAI_NPROF; i++) h.prof[i] = st.prof[i];
    }
    KAI_HD void hot_end() {
        const EngineHot& h = el().h; EngineState& st = *cx().st;
        st.decisions = h.decisions; st.index_queries = h.index_queries; st.index_refreshes = h.index_refreshes; st.rollbacks = h.rollbacks;
        st.jobs_attempted = h.jobs_attempted; st.jobs_committed = h.jobs_committed; st.out_len = h.out_len; st.stmts = h.stmts;
        for (int i = 0; i < KAI_NPROF; i++) st.prof[i] = h.prof[i];
    }
    KAI_HD void execute_allocate() { hot_begin(); execute_allocate_impl(); hot_end(); }
    KAI_HD void execute_allocate_impl() {  // actions/allocate/allocate.go:46-77
        int64_t t0 = be.clock(), t;
        be.hot(cx(), el().qn, el().qheap, el().root_heap);
        be.begin(cx());
        init_jobs_order();
        t = be.clock(); el().h.prof[5] += t - t0;
        for (;;) {
            if (cx().st->fault) break;
            int64_t ta = be.clock();
            int j = pop_next_job(); if (j < 0) break;
            int64_t tb = be.clock(); el().h.prof[0] += tb - ta;
            cx().st->ops_len = 0; cx().st->n_undo = 0;
            el().h.jobs_attempted++;
            el().fail_no_node = false;
            int fr = allocate_job_fast(j);
            bool ok = fr < 0 ? allocate_job(j, false) : fr == 1;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(KAI_ALLOC_TRACE)
            { static FILE* tf = std::getenv("KAI_ALLOC_TRACE") ? std::fopen(std::getenv("KAI_ALLOC_TRACE"), "w") : nullptr;
              if (tf) { int first = cx().j_first_pod[j]; int p0 = cx().j_pods_sorted[first]; std::fprintf(tf, "%d %d %d %d %d %d %lld\n", j, cx().j_queue[j], fr, (int)ok, (int)el().fail_no_node, cx().p_scls[p0], (long long)el().h.decisions); } }
#endif
            int64_t tc = be.clock(); el().h.prof[2] += tc - tb;
            if (ok) {  // attemptToAllocateJob :79-111 — ShouldPipelineJob (job_info.go:443-464)
                bool should_pipeline = false;
                for (int k = 0; k < cx().j_n_ps[j]; k++) { int s = cx().j_first_ps[j] + k; if (cx().s_pipelined[s] > 0 && (cx().s_active_alloc[s] - cx().s_pipelined[s]) < cx().s_min[s]) should_pipeline = true; }
                if (should_pipeline && !convert_all_allocated_to_pipelined(j)) ok = false;
            }
            if (ok) {
                el().h.jobs_committed++;
                commit();
                if (cx().j_n_pending[j] > 0) { int64_t tp = be.clock(); push_job(j); el().h.prof[PF_PUSH] += be.clock() - tp; }  // HasTasksToAllocate(job, true)
            } else {
                discard();
            }
            int64_t td = be.clock(); el().h.prof[3] += td - tc;
            if (!ok && el().fail_no_node && cx().use_index && cx().all_tracked) {  // nothing fits any class any more: the rest of the queue fails job by job
                flush_index();
                if (be.all_dead(cx())) { cx().st->drain_pending = 1; break; }
            }
            el().h.prof[4] += be.clock() - td;
        }
        el().h.prof[7] += be.clock() - t0;
    }

    // ------------------------------------------------------------------ reclaim / preempt / consolidation
#include "kai_engine_solver.inc"

};

// ======================================================================================================
// proportion fair-share division of ONE sibling set for ONE resource
// (plugins/proportion/resource_division/resource_division.go:33-42, 92-357).  Children are visited in index order
// (the reference ranges a Go map).  weight / rem_amt / rem_has are per-queue scratch of this resource.
// ======================================================================================================
KAI_HD double rd_remaining_requested(const QShare& s) {  // :317-325
    double requested = qs_requestable(s);
    if (requested < s.fair) return 0;
    return requested - s.fair;
}
KAI_HD bool rd_satisfied(const QShare& s) {  // :253-262
    if (s.request <= s.fair) return true;
    if (s.max_allowed != KAI_UNLIMITED && s.max_allowed <= s.fair) return true;
    return false;
}
KAI_HD double rd_floor(double x) {  // math.Floor for the non-negative finite values that occur here
    double t = (double)(int64_t)x;
    return t > x ? t - 1.0 : t;
}
// V: the sibling set as seen by the caller — share(i) = QShare of child i for the resource, prio / created / uid of child i, and three scratch slots per child
template <class V>
KAI_HD void divide_sibling_view(V& v, int nk, double total, double k_value) {
    // setDeservedResource :92-109
    double remaining = total;
    for (int i = 0; i < nk; i++) {
        QShare& s = v.share(i);
        double deserved = s.deserved; if (deserved == KAI_UNLIMITED) deserved = total;
        double amount = kmin(deserved, qs_requestable(s));
        s.fair += amount; remaining -= amount;
        v.rem_has(i) = 0;
    }
    if (!(remaining > 0)) return;
    // divideOverQuotaResource :111-144 — priorities descending
    const int64_t NO_PRIO = (int64_t)1 << 40;
    int64_t bound = NO_PRIO;
    for (;;) {
        int64_t p = -NO_PRIO;
        for (int i = 0; i < nk; i++) { int64_t qp = v.prio(i); if (qp < bound && qp > p) p = qp; }
        if (p == -NO_PRIO) break;
        bound = p;
        // divideUpToFairShare :164-222 on the queues of priority p
        double amount_left = remaining;
        for (;;) {
            bool another = false; double give_round = amount_left;
            double total_w = 0;  // getTotalWeightsForUnsatisfied :307-315
            for (int i = 0; i < nk; i++) { if (v.prio(i) != p) continue; const QShare& s = v.share(i); if (rd_remaining_requested(s) > 0) total_w += s.oqw; }
            double w_sum = 0.0;  // calcShareWeights :224-251
            if (total_w != 0) for (int i = 0; i < nk; i++) {
                if (v.prio(i) != p) continue; const QShare& s = v.share(i);
                if (rd_satisfied(s)) continue;
                double n_w = s.oqw / total_w;
                double w = kmax(0, n_w + k_value * (n_w - s.usage));
                v.weight(i) = w; w_sum += w;
            }
            if (w_sum == 0) break;
            for (int i = 0; i < nk; i++) {
                if (v.prio(i) != p) continue; QShare& s = v.share(i);
                if (amount_left == 0) break;
                if (rd_satisfied(s)) continue;
                double requested = rd_remaining_requested(s);
                if (s.oqw == 0) continue;
                double fair = give_round * (v.weight(i) / w_sum);
                double to_give = 0;  // getResourceToGiveInCurrentRound :283-305
                if (requested <= fair) { to_give = requested; v.rem_has(i) = 0; }
                else {
                    double rf = rd_floor(fair);
                    if (rf > 0) to_give = rf;
                    if (fair - to_give > 0) { v.rem_has(i) = 1; v.rem_amt(i) = fair - to_give; }
                }
                if (to_give == 0) continue;
                s.fair += to_give; amount_left -= to_give;
                another = another || requested < fair;
            }
            if (!another || amount_left == 0) break;
        }
        remaining = amount_left;
    }
    // second pass :129-142 — divideRemainingResource :264-281 per priority, largest remainder first
    bound = NO_PRIO;
    for (;;) {
        if (remaining <= 0) break;
        int64_t p = -NO_PRIO;
        for (int i = 0; i < nk; i++) { int64_t qp = v.prio(i); if (qp < bound && qp > p) p = qp; }
        if (p == -NO_PRIO) break;
        bound = p;
        for (;;) {
            if (remaining == 0) break;
            int best = -1;  // remainingRequestedOrderFn :337-357
            for (int i = 0; i < nk; i++) {
                if (v.prio(i) != p || !v.rem_has(i)) continue;
                if (best < 0) { best = i; continue; }
                if (v.rem_amt(i) > v.rem_amt(best)) { best = i; continue; }
                if (v.rem_amt(i) < v.rem_amt(best)) continue;
                if (v.created(i) != v.created(best)) { if (v.created(i) < v.created(best)) best = i; continue; }
                if (v.uid(i) < v.uid(best)) best = i;
            }
            if (best < 0) break;
            v.rem_has(best) = 0;
            double give = kmin(1, remaining);
            v.share(best).fair += give; remaining -= give;
        }
    }
}

// the sibling set straight out of the session arrays (host twin; sets too large for the staged kernel)
struct SiblingsGlobal {
    const KaiCtx& c; const int32_t* kids; int k; double* w; double* ra; uint8_t* rh;
    KAI_HD QShare& share(int i) const { return c.q_share[(size_t)kids[i] * 3 + k]; }
    KAI_HD int64_t prio(int i) const { return c.q_prio[kids[i]]; }
    KAI_HD int64_t created(int i) const { return c.q_created[kids[i]]; }
    KAI_HD uint32_t uid(int i) const { return c.q_uid_rank[kids[i]]; }
    KAI_HD double& weight(int i) const { return w[kids[i]]; }
    KAI_HD double& rem_amt(int i) const { return ra[kids[i]]; }
    KAI_HD uint8_t& rem_has(int i) const { return rh[kids[i]]; }
};
KAI_HD void divide_sibling_set(const KaiCtx& c, const int32_t* kids, int nk, int k, double total, double k_value,
                               double* weight, double* rem_amt, uint8_t* rem_has) {
    SiblingsGlobal v{c, kids, k, weight, rem_amt, rem_has};
    divide_sibling_view(v, nk, total, k_value);
}

// ======================================================================================================
// action init, per job and per leaf queue (run by k_job_init / k_leaf_init on the whole chip; host_sim runs them serially)
// ======================================================================================================
// InitializeWithJobs filters for the allocate action (actions/utils/input_jobs.go:24-63) + the job's elastic state:
// 0/1/2 = eligible with minAvailableState below/exactly/above, 3 = filtered out
KAI_HD uint8_t job_init_state(const KaiCtx& c, int j) {
    int q = c.j_queue[j];
    if (q < 0 || c.q_child_off[q + 1] != c.q_child_off[q]) return 3;  // queue missing or not a leaf
    if (c.j_n_pending[j] == 0) return 3;                               // FilterNonPending
    if (c.action == KAI_ACTION_CONSOLIDATION && !c.j_preempt[j]) return 3;  // FilterNonPreemptible (consolidation.go:41-46)
    bool exactly = true, below = false;
    for (int k = 0; k < c.j_n_ps[j]; k++) {
        int s = c.j_first_ps[j] + k;
        if (c.s_alive[s] - c.s_gated[s] < c.s_min[s]) return 3;       // FilterUnready (job_info.go:399-406, podset.go:114-120)
        int n = c.s_active_alloc[s], m = c.s_min[s];
        if (n < m) below = true; else if (n > m) exactly = false;
    }
    if (!(c.plugins & KAI_PLUGIN_ELASTIC)) return 0;                   // without the elastic plugin the state does not enter JobOrderFn
    return below ? 0 : exactly ? 1 : 2;
}

// job-order tree node of queue q at action start (k_leaf_init lane 0 / host_sim): static fields + queued jobs of a leaf
KAI_HD void qnode_init(const KaiCtx& c, int q, int queued) {
    QNode n; n.dom_with_job = 0; n.dom_no_job = 0; n.best_job = -1; n.prio = c.q_prio[q]; n.heap_off = c.q_child_off[q]; n.parent = c.q_parent[q];
    bool leaf = c.q_child_off[q + 1] == c.q_child_off[q];
    n.len = leaf ? queued : 0; n.flags = leaf ? QF_LEAF : 0;
    c.qn[q] = n;
}

}  // namespace kai
