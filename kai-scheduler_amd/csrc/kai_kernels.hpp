// kai_kernels.hpp — gfx950 kernels of the scheduling-cycle core (included by kai_core.hip only).
//
//  * session-open kernels: node accounting from the pods, proportion totals, queue usage roll-up
//    (segmented reductions pods → job → leaf queue → ancestors), fair-share division per tree level,
//    and the class-index build (one wavefront per 64-node block, every scan class);
//  * action-init kernels: per-job eligibility / elastic state, per-leaf-queue stable compaction into job order;
//  * the persistent action kernel: wave 0 / lane 0 drives kai::Engine, the other 15 wavefronts of the workgroup
//    keep the class index current (block refresh) and serve brute-force node scans out of the HBM-resident
//    node SoA (resource-major, so a wavefront reads 64 consecutive nodes of one resource = one 512-byte request);
//  * the drain kernel: once no class has a fitting node, the jobs still queued are resolved chip-wide.
//
// wave = 64 lanes; workgroup = 512 threads = 8 wavefronts (1 control + 7 service waves).
#pragma once
#include <hip/hip_runtime.h>

#include "kai_engine.hpp"
#include "kai_wave.hpp"

namespace kai {

constexpr int WG = 512;         // threads per workgroup of the action kernel (8 waves: the control lane's code wants > 128 VGPRs)
constexpr int WAVES = WG / 64;  // 8
constexpr int SCAN_LANES = WG - 64;
constexpr int SVC = WAVES - 1;  // service waves

// ------------------------------------------------------------------------------------------------------
// session open
// ------------------------------------------------------------------------------------------------------

// NodeInfo.AddTask for every active-used pod of the snapshot (api/node_info/node_info.go:384-493).
// Quantities are integers in float64 (milli-cores, bytes, devices, pod counts) so the f64 atomics commute exactly.
__global__ void k_node_accounting(KaiCtx c) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= c.P) return;
    int s = c.p_status[p], n = c.p_node[p];
    c.p_on_node[p] = -1; c.p_on_node_status[p] = 0; c.p_accepted[p] = 0; c.p_virtual[p] = 0;
    if (!st_active_used(s) || n < 0 || n >= c.N) return;
    c.p_on_node[p] = n; c.p_on_node_status[p] = s; c.p_accepted[p] = 1;
    for (int r = 0; r < c.R; r++) {
        double v = c.p_req[(size_t)r * c.P + p]; if (v == 0) continue;
        size_t i = (size_t)r * c.N + n;
        atomicAdd(&c.n_used[i], v);
        if (s == KAI_POD_RELEASING) { atomicAdd(&c.n_rel[i], v); atomicAdd(&c.n_idle[i], -v); }
        else if (s == KAI_POD_PIPELINED) atomicAdd(&c.n_rel[i], -v);
        else atomicAdd(&c.n_idle[i], -v);
    }
}

#ifdef KAI_SHARED_GPUS
// shared GPUs: NodeInfo.AddTask of a node's active pods in UID order by ONE lane per node — addSharedTaskResources' guards read what the
// pods before them left behind (api/node_info/gpu_sharing_node_info.go:83-136), so the order inside a node is part of the result
__global__ void k_pod_accounting_reset(KaiCtx c) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= c.P) return;
    c.p_on_node[p] = -1; c.p_on_node_status[p] = 0; c.p_accepted[p] = 0; c.p_virtual[p] = 0; c.p_on_group[p] = -1;
}
__global__ void k_node_accounting_shared(KaiCtx c, const int32_t* np_off, const int32_t* np_pods) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= c.N) return;
    for (int i = np_off[n]; i < np_off[n + 1]; i++) {
        const int p = np_pods[i], st = c.p_status[p];
        c.p_on_node[p] = n; c.p_on_node_status[p] = st; c.p_accepted[p] = 1;
        const bool frac = c.p_shared[p] != 0;
        if (frac && c.p_group[p] >= 0) c.p_on_group[p] = c.p_group[p];
        for (int r = 0; r < c.R; r++) {
            if (r == KAI_RES_GPU && frac) continue;  // getAcceptedTaskResourceWithoutSharedGPU :52-66
            const double v = c.p_req[(size_t)r * c.P + p]; if (v == 0) continue;
            const size_t x = (size_t)r * c.N + n;
            c.n_used[x] += v;
            if (st == KAI_POD_RELEASING) { c.n_rel[x] += v; c.n_idle[x] -= v; } else if (st == KAI_POD_PIPELINED) c.n_rel[x] -= v; else c.n_idle[x] -= v;
        }
        if (frac && c.p_on_group[p] >= 0) { SgNode g{c, n}; if (!g.add(st, c.p_mem[p], c.p_on_group[p])) { c.st->fault = FAULT_INTERNAL; c.st->fault_line = __LINE__; } }
    }
    { SgNode g{c, n}; g.refit(); }  // the class keys' summary bits of the node's groups (also clears what an earlier session of this snapshot left: kai_session_reset)
}
#endif

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// proportion.setTotalResources (plugins/proportion/proportion.go:252-288): Σ allocatable of ready nodes …
__global__ void k_total_nodes(KaiCtx c) {
    double acc[3] = {0, 0, 0};
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < c.N; n += gridDim.x * blockDim.x) {
        uint32_t f = c.n_flags[n];
        if (f & KAI_NODE_NOT_READY) continue;
        bool ignore_gpus = c.restrict_nodes && !(f & KAI_NODE_GPU_WORKER);
        acc[KAI_Q_CPU] += c.n_alloc[(size_t)KAI_RES_CPU * c.N + n];
        acc[KAI_Q_MEM] += c.n_alloc[(size_t)KAI_RES_MEM * c.N + n];
        if (!ignore_gpus) {  // QuantifyResource(Allocatable): GPUs + MIG instances by weight (resource_info.go:177-194)
            acc[KAI_Q_GPU] += c.n_alloc[(size_t)KAI_RES_GPU * c.N + n];
#ifdef KAI_SHARED_GPUS
            if (c.mig_on) for (int r = KAI_RES_PODS + 1; r < c.R; r++) if (c.res_mig_g[r] > 0) acc[KAI_Q_GPU] += (double)c.res_mig_g[r] * c.n_alloc[(size_t)r * c.N + n];
#endif
        }
    }
    for (int k = 0; k < 3; k++) { double v = wave_sum(acc[k]); if ((threadIdx.x & 63) == 0 && v != 0) atomicAdd(&c.st->total[k], v); }
}
// … minus the active pods of other schedulers on those nodes (:276-285)
__global__ void k_total_foreign(KaiCtx c) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= c.P) return;
    if (!(c.p_flags[p] & KAI_POD_FOREIGN_SCHEDULER)) return;
    int n = c.p_on_node[p]; if (n < 0) return;
    if (!st_active_used(c.p_on_node_status[p])) return;
    if (c.n_flags[n] & KAI_NODE_NOT_READY) return;
    atomicAdd(&c.st->total[KAI_Q_CPU], -c.p_req[(size_t)KAI_RES_CPU * c.P + p]);
    atomicAdd(&c.st->total[KAI_Q_MEM], -c.p_req[(size_t)KAI_RES_MEM * c.P + p]);
#ifdef KAI_SHARED_GPUS
    atomicAdd(&c.st->total[KAI_Q_GPU], -(c.quota_on ? c.p_quota_gpu[p] : c.p_req[(size_t)KAI_RES_GPU * c.P + p]));  // QuantifyResourceRequirements(ResReq)
#else
    atomicAdd(&c.st->total[KAI_Q_GPU], -c.p_req[(size_t)KAI_RES_GPU * c.P + p]);
#endif
}

// per-job sums + pod-set / job counters (api/podgroup_info/job_info.go:208-226, subgroup_info/podset.go:56-77).
// jsum[9][J]: (allocated, allocated_np, request) × (CPU, Memory, GPU) — proportion.updateQueuesCurrentResourceUsage :347-401
__global__ void k_job_usage(KaiCtx c, double* jsum) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c.J) return;
    double al[3] = {0, 0, 0}, rq[3] = {0, 0, 0}, ja[3] = {0, 0, 0};
    int pending = 0;
    for (int k = 0; k < c.j_n_ps[j]; k++) { int s = c.j_first_ps[j] + k; c.s_active_alloc[s] = 0; c.s_active_used[s] = 0; c.s_alive[s] = 0; c.s_gated[s] = 0; c.s_pipelined[s] = 0; }
    for (int i = 0; i < c.j_n_pods[j]; i++) {
        int p = c.j_first_pod[j] + i, s = c.p_status[p], ps = c.p_podset[p];
        if (st_active_allocated(s)) c.s_active_alloc[ps]++;
        if (st_active_used(s)) c.s_active_used[ps]++;
        if (st_alive(s)) c.s_alive[ps]++;
        if (s == KAI_POD_GATED) c.s_gated[ps]++;
        if (s == KAI_POD_PIPELINED) c.s_pipelined[ps]++;
        if (s == KAI_POD_PENDING) pending++;
        double q[3] = {c.p_req[(size_t)KAI_RES_CPU * c.P + p], c.p_req[(size_t)KAI_RES_MEM * c.P + p], c.p_req[(size_t)KAI_RES_GPU * c.P + p]};
        double qa = q[2], qp = q[2], qq = q[2];  // GPU quota of AcceptedResource / GPU weight of a pending request / quota of ResReq: they differ from ResReq.GPUs() for gpu-memory and MIG requests
#ifdef KAI_SHARED_GPUS
        if (c.quota_on) { qa = c.p_acc_gpu[p]; qp = c.p_pend_gpu[p]; qq = c.p_quota_gpu[p]; }
#endif
        if (st_allocated(s)) {
            for (int k = 0; k < 3; k++) ja[k] += k == 2 ? qq : q[k];  // PodGroupInfo.Allocated carries the MIG instances (job_info.go:231-251): QuantifyResource of it
            if (c.p_accepted[p]) for (int k = 0; k < 3; k++) { const double v = k == 2 ? qa : q[k]; al[k] += v; rq[k] += v; }  // AcceptedResource is empty for a pod no node holds
        } else if (s == KAI_POD_PENDING) {
            for (int k = 0; k < 3; k++) rq[k] += k == 2 ? qp : q[k];
        }
    }
    c.j_n_pending[j] = pending; c.j_tta_valid[j] = 0; c.j_tta_n[j] = 0;
    bool np = !c.j_preempt[j];
    for (int k = 0; k < 3; k++) {
        c.j_allocated[(size_t)j * 4 + k] = ja[k];
        jsum[(size_t)(0 + k) * c.J + j] = al[k]; jsum[(size_t)(3 + k) * c.J + j] = np ? al[k] : 0.0; jsum[(size_t)(6 + k) * c.J + j] = rq[k];
    }
}
// one wavefront per leaf queue: segmented reduction over the queue's jobs (jobs_static is CSR by q_job_off; the sum order is
// fixed by the lane stride, and the addends are integer-valued, so the result does not depend on it)
__global__ void k_leaf_usage(KaiCtx c, const double* jsum) {
    int q = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (q >= c.Q) return;
    int lane = threadIdx.x & 63, b = c.q_job_off[q], e = c.q_job_off[q + 1];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = b + lane; i < e; i += 64) { int j = c.jobs_static[i]; for (int x = 0; x < 9; x++) acc[x] += jsum[(size_t)x * c.J + j]; }
    for (int x = 0; x < 9; x++) acc[x] = wave_sum(acc[x]);
    if (lane == 0) for (int k = 0; k < 3; k++) {
        QShare& s = c.q_share[(size_t)q * 3 + k];
        s.allocated = acc[k]; s.allocated_np = acc[3 + k]; s.request = acc[6 + k]; s.fair = 0;
    }
}
// ancestors: level by level up the tree (leaf = height 0), one lane per (inner queue, field): a segmented reduction over the queue's children,
// added in child-index order like the sequential roll-up.  One workgroup; h_nodes / h_off list the queues by height (HostPrep::build_batch).
__global__ void k_tree_usage(KaiCtx c, const int32_t* h_off, const int32_t* h_nodes, int n_h) {
    for (int h = 1; h < n_h - 1; h++) {  // the virtual root (last height) holds no shares
        const int b = h_off[h], e = h_off[h + 1];
        for (int t = threadIdx.x; t < (e - b) * 9; t += blockDim.x) {
            const int x = h_nodes[b + t / 9], f = (t % 9) / 3, k = t % 3;
            if (x >= c.Q) continue;
            double acc = 0.0;
            for (int i = c.q_child_off[x]; i < c.q_child_off[x + 1]; i++) {
                const QShare& s = c.q_share[(size_t)c.q_children[i] * 3 + k];
                acc += f == 0 ? s.allocated : f == 1 ? s.allocated_np : s.request;
            }
            QShare& d = c.q_share[(size_t)x * 3 + k];
            if (f == 0) d.allocated += acc; else if (f == 1) d.allocated_np += acc; else d.request += acc;
        }
        __threadfence(); __syncthreads();
    }
}
// proportion.setFairShareForQueues (plugins/proportion/proportion.go:410-423), one tree level per launch (a level's totals are the parents' fair
// shares of the level above).  One wavefront per (sibling set, resource): the lanes stage the set — shares, priorities, creation times — into LDS,
// lane 0 runs the division (resource_division.go:26-357: rounds whose outcome feeds the next, sequential by construction) on the staged copy, the
// lanes write the fair shares back.  Sets of more than 64 queues run on the session arrays directly.
struct SiblingsLds {
    QShare* sh; int64_t* pr; int64_t* cr; uint32_t* ui; double* w; double* ra; uint8_t* rh;
    __device__ QShare& share(int i) const { return sh[i]; }
    __device__ int64_t prio(int i) const { return pr[i]; }
    __device__ int64_t created(int i) const { return cr[i]; }
    __device__ uint32_t uid(int i) const { return ui[i]; }
    __device__ double& weight(int i) const { return w[i]; }
    __device__ double& rem_amt(int i) const { return ra[i]; }
    __device__ uint8_t& rem_has(int i) const { return rh[i]; }
};
constexpr int FS_WAVES = 4, FS_MAX = 64;
__global__ void __launch_bounds__(FS_WAVES * 64) k_fair_share_level(KaiCtx c, const int32_t* lvl_parents, int first, int count, double* weight, double* rem_amt, uint8_t* rem_has) {
    __shared__ QShare s_sh[FS_WAVES][FS_MAX]; __shared__ int64_t s_pr[FS_WAVES][FS_MAX], s_cr[FS_WAVES][FS_MAX];
    __shared__ uint32_t s_ui[FS_WAVES][FS_MAX]; __shared__ double s_w[FS_WAVES][FS_MAX], s_ra[FS_WAVES][FS_MAX]; __shared__ uint8_t s_rh[FS_WAVES][FS_MAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = blockIdx.x * FS_WAVES + wave;
    if (t >= count * 3) return;
    const int par = lvl_parents[first + t / 3], k = t % 3;
    const double total = par == c.Q ? c.st->total[k] : c.q_share[(size_t)par * 3 + k].fair;
    const int32_t* kids = c.q_children + c.q_child_off[par];
    const int nk = c.q_child_off[par + 1] - c.q_child_off[par];
    if (nk > FS_MAX) { if (lane == 0) divide_sibling_set(c, kids, nk, k, total, c.k_value, weight + (size_t)k * c.Q, rem_amt + (size_t)k * c.Q, rem_has + (size_t)k * c.Q); return; }
    if (lane < nk) { const int q = kids[lane]; s_sh[wave][lane] = c.q_share[(size_t)q * 3 + k]; s_pr[wave][lane] = c.q_prio[q]; s_cr[wave][lane] = c.q_created[q]; s_ui[wave][lane] = c.q_uid_rank[q]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0) { SiblingsLds v{s_sh[wave], s_pr[wave], s_cr[wave], s_ui[wave], s_w[wave], s_ra[wave], s_rh[wave]}; divide_sibling_view(v, nk, total, c.k_value); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < nk) c.q_share[(size_t)kids[lane] * 3 + k].fair = s_sh[wave][lane].fair;
}

// ------------------------------------------------------------------------------------------------------
// class index
// ------------------------------------------------------------------------------------------------------
// arg-max of (key, node) over a wavefront; lanes are in ascending node / block order, ties go to the lowest lane (kai_wave.hpp)
__device__ __forceinline__ void wave_argmax(uint64_t& k, int& n) {
    unsigned long long kk = k; wave_argmax_first(kk, n); k = kk;
}
// L1 of the class index: one wavefront per 64-node block, every class (node state is read once per class from L1/L2)
__global__ void k_index_build(KaiCtx c) {
    int b = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (b >= c.NB) return;
    int lane = threadIdx.x & 63, n = b * KAI_BLOCK + lane;
    NodeRegs ns; load_node(c, n < c.N ? n : 0, ns);
    for (int k = 0; k < c.C; k++) {
        uint64_t key = n < c.N ? class_key_regs(c, c.cls[k], ns) : 0; int bn = n;
        wave_argmax(key, bn);
        if (lane == 0) { c.sum1_key[(size_t)k * c.NB + b] = key; c.sum1_node[(size_t)k * c.NB + b] = bn; }
    }
}

// ------------------------------------------------------------------------------------------------------
// action init
// ------------------------------------------------------------------------------------------------------
struct NullBackend {  // kernels that only need the engine's pure helpers
    static constexpr bool kVictim = false;
    template <class T> __device__ static void assume_tree(T*) {}
    const KaiCtx* cref = nullptr; EngineLocal loc;
    static constexpr bool kBig = false;  // (no scan lane ever reads the control lane's larger data; big() only has to exist)
    __device__ static EngineBig& big();
    __device__ void bind(const KaiCtx& c) { cref = &c; }
    __device__ const KaiCtx& ctx() const { return *cref; }
    __device__ EngineLocal& local() { return loc; }
    __device__ void minmax(const KaiCtx&, int, double&, double&) {}
    __device__ bool topo_scan(const KaiCtx&, TopoScan&) { return false; }
    __device__ bool pfor(const KaiCtx&, const PforReq&) { return false; }
    __device__ void or32(uint32_t* w, uint32_t bits) { atomicOr(w, bits); }
    __device__ static void add_f64(double* p, double v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }  // the roll-ups of subSetNodesFn: one workgroup, one L2
    __device__ static void add_i32(int32_t* p, int32_t v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ static double coh_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }  // coherent with the atomics of workgroups on other XCDs
    __device__ static int32_t coh_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ int best_node(const KaiCtx&, const ScanReq&, double* = nullptr) { return -1; }
    __device__ bool eval_nodes(const KaiCtx&, const ScanReq&, const int32_t*, int, double*, uint8_t*) { return false; }
    __device__ void begin(const KaiCtx&) {}
    __device__ bool dirty_add(int) { return true; }
    __device__ int dirty_count() { return 0; }
    __device__ void refresh(const KaiCtx&) {}
    __device__ void class_top(const KaiCtx&, int, uint64_t& k, int& n) { k = 0; n = -1; }
    __device__ bool all_dead(const KaiCtx&) { return false; }
    __device__ void stage_async(const KaiCtx&, int) {}
    __device__ bool staged(int) { return false; }
    __device__ void hot(const KaiCtx&, QNode*&, int32_t*&, int32_t*&) {}
    __device__ bool sim_tree(QNode*&, int32_t*&, int32_t*&) { return false; }
    __device__ int64_t clock() { return 0; }
};

// static fields of the job-order tree records (parent chain, priority, heap offsets): valid from session open on
__global__ void k_qnode_static(KaiCtx c) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < c.Q) qnode_init(c, q, 0);
}
__global__ void k_job_init(KaiCtx c) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c.J) return;
    uint8_t st = job_init_state(c, j);
    c.j_state[j] = st;
    // the tasks-to-allocate chunk of every queued job (allocation_info.go:27-113), which the queue comparator reads for the best
    // job of each queue: computed here chip-wide instead of lazily by the control lane.  Nothing is virtual before the first
    // statement of an action, so the chunk does not depend on isRealAllocation.
    if (st != 3 && c.j_n_ps[j] <= 64) { NullBackend nb; Engine<NullBackend> eng(c, nb); eng.ensure_tta(j, true); }
}
// One wavefront per leaf queue: stable compaction of the eligible "below minAvailable" jobs (host order: priority desc,
// creation, uid) into the leaf's sorted region; the few jobs in another elastic state go to the leaf's side heap.
__global__ void k_leaf_init(KaiCtx c) {
    int q = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (q >= c.Q) return;
    int lane = threadIdx.x & 63, b = c.q_job_off[q], e = c.q_job_off[q + 1];
    int cnt = 0, side = 0;
    const unsigned long long lt = (1ull << lane) - 1;
    for (int base = b; base < e; base += 64) {
        int i = base + lane, j = i < e ? c.jobs_static[i] : -1;
        int st = j >= 0 ? c.j_state[j] : 3;
        unsigned long long m0 = __ballot(st == 0), m12 = __ballot(st == 1 || st == 2);
        if (st == 0) c.lq_sorted[b + cnt + __popcll(m0 & lt)] = j;
        if (st == 1 || st == 2) c.lq_side[b + side + __popcll(m12 & lt)] = j;
        cnt += __popcll(m0); side += __popcll(m12);
    }
    __threadfence_block();
    if (lane == 0) {
        c.lq_cur[q] = 0; c.lq_end[q] = cnt; c.lq_side_len[q] = 0;
        qnode_init(c, q, cnt + side);
        if (side) {
            NullBackend nb; Engine<NullBackend> eng(c, nb);
            int32_t* h = c.lq_side + b;
            for (int i = 0; i < side; i++) { c.lq_side_len[q] = i + 1; eng.heap_up(h, i, typename Engine<NullBackend>::JobLess{&eng}); }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// the persistent action kernel
// ------------------------------------------------------------------------------------------------------
enum SvcCmd : int32_t { CMD_NONE = 0, CMD_MINMAX = 1, CMD_BEST = 2, CMD_EXIT = 3, CMD_BEGIN = 4, CMD_REFRESH = 5, CMD_LOADTREE = 6, CMD_STAGE = 7, CMD_TOPO = 8, CMD_PFOR = 9, CMD_EVAL = 10 };

// dynamic LDS of k_action / k_best_node: [s2_key C*NSB u64][s2_node C*NSB i32] and, when tree_in_lds, [QNode Q][qheap Q+1][root_heap Q+1]
extern __shared__ __align__(16) unsigned char kai_dyn_lds[];
__host__ __device__ inline size_t lds_index_bytes(int C, int NSB) { return (((size_t)C * NSB * 12) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t lds_tree_bytes(int Q) { return (size_t)Q * sizeof(QNode) + 2 * (size_t)(Q + 1) * 4 + 16; }

struct ActShared {
    ScanReq req;
    PforReq pfor;
    TopoScan topo; int32_t topo_min[WAVES][KAI_TOPO_SCAN_LEVELS], topo_max[WAVES][KAI_TOPO_SCAN_LEVELS], topo_any[WAVES];  // CMD_TOPO request and the waves' partial results
    int32_t eval_n, eval_nodes[KAI_BN_LOG]; uint8_t eval_ok[KAI_BN_LOG]; double eval_sc[KAI_BN_LOG];  // CMD_EVAL: scan_node_score of a few nodes for the request in `req` (Engine::best_node_kept)
    int32_t cmd, r, n_dirty, slice_hi;  // slice_hi: the node passes of the current command cover [0, slice_hi) here (N without a scan grid)
    int32_t dirty[KAI_MAXD];
    double part_min[WAVES], part_max[WAVES];
    unsigned long long part_key[WAVES];  // orderable score bits
    int32_t part_node[WAVES];
    unsigned long long top_key[KAI_CMAX]; int32_t top_node[KAI_CMAX];
    unsigned long long __attribute__((address_space(3)))* s2_key; int32_t __attribute__((address_space(3)))* s2_node;  // [C][NSB] in dynamic LDS (typed LDS pointers: ds_read / ds_write)
    QNode* qn; int32_t *qheap, *root_heap;          // job-order tree: dynamic LDS when it fits, else the HBM arrays
    int32_t tree_in_lds, in_flight;  // in_flight: 1 = a refresh, 2 = a job staging was published and its results not yet awaited (the control lane overlaps it with its own work)
    int32_t stage_job, pad3;
    KAI_GP(const uint32_t) nodeset;  // scope of brute-force scans: node-set bitmap (bit n of word n/32; engine node order), nullptr = all nodes,
    KAI_GP(const double) topo_score; int32_t topo_row, pad2;  // … and preferred-level topology scores per domain of level row topo_row (-1 = none)
    long long t_publish, t_wait, t_svc, t_seg[6];  // profiling: control lane through barrier 1 / barrier 2, service wave 1 busy time
};

// The scan grid of an allocate action on the sequential engine: beside the engine's workgroup, helper workgroups that take the passes over the NODES (pre-order range, best
// node of a decision without a class index — shared GPUs, node sets of the topology DFS —, the node loops of subSetNodesFn).  Workgroup g owns the nodes
// [g * sg_per, (g + 1) * sg_per); the control lane publishes a command here (payload, then seq with release at agent scope: the workgroups sit on different XCDs, whose L2s
// meet only through such accesses and fences), runs slice 0 on its own scan waves, waits for `done` and folds the helpers' partial results into its own.  One 64-node
// step of a 65 536-node pass per lane instead of 146.
constexpr int KAI_SG_MAX = 256;
struct ScanGridPart { unsigned long long key; int32_t node, any; double mn, mx; int32_t tmin[KAI_TOPO_SCAN_LEVELS], tmax[KAI_TOPO_SCAN_LEVELS]; int32_t pad[8]; };  // 128 bytes
struct ScanGrid {
    int32_t seq, done, cmd, r;  // seq: number of the command on the table (0 = none yet); done: helpers finished, summed over all commands
    int32_t topo_row, fault, ready, per;  // ready: number of the last command whose folded result is in `res`; per: nodes per slice
    int32_t xcc0, n_reg, checked, pad0;   // start-up: the engine's XCD + 1, helpers that signed on, workgroups that answered
    KAI_GP(const uint32_t) nodeset; KAI_GP(const double) topo_score;
    ScanReq req; TopoScan topo;
    ScanGridPart res;
    ScanGridPart part[KAI_SG_MAX];
};
__device__ __forceinline__ int sg_ld(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }  // coherent across the XCDs without touching the reader's L2
__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg(20 | (3 << 11)); }  // HW_REG_XCC_ID, bits 3:0

// the mailbox between the control lane and the service waves, the action's context and the engine's scalars: directly
// addressed LDS objects of the (single) workgroup
__shared__ ActShared g_sh;
__shared__ KaiCtx g_ctx;
__shared__ EngineLocal g_el;
__shared__ EngineBig g_eb;
__device__ inline EngineBig& NullBackend::big() { return g_eb; }  // (never dereferenced: kBig is false)

// monotone map f64 → u64 (larger double ⇒ larger key); scores here are finite and ≥ 0 but keep it general
__device__ __forceinline__ unsigned long long orderable(double d) {
    unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ double unorderable(unsigned long long k) {  // the inverse of orderable()
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

template <bool VICTIM, bool TREE_LDS = false>
struct DevBackendT {
    static constexpr bool kVictim = VICTIM;  // the victim search (reclaim / preempt / consolidation) is compiled into its own kernel
    // TREE_LDS: the job-order tree of this launch is in dynamic LDS (never assumed for the victim search, whose other instances are in HBM)
    template <class T> __device__ static void assume_tree(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (TREE_LDS && !VICTIM) __builtin_assume(__builtin_amdgcn_is_shared((const void*)p));
#else
        (void)p;
#endif
    }
    ActShared* const sh = &g_sh;  // every use below folds to a direct LDS address
    __device__ static void bind(const KaiCtx&) {}  // k_action copied the context into g_ctx before constructing the engine
    __device__ static const KaiCtx& ctx() { return g_ctx; }
    __device__ static EngineLocal& local() { return g_el; }
    static constexpr bool kBig = true;
    __device__ static EngineBig& big() { return g_eb; }
    // control lane side -------------------------------------------------------------------------------
    // A command is two workgroup barriers: publish, then results ready.  A refresh is split: refresh() publishes and returns, the
    // control lane goes on with work that neither reads the index nor touches the dirty list (commit, next pop, frame load), and
    // wait() passes the second barrier right before the next reader.  Node-state writes in between (a rollback) are followed by
    // their own mark_dirty + refresh, so a block evaluated mid-update is evaluated again before anyone reads it.
    __device__ void wait() {
        if (!sh->in_flight) return;
        long long t1 = clock64();
        __syncthreads();  // results ready
        sh->t_wait += clock64() - t1;
        if (sh->in_flight == 1) sh->n_dirty = 0;
        sh->in_flight = 0;
    }
    __device__ void call(int cmd) {
        wait();
        sh->cmd = cmd;
        __syncthreads();  // publish the command
        __syncthreads();  // results ready
    }
    __device__ void scope() { sh->nodeset = g_el.scope_bits; sh->topo_row = g_el.scope_row; sh->topo_score = g_el.scope_score; }
    // scan grid, control-lane side (never in the victim-search kernel, whose engines sit on replicas)
    __device__ static bool grid_on() { if constexpr (VICTIM) return false; else return g_ctx.sg_wgs > 1; }
    __device__ void node_pass(int cmd) {  // a command whose work is a pass over the nodes: on every workgroup of the grid at once
        if (!grid_on()) { sh->slice_hi = g_ctx.N; call(cmd); return; }
        wait();
        ScanGrid* g = g_ctx.sg;
        g->cmd = cmd; g->r = sh->r; g->nodeset = sh->nodeset; g->topo_row = sh->topo_row; g->topo_score = sh->topo_score;
        if (cmd == CMD_BEST) g->req = sh->req; else if (cmd == CMD_TOPO) g->topo = sh->topo;
#ifdef KAI_PROF_VICTIM
        const long long tg0 = clock64();
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the node state this lane changed since the last pass, and the payload: written back for the other XCDs
        __hip_atomic_store(&g->seq, ++g_el.sg_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef KAI_PROF_VICTIM
        const long long tg1 = clock64(); sh->t_seg[0] += tg1 - tg0; sh->t_seg[3]++;
#endif
        sh->slice_hi = g_ctx.sg_per < g_ctx.N ? g_ctx.sg_per : g_ctx.N;
        call(cmd);
#ifdef KAI_PROF_VICTIM
        const long long tg2 = clock64(); sh->t_seg[1] += tg2 - tg1;
#endif
        // the helpers' folded result: read with coherent loads only — an acquire fence here would empty this XCD's L2 under the control lane after every pass.  What the
        // helpers wrote elsewhere is never read here through a plain load first: bitmap words are read by the slice that wrote them, the survey's sums by op 16's coherent loads
        long long spins = 0;
        while (sg_ld(&g->ready) != g_el.sg_gen) {
            if (++spins > (1ll << 28)) { g_ctx.st->fault = FAULT_INTERNAL; g_ctx.st->fault_line = __LINE__; break; }  // a helper left the protocol
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef KAI_PROF_VICTIM
        sh->t_seg[2] += clock64() - tg2;
#endif
    }
    __device__ void minmax(const KaiCtx&, int r, double& mn, double& mx) {
        scope(); sh->r = r; node_pass(CMD_MINMAX);
        double lo = 1.7976931348623157e308, hi = 0;  // math.MaxFloat64, 0 (plugins/nodeplacement/pack.go:66-68)
        for (int w = 1; w < WAVES; w++) { if (sh->part_min[w] < lo) lo = sh->part_min[w]; if (sh->part_max[w] > hi) hi = sh->part_max[w]; }
        if (grid_on()) { const ScanGridPart& p = g_ctx.sg->res; const double a = __hip_atomic_load(&p.mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(&p.mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (a < lo) lo = a; if (b > hi) hi = b; }
        mn = lo; mx = hi;
    }
    __device__ bool pfor(const KaiCtx&, const PforReq& r) { if (r.n < 48) return false; sh->pfor = r; call(CMD_PFOR); return true; }  // only short loops stay on the control lane: two barriers cost ~2 k cycles, a body (victim filter, view state: a few dependent loads) ~300 per index on the lane
    __device__ void or32(uint32_t* w, uint32_t bits) { *w |= bits; }
    __device__ bool topo_scan(const KaiCtx&, TopoScan& t) {
        const bool grid = t.op == 1 || t.op == 4 || t.op == 15;  // passes over the nodes that go to the whole grid (ops 2 / 3 — only used without the survey — and the loops over domains: this workgroup's scan waves)
        sh->topo = t; if (grid) node_pass(CMD_TOPO); else { sh->slice_hi = g_ctx.N; call(CMD_TOPO); }
        if (t.op == 1 || t.op == 15) {
            int any = 0;
            for (int l = 0; l < t.L; l++) { int mn = 0x7fffffff, mx = -0x7fffffff - 1; for (int w = 1; w < WAVES; w++) { if (sh->topo_min[w][l] < mn) mn = sh->topo_min[w][l]; if (sh->topo_max[w][l] > mx) mx = sh->topo_max[w][l]; } t.lvl_min[l] = mn; t.lvl_max[l] = mx; }
            for (int w = 1; w < WAVES; w++) any |= sh->topo_any[w];
            if (grid_on()) {
                const ScanGridPart& p = g_ctx.sg->res;
                any |= sg_ld(&p.any);
                for (int l = 0; l < t.L; l++) { const int a = sg_ld(&p.tmin[l]), b = sg_ld(&p.tmax[l]); if (a < t.lvl_min[l]) t.lvl_min[l] = a; if (b > t.lvl_max[l]) t.lvl_max[l] = b; }
            }
            t.any = any;
        }
        if (t.op == 9) { int n = 0; for (int w = 1; w < WAVES; w++) n += sh->topo_any[w]; t.any = n; }  // chosen domains
        return true;
    }
    __device__ bool eval_nodes(const KaiCtx&, const ScanReq& q, const int32_t* nodes, int m, double* sc, uint8_t* ok) {  // a lane per node on this workgroup's scan waves
        wait();
        sh->req = q; sh->eval_n = m; for (int i = 0; i < m; i++) sh->eval_nodes[i] = nodes[i];
        call(CMD_EVAL);
        for (int i = 0; i < m; i++) { sc[i] = sh->eval_sc[i]; ok[i] = sh->eval_ok[i]; }
        return true;
    }
    __device__ int best_node(const KaiCtx&, const ScanReq& q, double* score_out = nullptr) {
        scope(); sh->req = q; node_pass(CMD_BEST);
        int best = -1; unsigned long long bk = 0;
        for (int w = 1; w < WAVES; w++) {
            int n = sh->part_node[w]; if (n < 0) continue;
            unsigned long long k = sh->part_key[w];
            if (best < 0 || k > bk || (k == bk && n < best)) { best = n; bk = k; }
        }
        if (grid_on()) {
            const ScanGridPart& p = g_ctx.sg->res;
            const int n = sg_ld(&p.node); const unsigned long long k = __hip_atomic_load(&p.key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n >= 0 && (best < 0 || k > bk || (k == bk && n < best))) { best = n; bk = k; }
        }
        if (score_out) *score_out = best >= 0 ? unorderable(bk) : 0.0;
        return best;
    }
    __device__ void begin(const KaiCtx& c) { if (c.use_index) call(CMD_BEGIN); }
    __device__ bool dirty_add(int b) {
        wait();  // the service waves read the list while a refresh is in flight
        int n = sh->n_dirty;
        for (int i = 0; i < n; i++) if (sh->dirty[i] == b) return true;
        if (n == KAI_MAXD) return false;
        sh->dirty[n] = b; sh->n_dirty = n + 1;
        return true;
    }
    __device__ int dirty_count() { return sh->in_flight ? 0 : sh->n_dirty; }
    __device__ void refresh(const KaiCtx&) {  // publish only; see wait()
        wait();
        sh->cmd = CMD_REFRESH; long long t0 = clock64();
        __syncthreads();
        sh->t_publish += clock64() - t0; sh->in_flight = 1;
    }
    __device__ void class_top(const KaiCtx&, int k, uint64_t& key, int& node) { wait(); key = sh->top_key[k]; node = sh->top_node[k]; }
    // job staging: published like a refresh, awaited by allocate_job_fast; skipped while blocks wait on the dirty list (the list belongs to the
    // next refresh)
    __device__ void stage_async(const KaiCtx&, int j) {
        wait();
        if (sh->n_dirty) { kai_pf_lds.job = -1; return; }
        sh->stage_job = j; sh->cmd = CMD_STAGE;
        __syncthreads();
        sh->in_flight = 2;
    }
    __device__ bool staged(int j) { if (sh->in_flight == 2) wait(); bool r = kai_pf_lds.job == j && kai_pf_lds.ok; kai_pf_lds.job = -1; return r; }
    __device__ bool all_dead(const KaiCtx& c) { wait(); for (int k = 0; k < c.C; k++) if (sh->top_key[k]) return false; return true; }
    __device__ void hot(const KaiCtx& c, QNode*& qn, int32_t*& qheap, int32_t*& root_heap) {
        if constexpr (VICTIM) { qn = c.qn; qheap = c.qheap; root_heap = c.root_heap; return; }  // the LDS region goes to the simulation queue (sim_tree)
        if (sh->tree_in_lds) call(CMD_LOADTREE);  // service waves copy the k_leaf_init records HBM → LDS
        qn = sh->qn; qheap = sh->qheap; root_heap = sh->root_heap;
    }
    // victim search: the job-order instance of the simulations (by far the hottest of the three) lives in the LDS tree region when it fits
    __device__ bool sim_tree(QNode*& qn, int32_t*& qheap, int32_t*& root_heap) { if (!sh->tree_in_lds) return false; qn = sh->qn; qheap = sh->qheap; root_heap = sh->root_heap; return true; }
    __device__ int64_t clock() { return (int64_t)clock64(); }
    // several engines of one victim action (MultiCtx, kai_engine_solver.inc solve_partial_multi): one workgroup each, on its own replica; MultiCtx is the only memory
    // they share.  Agent-scope atomics: the workgroups sit on different XCDs, whose L2s are only coherent through such accesses and the fences of the barrier.
    __device__ static void mw_store32(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ static int32_t mw_load32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ static void mw_store64(int64_t* p, int64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ static int64_t mw_load64(const int64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ static int32_t mw_fetch_add32(int32_t* p, int32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ static void mw_fetch_min32(int32_t* p, int32_t v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ static double coh_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ static int32_t coh_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ static void grid_sync(MultiCtx* m, int clear) {  // the control lanes of the action's workgroups (all resident: one per compute unit at most, kai_core.hip)
        const int gen = __hip_atomic_load(&m->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_thread_fence(__ATOMIC_RELEASE);
        if (__hip_atomic_fetch_add(&m->bar_count, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1 == m->world) {
            __hip_atomic_store(&m->next[clear], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&m->hit[clear], 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&m->bar_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&m->bar_gen, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long long spins = 0;
            while (__hip_atomic_load(&m->bar_gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1ll << 27)) { __hip_atomic_store(&m->fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }  // ≈ a minute: an engine left the protocol; every engine gives up
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    // the wave's outcomes over the GPUs of a node-sharded group (kai_victim_shard.hpp): the host carries the exchange.  Everything the engines wrote into MultiCtx went
    // there with agent-scope atomics before the barrier this lane just left; ring the mailbox (pinned host memory, system scope) and wait for the answer — the merged
    // wave is in MultiCtx by then (written by copies the host waited for), read by every engine with agent-scope loads after the next barrier.
    __device__ static void mw_exchange(const KaiCtx& c, MultiCtx* m, int b) {
        XMail* mail = c.mw_mail;
        if (!mail) { __hip_atomic_store(&m->fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        const int seq = __hip_atomic_load(&mail->req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;
        __hip_atomic_store(&mail->buf, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        __hip_atomic_store(&mail->req, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        long long spins = 0;
        while (__hip_atomic_load(&mail->resp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(127);
            if (++spins > (1ll << 24)) { __hip_atomic_store(&m->fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }  // about a minute: the host (or another rank) is gone; every engine gives up
        }
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
    }
    __device__ void finish() {
        wait();
        // blocks still on the dirty list: bring the HBM level of the index up to date, the next action of the session starts from it
        if (sh->n_dirty) { sh->cmd = CMD_REFRESH; __syncthreads(); sh->in_flight = 1; wait(); }
        if (!VICTIM && g_ctx.sg) { g_ctx.sg->cmd = CMD_EXIT; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __hip_atomic_store(&g_ctx.sg->seq, ++g_el.sg_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        sh->cmd = CMD_EXIT; __syncthreads();
    }
};

using DevBackend = DevBackendT<false, false>;

// L2 entry (class k, super-block sb) from the 64 L1 entries below it, then the class top from the L2 row
__device__ __forceinline__ void svc_l2(const KaiCtx& c, ActShared* sh, int k, int sb, int lane) {
    int e = sb * 64 + lane;
    uint64_t key = e < c.NB ? c.sum1_key[(size_t)k * c.NB + e] : 0; int n = e < c.NB ? c.sum1_node[(size_t)k * c.NB + e] : 0x7fffffff;
    wave_argmax(key, n);
    if (lane == 0) { sh->s2_key[k * c.NSB + sb] = key; sh->s2_node[k * c.NSB + sb] = n; }
}
__device__ __forceinline__ void svc_top(const KaiCtx& c, ActShared* sh, int k, int lane) {
    uint64_t key = lane < c.NSB ? sh->s2_key[k * c.NSB + lane] : 0; int n = lane < c.NSB ? sh->s2_node[k * c.NSB + lane] : 0x7fffffff;
    wave_argmax(key, n);
    if (lane == 0) { sh->top_key[k] = key; sh->top_node[k] = n; }
}

// The passes over the NODES a command asks for (pre-order range, best node, the node loops of subSetNodesFn), over the slice [n_lo, n_hi) with `lanes` lanes of this
// workgroup (`slot` = this lane's number among them); per-wave partial results go to the workgroup's ActShared.  The control lane's own workgroup runs it on its scan
// waves; with a scan grid (ScanGrid below) the helper workgroups run it on theirs at the same time.
__device__ __forceinline__ void scan_cmd(const KaiCtx& c, ActShared* sh, int cmd, int n_lo, int n_hi, int slot, int lanes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (cmd == CMD_MINMAX) {  // getMinMaxPerNode (plugins/nodeplacement/pack.go:66-86)
        int r = sh->r;
        double lo = 1.7976931348623157e308, hi = 0;
        KAI_GP(const uint32_t) ns_bits = sh->nodeset;
        for (int n = n_lo + slot; n < n_hi; n += lanes) {
            if (ns_bits && !((ns_bits[n >> 5] >> (n & 31)) & 1)) continue;  // the pre-order scan ranges the node set (pack.go:66-86)
            if (c.n_alloc[(size_t)r * c.N + n] == 0) continue;
            double cur = c.n_idle[(size_t)r * c.N + n] + c.n_rel[(size_t)r * c.N + n];
            if (cur < lo) lo = cur;
            if (cur > hi) hi = cur;
        }
        for (int o = 32; o > 0; o >>= 1) { double a = __shfl_xor(lo, o, 64), b = __shfl_xor(hi, o, 64); if (a < lo) lo = a; if (b > hi) hi = b; }
        if (lane == 0) { sh->part_min[wave] = lo; sh->part_max[wave] = hi; }
    } else if (cmd == CMD_TOPO) {  // the node loops of subSetNodesFn (kai_engine.hpp subset_nodes)
        const TopoScan t = sh->topo;
        if (t.op == 1) {
            int mn[KAI_TOPO_SCAN_LEVELS], mx[KAI_TOPO_SCAN_LEVELS]; int any = 0;
            for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { mn[l] = 0x7fffffff; mx[l] = -0x7fffffff - 1; }
            for (int n = n_lo + slot; n < n_hi; n += lanes) {
                if (t.parent && !((t.parent[n >> 5] >> (n & 31)) & 1)) continue;
                if (c.node_domain[(size_t)t.row0 * c.N + n] < 0) continue;
                any = 1;
                for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) if (l < t.L) { const int dd = c.node_domain[(size_t)(t.row0 + l) * c.N + n]; if (dd < mn[l]) mn[l] = dd; if (dd > mx[l]) mx[l] = dd; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                any |= __shfl_xor(any, o, 64);
                for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { const int a = __shfl_xor(mn[l], o, 64), b = __shfl_xor(mx[l], o, 64); if (a < mn[l]) mn[l] = a; if (b > mx[l]) mx[l] = b; }
            }
            if (lane == 0) { for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { sh->topo_min[wave][l] = mn[l]; sh->topo_max[wave][l] = mx[l]; } sh->topo_any[wave] = any; }
        } else if (t.op == 15) {  // the survey: ops 1, 2 and 3 in one pass, two 64-node steps of a wave in flight at a time (their loads go out together)
            int mn[KAI_TOPO_SCAN_LEVELS], mx[KAI_TOPO_SCAN_LEVELS]; int any = 0;
            for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { mn[l] = 0x7fffffff; mx[l] = -0x7fffffff - 1; }
            const int DT = c.D + c.T;
            constexpr int U = 2;
            for (int nb = n_lo + slot - lane; nb < n_hi; nb += lanes * U) {
                int n[U], top[U], leaf[U], dlv[U][KAI_TOPO_SCAN_LEVELS]; uint32_t pw[U]; double av[U][KAI_MAX_RES];
                for (int u = 0; u < U; u++) {  // every load unconditional, on a clamped index
                    const int nn = nb + u * lanes + lane; n[u] = nn;
                    const int x = nn < n_hi ? nn : n_hi - 1;
                    top[u] = c.node_domain[(size_t)t.row0 * c.N + x];
                    leaf[u] = c.node_domain[(size_t)(t.row0 + t.L - 1) * c.N + x];
                    pw[u] = t.parent ? t.parent[x >> 5] : 0xffffffffu;
                    for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) dlv[u][l] = l < t.L ? c.node_domain[(size_t)(t.row0 + l) * c.N + x] : 0;
                    for (int r = 0; r < KAI_MAX_RES; r++) av[u][r] = r < t.R ? c.n_idle[(size_t)r * c.N + x] + c.n_rel[(size_t)r * c.N + x] : 0.0;
                }
                for (int u = 0; u < U; u++) {
                    const bool act = n[u] < n_hi && top[u] >= 0 && leaf[u] >= 0;  // a node of the topology
                    if (act && ((pw[u] >> (n[u] & 31)) & 1u)) {
                        any = 1;
                        for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) if (l < t.L) { if (dlv[u][l] < mn[l]) mn[l] = dlv[u][l]; if (dlv[u][l] > mx[l]) mx[l] = dlv[u][l]; }
                    }
                    const unsigned long long am = __ballot(act);
                    if (!am) continue;
                    const int l0 = __shfl(leaf[u], __ffsll((long long)am) - 1, 64);
                    const bool uni = __ballot(act && leaf[u] != l0) == 0;
                    for (int r = 0; r < KAI_MAX_RES; r++) if (r < t.R) {
                        double v = act ? av[u][r] : 0.0;
                        if (uni) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); if (lane == 0 && v != 0.0) atomicAdd((double*)&c.dom_free[(size_t)l0 * KAI_MAX_RES + r], v); }
                        else if (act) atomicAdd((double*)&c.dom_free[(size_t)leaf[u] * KAI_MAX_RES + r], v);
                    }
                    if (t.what & 2) {
                        int count = act ? topo_node_count(t, av[u]) : 0;
                        if (uni) { for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o, 64); if (lane == 0 && count) atomicAdd((int*)&c.dom_tmp[2 * DT + l0], count); }
                        else if (count) atomicAdd((int*)&c.dom_tmp[2 * DT + leaf[u]], count);
                    }
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                any |= __shfl_xor(any, o, 64);
                for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { const int a = __shfl_xor(mn[l], o, 64), b = __shfl_xor(mx[l], o, 64); if (a < mn[l]) mn[l] = a; if (b > mx[l]) mx[l] = b; }
            }
            if (lane == 0) { for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { sh->topo_min[wave][l] = mn[l]; sh->topo_max[wave][l] = mx[l]; } sh->topo_any[wave] = any; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the sums are in memory (agent-scope atomics) before op 16 reads them coherently; no acquire: nothing here is read back through a plain load
        } else if (t.op >= 5) {  // loops over the domains of a topology (Engine::topo_dom_body)
            NullBackend nb; Engine<NullBackend> eng(c, nb);
            int cnt = 0;
            for (int d = slot; d < c.D + c.T; d += lanes) cnt += eng.topo_dom_body(t, d);
            if (t.op == 9) { for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64); if (lane == 0) sh->topo_any[wave] = cnt; }
            // a loop over domains stays inside this workgroup: its atomics (NullBackend::add_*) and its fence have workgroup scope, so this XCD's L2 is neither written back nor emptied —
            // except after ops 5 and 7, whose stores must be in memory before the agent-scope atomics of the survey (ops 2 / 3) add to them
            if (t.op == 5 || t.op == 7) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        } else if (t.op == 4) {  // build_node_set: one 32-node word of the bitmap per step
            for (int w = (n_lo >> 5) + slot; w < ((n_hi + 31) >> 5); w += lanes) {  // (slices begin on multiples of 64 nodes)
                uint32_t word = 0; const uint32_t pw = t.parent ? t.parent[w] : 0xffffffffu;
                for (int b = 0; b < 32; b++) {
                    const int n = w * 32 + b; if (n >= c.N) break;
                    bool in = (pw >> b) & 1u;
                    if (in && t.domain >= 0) in = t.dl < 0 ? c.node_domain[(size_t)t.row0 * c.N + n] >= 0 : c.node_domain[(size_t)(t.row0 + t.dl) * c.N + n] == t.domain;
                    if (in) word |= 1u << b;
                }
                t.out[w] = word;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // (a slice reads only the words it wrote itself)
        } else {
            // ops 2 / 3: a wave takes 64 consecutive nodes per step — normally the nodes of ONE leaf domain, whose contributions are then summed
            // across the wave and added to the domain once (the amounts are integral under exact_sums, counts are integers: the grouping of the
            // additions does not show); a step that straddles leaf domains falls back to one atomic per node
            for (int nb = n_lo + slot - lane; nb < n_hi; nb += lanes) {
                const int n = nb + lane;
                const bool act = n < n_hi && topo_node_in_domain(c, t, n);
                const int leaf = act ? c.node_domain[(size_t)(t.row0 + t.L - 1) * c.N + n] : -1;
                const unsigned long long am = __ballot(act);
                if (!am) continue;
                const int l0 = __shfl(leaf, __ffsll((long long)am) - 1, 64);
                const bool uni = __ballot(act && leaf != l0) == 0;
                if (t.op == 2) {
                    for (int r = 0; r < t.R; r++) {
                        double v = act ? c.n_idle[(size_t)r * c.N + n] + c.n_rel[(size_t)r * c.N + n] : 0.0;
                        if (uni) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); if (lane == 0 && v != 0.0) atomicAdd((double*)&c.dom_free[(size_t)l0 * KAI_MAX_RES + r], v); }
                        else if (act) atomicAdd((double*)&c.dom_free[(size_t)leaf * KAI_MAX_RES + r], v);
                    }
                } else {
                    int count = 0;
                    if (act) {
                        if (t.one_pod) count = t.tasks;
                        else {
                            double cur[KAI_MAX_RES]; for (int r = 0; r < KAI_MAX_RES; r++) cur[r] = t.mx[r];  // k-th test pod = k x the maximal pod, by repeated addition
                            for (;;) { if (!fits(c, cur, n, true)) break; count++; for (int r = 0; r < t.R; r++) cur[r] += t.mx[r]; }
                        }
                    }
                    if (uni) { for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o, 64); if (lane == 0 && count) atomicAdd((int*)&c.dom_alloc_pods[l0], count); }
                    else if (count) atomicAdd((int*)&c.dom_alloc_pods[leaf], count);
                }
            }
            __threadfence();
        }
    } else if (cmd == CMD_BEST) {  // OrderedNodesByTask + FittingNode collapsed to an arg-max (framework/session.go:201-264, 466-485)
        const ScanReq& q = sh->req;
        int best = -1; unsigned long long bk = 0;
        KAI_GP(const uint32_t) ns_bits = sh->nodeset; const int trow = sh->topo_row; KAI_GP(const double) tscore = sh->topo_score;
        for (int n = n_lo + slot; n < n_hi; n += lanes) {
            if (ns_bits && !((ns_bits[n >> 5] >> (n & 31)) & 1)) continue;
            double sc = 0;
            if (!scan_node_score(c, q, n, sc)) continue;
            if (trow >= 0) {  // topology.nodeOrderFn (plugins/topology/node_scoring.go:17-35): a node without a score is dropped (session.go:247-251)
                int dd = c.node_domain[(size_t)trow * c.N + n]; double ts = dd >= 0 ? tscore[dd] : -1.0;
                if (ts < 0) continue;
                sc += ts;
            }
            unsigned long long k = orderable(sc);
            if (best < 0 || k > bk) { best = n; bk = k; }                        // n ascends: the first of equal scores is the lowest name rank
        }
        for (int o = 32; o > 0; o >>= 1) {
            int on = __shfl_xor(best, o, 64); unsigned long long ok = __shfl_xor(bk, o, 64);
            if (on >= 0 && (best < 0 || ok > bk || (ok == bk && on < best))) { best = on; bk = ok; }
        }
        if (lane == 0) { sh->part_node[wave] = best; sh->part_key[wave] = bk; }
    }
}

// service-wave side: each of the 7 service waves owns the classes  k ≡ wave-1 (mod 7)  of the class index and the nodes
// (wave-1)*64 + lane (mod 448) of this workgroup's slice of a pass over the nodes
__device__ void service_loop(const KaiCtx& cref, ActShared* sh) {
    const KaiCtx c = cref;  // loop-invariant: a register copy of the fields used below instead of an LDS read + wait in front of every access
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, slot = threadIdx.x - 64, hw = wave - 1;
    for (;;) {
        __syncthreads();  // wait for a command
        int cmd = sh->cmd;
        if (cmd == CMD_EXIT) return;
        long long ts = clock64();
        if (cmd == CMD_LOADTREE) {
            const int4* src = reinterpret_cast<const int4*>(c.qn); int4* dst = reinterpret_cast<int4*>(sh->qn);
            int n16 = (int)(((size_t)c.Q * sizeof(QNode) + 15) / 16);
            for (int i = slot; i < n16; i += SCAN_LANES) dst[i] = src[i];
        } else if (cmd == CMD_BEGIN) {
            for (int k = hw; k < c.C; k += SVC) {
                for (int sb = 0; sb < c.NSB; sb++) svc_l2(c, sh, k, sb, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                svc_top(c, sh, k, lane);
            }
        } else if (cmd == CMD_REFRESH) {
            const int nd = sh->n_dirty;
            if (nd == 1) {  // the common case (one placement): everything in one pass, upper levels patched in registers
                const int b = sh->dirty[0], n = b * KAI_BLOCK + lane, sb = b / 64, e = sb * 64 + lane;
                NodeRegs ns; load_node(c, n < c.N ? n : 0, ns);
                for (int k = hw; k < c.C; k += SVC) {
                    uint64_t rk = e < c.NB ? c.sum1_key[(size_t)k * c.NB + e] : 0; int rn = e < c.NB ? c.sum1_node[(size_t)k * c.NB + e] : 0x7fffffff;  // L1 row, in flight with the node loads
                    uint64_t tk = lane < c.NSB ? sh->s2_key[k * c.NSB + lane] : 0; int tn = lane < c.NSB ? sh->s2_node[k * c.NSB + lane] : 0x7fffffff;
                    uint64_t key = n < c.N ? class_key_regs(c, c.cls[k], ns) : 0; int bn = n;
                    wave_argmax(key, bn);
                    if (lane == 0) { c.sum1_key[(size_t)k * c.NB + b] = key; c.sum1_node[(size_t)k * c.NB + b] = bn; }
                    if (lane == (b & 63)) { rk = key; rn = bn; }
                    wave_argmax(rk, rn);
                    if (lane == 0) { sh->s2_key[k * c.NSB + sb] = rk; sh->s2_node[k * c.NSB + sb] = rn; }
                    if (lane == sb) { tk = rk; tn = rn; }
                    wave_argmax(tk, tn);
                    if (lane == 0) { sh->top_key[k] = tk; sh->top_node[k] = tn; }
                }
            } else {
                // L1: re-evaluate the dirty blocks for this wave's classes
                for (int i = 0; i < nd; i++) {
                    int b = sh->dirty[i], n = b * KAI_BLOCK + lane;
                    NodeRegs ns; load_node(c, n < c.N ? n : 0, ns);
                    for (int k = hw; k < c.C; k += SVC) {
                        uint64_t key = n < c.N ? class_key_regs(c, c.cls[k], ns) : 0; int bn = n;
                        wave_argmax(key, bn);
                        if (lane == 0) { c.sum1_key[(size_t)k * c.NB + b] = key; c.sum1_node[(size_t)k * c.NB + b] = bn; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // this wave re-reads its own L1 rows
                for (int k = hw; k < c.C; k += SVC) {
                    for (int i = 0; i < nd; i++) {
                        int sb = sh->dirty[i] / 64; bool seen = false;
                        for (int x = 0; x < i; x++) if (sh->dirty[x] / 64 == sb) seen = true;
                        if (!seen) svc_l2(c, sh, k, sb, lane);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    svc_top(c, sh, k, lane);
                }
            }
        } else if (cmd == CMD_STAGE) {  // gather one job for the staged path (JobPf, stage_job_lane): one wave, a lane per pod of the chunk
            if (hw == 0) {
                int bad = stage_job_lane(c, sh->stage_job, kai_frame_lds, kai_pf_lds, lane, 64);
                bad = __any(bad);
                if (lane == 0) kai_pf_lds.ok = kai_pf_lds.shape && !bad;
            }
        } else if (cmd == CMD_PFOR) {  // an index loop of the victim search (Engine::pfor_body)
            const PforReq r = sh->pfor;
            NullBackend nb; Engine<NullBackend> eng(c, nb);
            for (int i = slot; i < r.n; i += SCAN_LANES) eng.pfor_body(r, i);
            __threadfence();
        } else if (cmd == CMD_MINMAX || cmd == CMD_TOPO || cmd == CMD_BEST) {
            scan_cmd(c, sh, cmd, 0, sh->slice_hi, slot, SCAN_LANES);
        } else if (cmd == CMD_EVAL) {
            if (slot < sh->eval_n) { double sc = 0; const bool ok = scan_node_score(c, sh->req, sh->eval_nodes[slot], sc); sh->eval_ok[slot] = ok ? 1 : 0; sh->eval_sc[slot] = sc; }
        }
        if (cmd == CMD_REFRESH && threadIdx.x == 64) sh->t_svc += clock64() - ts;
        __syncthreads();  // results ready
    }
}

// A helper workgroup of the scan grid: all eight waves scan; wave 0 watches the table, and the helper that finishes a command last folds everybody's partial results.
__device__ void helper_loop(const KaiCtx& cref, ActShared* sh) {
    const KaiCtx c = cref;
    ScanGrid* g = c.sg;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // sign on — unless this workgroup sits on the engine's XCD: its acquire fences would empty the L2 the control lane lives in
    if (threadIdx.x == 0) {
        int x0; long long spins = 0;
        while ((x0 = sg_ld(&g->xcc0)) == 0) { __builtin_amdgcn_s_sleep(8); if (++spins > (1ll << 28)) break; }
        int h = -1;
        if (x0 != 0 && xcc_id() + 1 != x0) h = __hip_atomic_fetch_add(&g->n_reg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        __hip_atomic_fetch_add(&g->checked, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        sh->r = h;
    }
    __syncthreads();
    const int h = sh->r;
    __syncthreads();
    if (h < 0) return;
    int gen = 0, n_lo = 0, n_hi = 0, H = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            long long spins = 0; bool lost = false;
            while (sg_ld(&g->seq) == gen) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1ll << 30)) { lost = true; break; }  // minutes without a command: the engine is gone
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (lost) { __hip_atomic_store(&g->fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sh->cmd = CMD_EXIT; }
            else {
                sh->cmd = g->cmd; sh->r = g->r; sh->nodeset = g->nodeset; sh->topo_row = g->topo_row; sh->topo_score = g->topo_score;
                if (sh->cmd == CMD_BEST) sh->req = g->req; else if (sh->cmd == CMD_TOPO) sh->topo = g->topo;
                sh->slice_hi = g->per; sh->n_dirty = g->n_reg;
            }
        }
        gen++;
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int cmd = sh->cmd;
        if (cmd == CMD_EXIT) return;
        if (gen == 1) { const int per = sh->slice_hi; H = sh->n_dirty; n_lo = h * per < c.N ? h * per : c.N; n_hi = (h + 1) * per < c.N ? (h + 1) * per : c.N; }
        scan_cmd(c, sh, cmd, n_lo, n_hi, threadIdx.x, WG);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (wave != 0) continue;
        const int top = cmd == CMD_TOPO ? sh->topo.op : 0;
        int last = 0;
        if (lane == 0) {
            ScanGridPart& p = g->part[h];
            if (cmd == CMD_MINMAX) { double lo = 1.7976931348623157e308, hi = 0; for (int w = 0; w < WAVES; w++) { if (sh->part_min[w] < lo) lo = sh->part_min[w]; if (sh->part_max[w] > hi) hi = sh->part_max[w]; } p.mn = lo; p.mx = hi; }
            else if (cmd == CMD_BEST) {
                int best = -1; unsigned long long bk = 0;
                for (int w = 0; w < WAVES; w++) { int n = sh->part_node[w]; if (n < 0) continue; unsigned long long k = sh->part_key[w]; if (best < 0 || k > bk || (k == bk && n < best)) { best = n; bk = k; } }
                p.node = best; p.key = bk;
            } else if (top == 1 || top == 15) {
                int any = 0;
                for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { int mn = 0x7fffffff, mx = -0x7fffffff - 1; for (int w = 0; w < WAVES; w++) { if (sh->topo_min[w][l] < mn) mn = sh->topo_min[w][l]; if (sh->topo_max[w][l] > mx) mx = sh->topo_max[w][l]; } p.tmin[l] = mn; p.tmax[l] = mx; }
                for (int w = 0; w < WAVES; w++) any |= sh->topo_any[w];
                p.any = any;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            last = __hip_atomic_fetch_add(&g->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == gen * H;
        }
        last = __shfl(last, 0, 64);
        if (!last) continue;
        // every helper's partial result is on the table: fold them, a lane per helper
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ScanGridPart& res = g->res;
        if (cmd == CMD_MINMAX) {
            double lo = 1.7976931348623157e308, hi = 0;
            for (int x = 1 + lane; x <= H; x += 64) { const ScanGridPart& p = g->part[x]; if (p.mn < lo) lo = p.mn; if (p.mx > hi) hi = p.mx; }
            for (int o = 32; o > 0; o >>= 1) { double a = __shfl_xor(lo, o, 64), b = __shfl_xor(hi, o, 64); if (a < lo) lo = a; if (b > hi) hi = b; }
            if (lane == 0) { res.mn = lo; res.mx = hi; }
        } else if (cmd == CMD_BEST) {
            int best = -1; unsigned long long bk = 0;
            for (int x = 1 + lane; x <= H; x += 64) { const ScanGridPart& p = g->part[x]; const int n = p.node; const unsigned long long k = p.key; if (n >= 0 && (best < 0 || k > bk || (k == bk && n < best))) { best = n; bk = k; } }
            for (int o = 32; o > 0; o >>= 1) {
                int on = __shfl_xor(best, o, 64); unsigned long long ok = __shfl_xor(bk, o, 64);
                if (on >= 0 && (best < 0 || ok > bk || (ok == bk && on < best))) { best = on; bk = ok; }
            }
            if (lane == 0) { res.node = best; res.key = bk; }
        } else if (top == 1 || top == 15) {
            int any = 0, mn[KAI_TOPO_SCAN_LEVELS], mx[KAI_TOPO_SCAN_LEVELS];
            for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { mn[l] = 0x7fffffff; mx[l] = -0x7fffffff - 1; }
            for (int x = 1 + lane; x <= H; x += 64) { const ScanGridPart& p = g->part[x]; any |= p.any; for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { if (p.tmin[l] < mn[l]) mn[l] = p.tmin[l]; if (p.tmax[l] > mx[l]) mx[l] = p.tmax[l]; } }
            for (int o = 32; o > 0; o >>= 1) {
                any |= __shfl_xor(any, o, 64);
                for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { const int a = __shfl_xor(mn[l], o, 64), b = __shfl_xor(mx[l], o, 64); if (a < mn[l]) mn[l] = a; if (b > mx[l]) mx[l] = b; }
            }
            if (lane == 0) { res.any = any; for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { res.tmin[l] = mn[l]; res.tmax[l] = mx[l]; } }
        }
        if (lane == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __hip_atomic_store(&g->ready, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

// The engine's workgroup: wave 0 lane 0 = control, waves 1..7 = service.  scan_wgs > 1: workgroups 1.. are the helpers of the scan grid (same context).
template <bool VICTIM, bool TREE_LDS>
__global__ void __launch_bounds__(WG) k_action(const KaiCtx* __restrict__ cp, int action, int tree_in_lds, int scan_wgs) {
    {   // the context into LDS: every pointer fetch of the engine is a ds_read the compiler can batch
        const int* src = reinterpret_cast<const int*>(cp + (scan_wgs > 1 ? 0 : blockIdx.x)); int* dst = reinterpret_cast<int*>(&g_ctx);  // (a victim action on several workgroups: one context — one replica of the session state — each)
        for (int i = threadIdx.x; i < (int)(sizeof(KaiCtx) / 4); i += WG) dst[i] = src[i];
    }
    if (scan_wgs > 1 && blockIdx.x > 0) {
        __syncthreads();
        if (threadIdx.x == 0) { g_sh.cmd = CMD_NONE; g_sh.nodeset = nullptr; g_sh.topo_row = -1; g_sh.topo_score = nullptr; }
        __syncthreads();
        helper_loop(g_ctx, &g_sh);
        return;
    }
    __syncthreads();
    const KaiCtx& c = g_ctx;
    ActShared& sh = g_sh;
    if (threadIdx.x == 0) {
        kai_pf_lds.job = -1; kai_pf_lds.ok = 0;
        sh.cmd = CMD_NONE; sh.n_dirty = 0; sh.in_flight = 0; sh.slice_hi = c.N; g_el.sg_gen = 0; sh.tree_in_lds = tree_in_lds; sh.nodeset = nullptr; sh.topo_row = -1; sh.topo_score = nullptr; sh.t_publish = 0; sh.t_wait = 0; sh.t_svc = 0; for (int i = 0; i < 6; i++) sh.t_seg[i] = 0;
        size_t off = 0;
        sh.s2_key = (unsigned long long __attribute__((address_space(3)))*)(kai_dyn_lds); sh.s2_node = (int32_t __attribute__((address_space(3)))*)(kai_dyn_lds + (size_t)c.C * c.NSB * 8);
        off = lds_index_bytes(c.C, c.NSB);
        if (tree_in_lds) { sh.qn = reinterpret_cast<QNode*>(kai_dyn_lds + off); sh.qheap = reinterpret_cast<int32_t*>(kai_dyn_lds + off + (size_t)c.Q * sizeof(QNode)); sh.root_heap = sh.qheap + (c.Q + 1); }
        else { sh.qn = c.qn; sh.qheap = c.qheap; sh.root_heap = c.root_heap; }
    }
    __syncthreads();
    if (threadIdx.x >= 64) { service_loop(c, &g_sh); return; }
    if (threadIdx.x != 0) return;  // the rest of wave 0 idles: s_barrier counts wavefronts, not lanes
    if constexpr (!VICTIM) if (scan_wgs > 1) {  // the scan grid signs on: the helpers that do not share this XCD (their fences would empty its L2)
        ScanGrid* g = g_ctx.sg;
        __hip_atomic_store(&g->xcc0, xcc_id() + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long spins = 0;
        while (sg_ld(&g->checked) != scan_wgs - 1) { if (++spins > (1ll << 28)) break; }
        const int H = sg_ld(&g->checked) == scan_wgs - 1 ? sg_ld(&g->n_reg) : 0;
        g_ctx.sg_wgs = H + 1; g_ctx.sg_per = (((c.N + H) / (H + 1) + 63) / 64) * 64;
        g->per = g_ctx.sg_per;
        if (H == 0) g_ctx.sg_wgs = 1;  // (helpers that never answered would wait for a command forever: they get their EXIT from finish() below only when the grid is on — so keep it on if any signed on)
    }
    DevBackendT<VICTIM, TREE_LDS> be;
    Engine<DevBackendT<VICTIM, TREE_LDS>> eng(c, be);
    if constexpr (VICTIM) eng.execute_victim_action(); else if (action == KAI_ACTION_ALLOCATE) eng.execute_allocate();
    c.st->prof[1] = sh.t_publish; c.st->prof[6] = sh.t_wait; c.st->prof[PF_PUSH] = sh.t_svc;
#ifdef KAI_PROF_VICTIM
    if constexpr (!VICTIM) { c.st->prof[36] = sh.t_seg[0]; c.st->prof[37] = sh.t_seg[1]; c.st->prof[38] = sh.t_seg[2]; c.st->prof[39] = sh.t_seg[3]; }  // scan grid: publish (release fence), own slice, wait for the fold, commands
#endif
    be.finish();
}

// Once no class has a fitting node at a committed state, every job still queued is attempted, gated and fails at its first
// task without changing any state (Engine::drain_job): resolve them chip-wide.  Slot i of a leaf's region holds a job of the
// sorted part if cur <= pos < end and a job of the side heap if pos < side_len.
__global__ void k_drain(KaiCtx c, const int32_t* slot_queue) {
    if (!c.st->drain_pending) return;
    NullBackend nb; Engine<NullBackend> eng(c, nb);
    int64_t att = 0, dec = 0, rb = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.J; i += gridDim.x * blockDim.x) {
        int q = slot_queue[i]; if (q < 0) continue;
        int pos = i - c.q_job_off[q];
        if (pos >= c.lq_cur[q] && pos < c.lq_end[q]) eng.drain_job(c.lq_sorted[i], att, dec, rb);
        if (pos < c.lq_side_len[q]) eng.drain_job(c.lq_side[i], att, dec, rb);
    }
    for (int o = 32; o > 0; o >>= 1) { att += __shfl_xor((long long)att, o, 64); dec += __shfl_xor((long long)dec, o, 64); rb += __shfl_xor((long long)rb, o, 64); }
    if ((threadIdx.x & 63) == 0 && att) {
        atomicAdd((unsigned long long*)&c.st->jobs_attempted, (unsigned long long)att); atomicAdd((unsigned long long*)&c.st->decisions, (unsigned long long)dec);
        atomicAdd((unsigned long long*)&c.st->rollbacks, (unsigned long long)rb);
        atomicAdd((unsigned long long*)&c.st->drained_jobs, (unsigned long long)att); atomicAdd((unsigned long long*)&c.st->drained_decisions, (unsigned long long)dec);
    }
}

// kai_best_node: one OrderedNodesByTask + FittingNode against the current session state (brute-force scan)
__global__ void __launch_bounds__(WG) k_best_node(KaiCtx cv, int pod, int pipeline_only, int32_t* out, const uint32_t* nodeset) {
    if (threadIdx.x == 0) { g_ctx = cv; g_ctx.use_index = 0; }
    __syncthreads();
    const KaiCtx& c = g_ctx;
    ActShared& sh = g_sh;
    if (threadIdx.x == 0) { sh.cmd = CMD_NONE; sh.n_dirty = 0; sh.in_flight = 0; sh.slice_hi = c.N; g_ctx.sg_wgs = 0; sh.tree_in_lds = 0; sh.nodeset = nullptr; sh.topo_row = -1; sh.topo_score = nullptr; sh.s2_key = nullptr; sh.s2_node = nullptr; sh.qn = c.qn; sh.qheap = c.qheap; sh.root_heap = c.root_heap; }
    __syncthreads();
    if (threadIdx.x >= 64) { service_loop(c, &g_sh); return; }
    if (threadIdx.x != 0) return;
    DevBackend be;
    Engine<DevBackend> eng(c, be);
    g_el.scope_bits = (KAI_GP(const uint32_t))nodeset;  // the caller's node set (SubsetNodesFn result)
    int n = -1, pipe = 0;
    if (!((c.plugins & KAI_PLUGIN_PREDICATES) && eng.task_over_capacity(pod))) {
        bool allocatable = false;
        n = eng.find_node(pod, allocatable);
        if (n >= 0) pipe = (pipeline_only || !allocatable) ? 1 : 0;
    }
    out[0] = n; out[1] = pipe;
    be.finish();
}

}  // namespace kai
