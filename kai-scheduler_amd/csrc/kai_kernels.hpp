// kai_kernels.hpp — gfx950 kernels of the scheduling-cycle core (included by kai_core.hip only).
//
//  * session-open kernels: node accounting from the pods, proportion totals, queue usage roll-up
//    (segmented reductions pods → job → leaf queue → ancestors), fair-share division per tree level;
//  * the persistent action kernel: wave 0 / lane 0 drives kai::Engine, the other wavefronts of the
//    workgroup serve its node scans out of the HBM-resident node SoA (resource-major, so a wavefront reads
//    64 consecutive nodes of one resource = one 512-byte coalesced request).
//
// wave = 64 lanes; workgroup = 1024 threads = 16 wavefronts (1 control + 15 scan waves).
#pragma once
#include <hip/hip_runtime.h>

#include "kai_engine.hpp"

namespace kai {

constexpr int WG = 1024;        // threads per workgroup of the action kernel
constexpr int WAVES = WG / 64;  // 16
constexpr int SCAN_LANES = WG - 64;

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
        if (!ignore_gpus) acc[KAI_Q_GPU] += c.n_alloc[(size_t)KAI_RES_GPU * c.N + n];
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
    atomicAdd(&c.st->total[KAI_Q_GPU], -c.p_req[(size_t)KAI_RES_GPU * c.P + p]);
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
        if (st_allocated(s)) {
            for (int k = 0; k < 3; k++) ja[k] += q[k];
            if (c.p_accepted[p]) for (int k = 0; k < 3; k++) { al[k] += q[k]; rq[k] += q[k]; }  // AcceptedResource is empty for a pod no node holds
        } else if (s == KAI_POD_PENDING) {
            for (int k = 0; k < 3; k++) rq[k] += q[k];
        }
    }
    c.j_n_pending[j] = pending; c.j_tta_valid[j] = 0; c.j_tta_n[j] = 0;
    bool np = !c.j_preempt[j];
    for (int k = 0; k < 3; k++) {
        c.j_allocated[(size_t)k * c.J + j] = ja[k];
        jsum[(size_t)(0 + k) * c.J + j] = al[k]; jsum[(size_t)(3 + k) * c.J + j] = np ? al[k] : 0.0; jsum[(size_t)(6 + k) * c.J + j] = rq[k];
    }
}
// one wavefront per leaf queue: segmented reduction over the queue's jobs (jobs_by_queue is CSR by q_job_off)
__global__ void k_leaf_usage(KaiCtx c, const double* jsum, const int32_t* jobs_by_queue) {
    int q = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (q >= c.Q) return;
    int lane = threadIdx.x & 63, b = c.q_job_off[q], e = c.q_job_off[q + 1];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = b + lane; i < e; i += 64) { int j = jobs_by_queue[i]; for (int x = 0; x < 9; x++) acc[x] += jsum[(size_t)x * c.J + j]; }
    for (int x = 0; x < 9; x++) acc[x] = wave_sum(acc[x]);
    if (lane == 0) for (int k = 0; k < 3; k++) {
        QShare& s = c.q_share[(size_t)q * 3 + k];
        s.allocated = acc[k]; s.allocated_np = acc[3 + k]; s.request = acc[6 + k]; s.fair = 0;
    }
}
// ancestors: queues in decreasing depth push their sums to the parent (depth_order from the host; Q is small)
__global__ void k_tree_usage(KaiCtx c, const int32_t* depth_order) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int i = 0; i < c.Q; i++) {
        int q = depth_order[i], par = c.q_parent[q]; if (par < 0) continue;
        for (int k = 0; k < 3; k++) {
            QShare& s = c.q_share[(size_t)q * 3 + k]; QShare& d = c.q_share[(size_t)par * 3 + k];
            d.allocated += s.allocated; d.allocated_np += s.allocated_np; d.request += s.request;
        }
    }
}
// proportion.setFairShareForQueues (plugins/proportion/proportion.go:410-423): level by level, one lane per
// (sibling set, resource); a level's totals are the parents' fair shares of the previous level.
// child_off / children carry a virtual root at index Q whose children are the top queues.
__global__ void k_fair_share(KaiCtx c, const int32_t* lvl_off, const int32_t* lvl_parents, int n_levels,
                             double* weight, double* rem_amt, uint8_t* rem_has) {
    for (int l = 0; l < n_levels; l++) {
        int b = lvl_off[l], e = lvl_off[l + 1];
        for (int t = threadIdx.x; t < (e - b) * 3; t += blockDim.x) {
            int par = lvl_parents[b + t / 3], k = t % 3;
            double total = par == c.Q ? c.st->total[k] : c.q_share[(size_t)par * 3 + k].fair;
            const int32_t* kids = c.q_children + c.q_child_off[par];
            int nk = c.q_child_off[par + 1] - c.q_child_off[par];
            divide_sibling_set(c, kids, nk, k, total, c.k_value, weight + (size_t)k * c.Q, rem_amt + (size_t)k * c.Q, rem_has + (size_t)k * c.Q);
        }
        __threadfence(); __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// the cooperative node scan of the action kernel
// ------------------------------------------------------------------------------------------------------
enum ScanCmd : int32_t { CMD_NONE = 0, CMD_MINMAX = 1, CMD_BEST = 2, CMD_EXIT = 3 };

struct ScanShared {
    ScanReq req;
    int32_t cmd, r, pad0, pad1;
    double part_min[WAVES], part_max[WAVES];
    unsigned long long part_key[WAVES];  // orderable score bits
    uint32_t part_rank[WAVES];
    int32_t part_node[WAVES];
};

// monotone map f64 → u64 (larger double ⇒ larger key); scores here are finite and ≥ 0 but keep it general
__device__ __forceinline__ unsigned long long orderable(double d) {
    unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

struct DevScanner {
    ScanShared* sh;
    // control lane side -------------------------------------------------------------------------------
    __device__ void minmax(const KaiCtx&, int r, double& mn, double& mx) {
        sh->cmd = CMD_MINMAX; sh->r = r;
        __syncthreads();  // publish the command
        __syncthreads();  // partials ready
        double lo = 1.7976931348623157e308, hi = 0;  // math.MaxFloat64, 0 (plugins/nodeplacement/pack.go:66-68)
        for (int w = 1; w < WAVES; w++) { if (sh->part_min[w] < lo) lo = sh->part_min[w]; if (sh->part_max[w] > hi) hi = sh->part_max[w]; }
        mn = lo; mx = hi;
    }
    __device__ int best_node(const KaiCtx&, const ScanReq& q) {
        sh->req = q; sh->cmd = CMD_BEST;
        __syncthreads();
        __syncthreads();
        int best = -1; unsigned long long bk = 0; uint32_t br = 0;
        for (int w = 1; w < WAVES; w++) {
            int n = sh->part_node[w]; if (n < 0) continue;
            unsigned long long k = sh->part_key[w]; uint32_t rk = sh->part_rank[w];
            if (best < 0 || k > bk || (k == bk && rk < br)) { best = n; bk = k; br = rk; }
        }
        return best;
    }
    __device__ void finish() { sh->cmd = CMD_EXIT; __syncthreads(); }
};

// scan-wave side: each of the 15 scan waves owns the nodes  (wave-1)*64 + lane  (mod 960)
__device__ void scan_service(const KaiCtx& c, ScanShared* sh) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, slot = threadIdx.x - 64;
    for (;;) {
        __syncthreads();  // wait for a command
        int cmd = sh->cmd;
        if (cmd == CMD_EXIT) return;
        if (cmd == CMD_MINMAX) {  // getMinMaxPerNode (plugins/nodeplacement/pack.go:66-86)
            int r = sh->r;
            double lo = 1.7976931348623157e308, hi = 0;
            for (int n = slot; n < c.N; n += SCAN_LANES) {
                if (c.n_alloc[(size_t)r * c.N + n] == 0) continue;
                double cur = c.n_idle[(size_t)r * c.N + n] + c.n_rel[(size_t)r * c.N + n];
                if (cur < lo) lo = cur;
                if (cur > hi) hi = cur;
            }
            for (int o = 32; o > 0; o >>= 1) { double a = __shfl_xor(lo, o, 64), b = __shfl_xor(hi, o, 64); if (a < lo) lo = a; if (b > hi) hi = b; }
            if (lane == 0) { sh->part_min[wave] = lo; sh->part_max[wave] = hi; }
        } else {  // CMD_BEST: OrderedNodesByTask + FittingNode collapsed to an arg-max (framework/session.go:201-264, 466-485)
            const ScanReq& q = sh->req;
            int best = -1; unsigned long long bk = 0; uint32_t br = 0;
            for (int n = slot; n < c.N; n += SCAN_LANES) {
                if (!fits(c, q, n, true)) continue;              // IsTaskAllocatableOnReleasingOrIdle
                if (!node_predicates(c, q, n)) continue;         // ssn.PredicateFn
                bool fit_idle = q.best_effort || fits(c, q, n, false);
                unsigned long long k = orderable(node_score(c, q, n, fit_idle));
                uint32_t rk = c.n_name_rank[n];
                if (best < 0 || k > bk || (k == bk && rk < br)) { best = n; bk = k; br = rk; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                int on = __shfl_xor(best, o, 64); unsigned long long ok = __shfl_xor(bk, o, 64); uint32_t orr = __shfl_xor(br, o, 64);
                if (on >= 0 && (best < 0 || ok > bk || (ok == bk && orr < br))) { best = on; bk = ok; br = orr; }
            }
            if (lane == 0) { sh->part_node[wave] = best; sh->part_key[wave] = bk; sh->part_rank[wave] = br; }
        }
        __syncthreads();  // partials ready
    }
}

// One workgroup; wave 0 lane 0 = control, waves 1..15 = scan service.
__global__ void __launch_bounds__(WG) k_action(KaiCtx c, int action) {
    __shared__ ScanShared sh;
    if (threadIdx.x == 0) sh.cmd = CMD_NONE;
    __syncthreads();
    if (threadIdx.x >= 64) { scan_service(c, &sh); return; }
    if (threadIdx.x != 0) return;  // the rest of wave 0 idles: s_barrier counts wavefronts, not lanes
    DevScanner sc{&sh};
    Engine<DevScanner> eng(c, sc);
    if (action == KAI_ACTION_ALLOCATE) eng.execute_allocate();
    sc.finish();
}

// kai_best_node: one OrderedNodesByTask + FittingNode against the current session state
__global__ void __launch_bounds__(WG) k_best_node(KaiCtx c, int pod, int pipeline_only, int32_t* out) {
    __shared__ ScanShared sh;
    if (threadIdx.x == 0) sh.cmd = CMD_NONE;
    __syncthreads();
    if (threadIdx.x >= 64) { scan_service(c, &sh); return; }
    if (threadIdx.x != 0) return;
    DevScanner sc{&sh};
    Engine<DevScanner> eng(c, sc);
    ScanReq q; eng.fill_req(q, pod);
    int n = -1, pipe = 0;
    if (!((c.plugins & KAI_PLUGIN_PREDICATES) && eng.task_over_capacity(pod))) {
        if ((c.plugins & KAI_PLUGIN_NODEPLACEMENT) && q.strategy == KAI_BINPACK) sc.minmax(c, q.r_place, q.min_a, q.max_a);
        n = sc.best_node(c, q);
        if (n >= 0) { bool allocatable = q.best_effort || fits(c, q, n, false); pipe = (pipeline_only || !allocatable) ? 1 : 0; }
    }
    out[0] = n; out[1] = pipe;
    sc.finish();
}

}  // namespace kai
