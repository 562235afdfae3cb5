// kai_host_prep.hpp — index structures derived from a kai_snapshot_soa on the host.
//
// Pure re-orderings / groupings of the input (no scheduling arithmetic): nodes permuted into name-rank order, each job's
// pods in TaskOrderFn order, the queue tree as CSR with a virtual root, each queue's jobs in their static order, queues by
// depth, fair-share levels, the proportion plugin's per-queue quota records, and the scan classes (pods grouped by request
// vector + predicate class) with the guards that admit a class to the device's class index.
// Shared by kai_core.hip (uploads them to HBM) and by tests/host_sim (which debugs the engine's control flow without a GPU).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kai_engine.hpp"
#include "kai_parallel.hpp"

namespace kai {

// Shared-GPU requests (a fraction of one device: pod_gpu_portion; MiB of one device: pod_gpu_memory) as the per-pod quantities the engine reads.  The
// engine admits them only with ONE GPU memory size M for the whole cluster, so what a request takes on "its node" is known here.
struct SharedPods {
    bool any = false, mig = false, on = false; std::string err;  // any: shared-GPU requests; mig: some resource row is a MIG profile; on = any || mig: the arrays below are in use
    raw_vector<uint8_t> shared, kind; raw_vector<int64_t> mem, gmem; raw_vector<double> acc_gpu, pend_gpu, quota_gpu, mig_q;  // every element is written by build()'s parallel loop
    int32_t mig_g[KAI_MAX_RES] = {0}; int64_t mig_m[KAI_MAX_RES] = {0};
    enum { K_MIG = 1, K_LEGACY = 2, K_REGULAR = 4, K_GPUS = 8 };  // PodInfo: IsMigCandidate / IsLegacyMIGtask / IsRegularGPURequest / ResReq.GPUs() > 0
    static bool has(const kai_snapshot_soa* s) {
        if (!s->pod_gpu_portion && !s->pod_gpu_memory) return false;
        const size_t P = (size_t)std::max(s->n_pods, 0);
        std::vector<char> found((size_t)chunk_count(P), 0);
        parallel_chunks(P, [&](int ci, size_t p0, size_t p1) {
            for (size_t p = p0; p < p1; p++) if ((s->pod_gpu_portion && s->pod_gpu_portion[p] > 0) || (s->pod_gpu_memory && s->pod_gpu_memory[p] > 0)) { found[(size_t)ci] = 1; return; }
        });
        for (char f : found) if (f) return true;
        return false;
    }
    bool build(const kai_config& cfg, const kai_snapshot_soa* s) {  // false: refused, err says why
        const int P = s->n_pods, N = s->n_nodes;
        mig = false; on = false; err.clear(); for (int r = 0; r < KAI_MAX_RES; r++) { mig_g[r] = 0; mig_m[r] = 0; }  // (the object may be one a handle keeps between sessions)
        any = has(s);
        const size_t n = (size_t)std::max(P, 1);
        shared.resize(n); kind.resize(n); mem.resize(n); gmem.resize(n); acc_gpu.resize(n); pend_gpu.resize(n); quota_gpu.resize(n); mig_q.resize(n);
        if (P == 0) { shared[0] = 0; kind[0] = 0; mem[0] = 0; gmem[0] = 0; acc_gpu[0] = pend_gpu[0] = quota_gpu[0] = mig_q[0] = 0.0; }
        for (int r = KAI_RES_PODS + 1; r < s->n_res && r < KAI_MAX_RES; r++) { mig_g[r] = s->res_mig_gpus ? s->res_mig_gpus[r] : 0; mig_m[r] = s->res_mig_memory ? s->res_mig_memory[r] : 0; if (mig_g[r] > 0) mig = true; }
        on = any || mig;
        // one GPU memory size for the cluster — among the nodes that HAVE devices: a node without the gpu.memory label carries DefaultGpuMemory = 100 (node_info.go:673-687),
        // which on a CPU-only node is never read
        auto has_gpus = [&](int i) { if (s->node_allocatable[(size_t)KAI_RES_GPU * N + i] > 0) return true; for (int r = KAI_RES_PODS + 1; r < s->n_res && r < KAI_MAX_RES; r++) if (mig_g[r] > 0 && s->node_allocatable[(size_t)r * N + i] > 0) return true; return false; };
        int first_gpu = -1;
        if (s->node_gpu_memory) for (int i = 0; i < N; i++) if (has_gpus(i)) { if (first_gpu < 0) first_gpu = i; else if (any && s->node_gpu_memory[i] != s->node_gpu_memory[first_gpu]) { err = "shared GPUs with different node_gpu_memory values: leave the cycle to the host path"; return false; } }
        const int64_t M = (s->node_gpu_memory && N > 0) ? s->node_gpu_memory[first_gpu >= 0 ? first_gpu : 0] : 100;
        // per pod, independent of the others: chunks of the pod range on the host's cores; the error reported is the one of the lowest pod
        const int K = chunk_count((size_t)P);
        std::vector<std::string> cerr((size_t)K);
        parallel_chunks((size_t)P, [&](int ci, size_t p0, size_t p1) {
        for (size_t p = p0; p < p1; p++) {
            const double g = s->pod_req[(size_t)KAI_RES_GPU * P + p];
            const double por = s->pod_gpu_portion ? s->pod_gpu_portion[p] : 0.0;
            const int64_t gm = (s->pod_gpu_memory && !(por > 0)) ? s->pod_gpu_memory[p] : 0;
            acc_gpu[p] = g; pend_gpu[p] = g; quota_gpu[p] = g; shared[p] = 0; mem[p] = 0; gmem[p] = 0;
            if (por > 0) { shared[p] = 1; mem[p] = (int64_t)(por * (double)M); }  // GetResourceGpuMemory (node_info.go:653-659); AcceptedResource keeps the portion
            else if (gm > 0) {
                if (gm > M || M <= 0) { cerr[(size_t)ci] = "a gpu-memory request above one device's memory: leave the cycle to the host path"; return; }  // isValidGpuPortion :668-671
                if (g != 0) { cerr[(size_t)ci] = "a gpu-memory request beside a GPU count"; return; }
                if (cfg.min_node_gpu_memory <= 0) { cerr[(size_t)ci] = "gpu-memory requests need kai_config.min_node_gpu_memory > 0"; return; }
                shared[p] = 1; mem[p] = gm; gmem[p] = gm;
                double x = (double)gm / (double)M * 100; double f = (double)(int64_t)x; if (f < x) f += 1;  // getGpuMemoryFractionalOnNode :329-332 (ceil to 1/100)
                const double frac = f / 100;
                acc_gpu[p] = (double)(int64_t)(frac * 100.0 + 0.5) / 100.0;  // GPUs() of NewGpuResourceRequirementWithMultiFraction(1, that portion, …): fixed point 1/100
                pend_gpu[p] = cfg.min_node_gpu_memory > 0 ? (double)gm / (double)cfg.min_node_gpu_memory : 0.0;  // proportion.go:360-366, allocation_info.go:103-107
            }
            // MIG instances: GetGpusQuota adds weight x count to every GPU quantity of the request except GPUs() itself (gpu_resource_requirment.go:163-178)
            double mq = 0; if (mig) for (int r = KAI_RES_PODS + 1; r < s->n_res && r < KAI_MAX_RES; r++) if (mig_g[r] > 0) mq += (double)mig_g[r] * s->pod_req[(size_t)r * P + p];
            mig_q[p] = mq; if (mig) { acc_gpu[p] += mq; pend_gpu[p] += mq; quota_gpu[p] += mq; }  // (without MIG rows the three stay the request bit for bit: kai_session_open copies that row on the device)
            uint8_t k = 0;
            if (mq > 0) k |= K_MIG;
            if (s->pod_flags && (s->pod_flags[p] & KAI_POD_LEGACY_MIG)) k |= K_LEGACY;
            if (!(k & K_MIG) && !shared[p]) k |= K_REGULAR;
            if (g > 0) k |= K_GPUS;
            kind[p] = k;
        }
        });
        for (int ci = 0; ci < K; ci++) if (!cerr[(size_t)ci].empty()) { err = cerr[(size_t)ci]; return false; }
        return true;
    }
};

struct HostPrep {
    // (the arrays with an element per pod / job / pod-set are raw_vectors: the parallel loop that fills them is their first touch — a plain vector's resize would
    // zero tens of megabytes, page by page, on one core first)
    std::vector<int32_t> child_off, children, depth, job_off, depth_order, lvl_off, lvl_parents;
    raw_vector<int32_t> sorted, jobs_static, slot_queue;
    std::vector<QShare> shares;
    int n_levels = 0;
    // nodes in name-rank order: perm[i] = caller's index of the node with rank i
    std::vector<int32_t> perm, node_gpu_count, node_class;
    raw_vector<int32_t> pod_node, pod_nominated;
    std::vector<double> node_alloc; std::vector<uint32_t> node_flags;
    // scan classes
    std::vector<ClassRec> classes; raw_vector<int32_t> pod_scls; int all_tracked = 1, fast_ok = 1;
    // topology + sub-group tree (defaults synthesised when the snapshot carries none)
    int T = 0, TL = 0, D = 0, G = 0;
    bool groups_default = false;   // no sub-group tree in the snapshot: the group / pod-set tables below are the synthesised identity tables (kai_session_open writes their constants on the device)
    bool any_nominated = false;    // some pod names a nominated node (pod_nominated is -1 everywhere otherwise)
    std::vector<int32_t> topo_level_off, node_domain, dom_level, dom_topo, dom_parent, dom_child_off, dom_children;
    std::vector<uint32_t> dom_id_rank; raw_vector<uint32_t> g_name_rank;
    raw_vector<int32_t> g_job, g_parent, g_topo, g_req, g_pref, j_root_group, g_child_off, g_children, s_group, s_topo, s_req, s_pref;
    raw_vector<uint8_t> j_has_topology;
    // v[0 .. n) = val on the host's cores (v: a raw_vector — resize leaves the elements to this loop)
    template <class V, class X> static void par_fill(V& v, size_t n, X val) {
        v.resize(n);
        parallel_chunks(n, [&](int, size_t a, size_t b) { std::fill(v.begin() + (ptrdiff_t)a, v.begin() + (ptrdiff_t)b, (typename V::value_type)val); }, 65536);
    }
    // the message of the lowest index i in [0, n) for which check(i) returns one (nullptr: none fails) — what a sequential loop that stops at its first failure reports
    template <class F> static const char* first_failure(size_t n, F&& check) {
        const int K = chunk_count(n);
        std::vector<const char*> e((size_t)K, nullptr);
        parallel_chunks(n, [&](int ci, size_t a, size_t b) { for (size_t i = a; i < b; i++) if (const char* m = check(i)) { e[(size_t)ci] = m; return; } });
        for (const char* m : e) if (m) return m;
        return nullptr;
    }
    // batch path (kai_batch.hpp): queue nodes by height (leaf = 0, the virtual root at index Q on top), and whether the snapshot's quantities add
    // exactly in any order (the batch path sums shares and node accounting in parallel)
    struct BatchShape { int n_leaves = 0, n_heights = 0; std::vector<int> h_count; };
    std::vector<int32_t> q_islot; int n_inner = 0;  // inner queue nodes numbered in h_nodes order behind the leaves (-1: leaf / virtual root)
    std::vector<int32_t> q_height, h_off, h_nodes; int n_heights = 0, batch_ok = 0, exact_sums = 0; BatchShape shape;  // exact_sums: sums of node / pod quantities do not depend on the order of addition

    // the required arrays of a snapshot with that many nodes / pods / pod-sets / jobs / queues: the first one that is missing, as a message (nullptr: all there)
    static const char* missing_array(const kai_snapshot_soa* s) {
        const int N = s->n_nodes, P = s->n_pods, S = s->n_podsets, J = s->n_jobs, Q = s->n_queues;
        if (N > 0 && (!s->node_allocatable || !s->node_flags || !s->node_name_rank)) return "a required node array is NULL";
        if (P > 0 && (!s->pod_req || !s->pod_job || !s->pod_podset || !s->pod_status || !s->pod_node || !s->pod_uid_rank)) return "a required pod array is NULL";
        if (S > 0 && (!s->podset_job || !s->podset_min_available || !s->podset_name_rank)) return "a required pod-set array is NULL";
        if (J > 0 && (!s->job_queue || !s->job_priority || !s->job_preemptible || !s->job_created_ns || !s->job_uid_rank || !s->job_first_pod || !s->job_n_pods || !s->job_first_podset || !s->job_n_podsets)) return "a required job array is NULL";
        if (Q > 0 && (!s->queue_parent || !s->queue_priority || !s->queue_created_ns || !s->queue_uid_rank || !s->queue_deserved || !s->queue_limit || !s->queue_oqw)) return "a required queue array is NULL";
        return nullptr;
    }
    double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // host clocks of build(): range checks / nodes / pods / task order / queues + job lists / shares + topology / classes / batch shape
    // returns 0 or KAI_ERR_INVALID_ARG with err set
    int build(const kai_config& cfg, const kai_snapshot_soa* s, std::string& err) {
        const int N = s->n_nodes, P = s->n_pods, J = s->n_jobs, Q = s->n_queues, R = s->n_res;
        auto fail = [&](const char* m) { err = m; return (int)KAI_ERR_INVALID_ARG; };
        for (double& x : phase_ms) x = 0;  // (the object may be one a handle keeps between sessions: everything build() hands out is rewritten below)
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](int i) { const auto t = std::chrono::steady_clock::now(); phase_ms[i] += std::chrono::duration<double, std::milli>(t - t_last).count(); t_last = t; };
        // ---- the snapshot indexes device arrays directly: every index is range-checked here, every required array must be present
        const int S = s->n_podsets;
        if (const char* m = missing_array(s)) return fail(m);
        // (each loop on the host's cores; the failure reported is the one a sequential pass would stop at)
        if (const char* m = first_failure((size_t)S, [&](size_t k) -> const char* { return (s->podset_job[k] < 0 || s->podset_job[k] >= J) ? "podset_job out of range" : nullptr; })) return fail(m);
        if (const char* m = first_failure((size_t)J, [&](size_t j) -> const char* {
                const int b = s->job_first_podset[j], n = s->job_n_podsets[j];
                if (b < 0 || n < 0 || b + n > S) return "job pod-set range out of bounds";
                if (s->job_queue[j] < -1) return "bad job_queue";
                return nullptr; })) return fail(m);
        if (const char* m = first_failure((size_t)P, [&](size_t p) -> const char* {
                const int j = s->pod_job[p];
                if (j < -1 || j >= J) return "pod_job out of range";
                if (j >= 0) { const int ps = s->pod_podset[p]; if (ps < s->job_first_podset[j] || ps >= s->job_first_podset[j] + s->job_n_podsets[j]) return "pod_podset outside its job's pod-sets"; }
                else if (s->pod_podset[p] < -1 || s->pod_podset[p] >= S) return "pod_podset out of range";
                return nullptr; })) return fail(m);
        for (int d = 0; d < s->n_domains; d++) if (s->domain_parent && (s->domain_parent[d] < -1 || s->domain_parent[d] >= s->n_domains)) return fail("domain_parent out of range");
        for (int g = 0; g < s->n_groups; g++) if (s->group_parent && s->group_parent[g] < -1) return fail("group_parent out of range");
        lap(0);
        // ---- nodes: permute into name-rank order (framework/session.go:480-485 breaks score ties by node name)
        perm.assign(N, -1);
        for (int i = 0; i < N; i++) { uint32_t rk = s->node_name_rank[i]; if (rk >= (uint32_t)N || perm[rk] >= 0) return fail("node_name_rank must be a permutation of 0..N-1"); perm[rk] = i; }
        node_alloc.resize((size_t)R * N); node_flags.resize(N); node_gpu_count.resize(N); node_class.resize(N);
        for (int i = 0; i < N; i++) {
            int o = perm[i];
            for (int r = 0; r < R; r++) node_alloc[(size_t)r * N + i] = s->node_allocatable[(size_t)r * N + o];
            node_flags[i] = s->node_flags[o] & 0x0FFFFFFFu /* the four highest bits are the library's own (kai_engine.hpp KAI_NODE_*_I) */; node_gpu_count[i] = s->node_gpu_count ? s->node_gpu_count[o] : -1; node_class[i] = s->node_class ? s->node_class[o] : 0;
            if (node_class[i] < 0 || node_class[i] >= std::max(1, s->n_node_classes)) return fail("node_class out of range");
        }
        lap(1);
        pod_node.resize(P); pod_nominated.resize(P);
        { std::vector<char> bad((size_t)chunk_count((size_t)P), 0);
          parallel_chunks((size_t)P, [&](int ci, size_t p0, size_t p1) {
              char f = 0;
              for (size_t p = p0; p < p1; p++) {
                  int n = s->pod_node[p]; pod_node[p] = (n >= 0 && n < N) ? (int32_t)s->node_name_rank[n] : -1;
                  int m = s->pod_nominated_node ? s->pod_nominated_node[p] : -1; pod_nominated[p] = (m >= 0 && m < N) ? (int32_t)s->node_name_rank[m] : -1;
                  if (pod_nominated[p] != -1) f |= 2;
                  int pc = s->pod_class ? s->pod_class[p] : 0; if (pc < 0 || pc >= std::max(1, s->n_pod_classes)) f |= 1;
              }
              bad[(size_t)ci] = f;
          });
          any_nominated = false;
          for (char b : bad) { if (b & 1) return fail("pod_class out of range"); if (b & 2) any_nominated = true; } }
        lap(2);
        // each job's pods in TaskOrderFn order (framework/session_plugins.go:244-260 + plugins/taskorder/task_order.go:28-63)
        sorted.resize(P);
        const bool taskorder = cfg.plugins & KAI_PLUGIN_TASKORDER;
        if (const char* m = first_failure((size_t)J, [&](size_t j) -> const char* { const int b = s->job_first_pod[j], n = s->job_n_pods[j]; return (b < 0 || n < 0 || b + n > P) ? "job pod range out of bounds" : nullptr; })) return fail(m);
        {   // … and disjoint: the per-job sorts below run on the host's cores, two jobs over one slice of `sorted` would race (jobs in pod order is the usual case, checked in O(J))
            bool mono = true; int end = 0;
            for (int j = 0; j < J && mono; j++) { const int b = s->job_first_pod[j], n = s->job_n_pods[j]; if (n == 0) continue; if (b < end) mono = false; end = b + n; }
            if (!mono) {
                std::vector<char> own((size_t)P, 0);
                for (int j = 0; j < J; j++) for (int b = s->job_first_pod[j], i = 0; i < s->job_n_pods[j]; i++) { if (own[(size_t)b + i]) return fail("job pod ranges overlap"); own[(size_t)b + i] = 1; }
            }
        }
        parallel_chunks((size_t)P, [&](int, size_t p0, size_t p1) { for (size_t p = p0; p < p1; p++) sorted[p] = (int32_t)p; });
        parallel_chunks((size_t)J, [&](int, size_t j0, size_t j1) {
            for (size_t j = j0; j < j1; j++) {
                int b = s->job_first_pod[j], n = s->job_n_pods[j];
                if (n > 1) std::sort(sorted.begin() + b, sorted.begin() + b + n, [&](int l, int r) {
                    if (taskorder) {
                        bool ll = s->pod_flags && (s->pod_flags[l] & KAI_POD_HAS_TASK_PRIORITY), rl = s->pod_flags && (s->pod_flags[r] & KAI_POD_HAS_TASK_PRIORITY);
                        if (ll != rl) return ll;
                        if (ll && rl && s->pod_task_priority[l] != s->pod_task_priority[r]) return s->pod_task_priority[l] > s->pod_task_priority[r];
                    }
                    int64_t lc = s->pod_created_ns ? s->pod_created_ns[l] : 0, rc2 = s->pod_created_ns ? s->pod_created_ns[r] : 0;
                    if (lc != rc2) return lc < rc2;
                    return s->pod_uid_rank[l] < s->pod_uid_rank[r];
                });
            }
        }, 2048);
        lap(3);
        // queue children CSR with a virtual root at index Q (cache/cluster_info/queue.go:95-103)
        child_off.assign(Q + 2, 0); children.assign(std::max(Q, 1), 0); depth.assign(Q, 0);
        for (int q = 0; q < Q; q++) { int par = s->queue_parent[q]; if (par < -1 || par >= Q || par == q) return fail("bad queue_parent"); child_off[(par < 0 ? Q : par) + 1]++; }
        for (int i = 0; i < Q + 1; i++) child_off[i + 1] += child_off[i];
        { std::vector<int32_t> fill(child_off.begin(), child_off.end() - 1); for (int q = 0; q < Q; q++) { int par = s->queue_parent[q]; children[fill[par < 0 ? Q : par]++] = q; } }
        for (int q = 0; q < Q; q++) { int d = 0; for (int x = s->queue_parent[q]; x >= 0; x = s->queue_parent[x]) { if (++d > Q) return fail("queue cycle"); } depth[q] = d; }
        // per-queue job lists in the static part of JobOrderFn (session_plugins.go:227-242): priority desc, creation, uid.
        // The elastic state, the only dynamic operand, is applied on the device (k_leaf_init).
        // (a counting sort by queue over chunks of the job range: per chunk a histogram, the chunks' write positions from the histograms of the chunks before them —
        // every queue's jobs come out in ascending job index, as one pass over the jobs leaves them)
        job_off.assign(Q + 1, 0); jobs_static.resize((size_t)std::max(J, 1)); slot_queue.resize((size_t)std::max(J, 1));
        {
            const int K = chunk_count((size_t)J);
            std::vector<std::vector<int32_t>> hist((size_t)K); std::vector<char> badq((size_t)K, 0);
            parallel_chunks((size_t)J, [&](int ci, size_t j0, size_t j1) {
                std::vector<int32_t>& h = hist[(size_t)ci]; h.assign((size_t)Q + 1, 0);
                for (size_t j = j0; j < j1; j++) { const int q = s->job_queue[j]; if (q >= Q) { badq[(size_t)ci] = 1; return; } if (q >= 0) h[(size_t)q]++; }
            });
            for (char b : badq) if (b) return fail("bad job_queue");
            for (int q = 0; q < Q; q++) { int32_t n = 0; for (int ci = 0; ci < K; ci++) n += hist[(size_t)ci].empty() ? 0 : hist[(size_t)ci][(size_t)q]; job_off[q + 1] = job_off[q] + n; }
            for (int q = 0; q < Q; q++) { int32_t at = job_off[q]; for (int ci = 0; ci < K; ci++) { if (hist[(size_t)ci].empty()) continue; const int32_t n = hist[(size_t)ci][(size_t)q]; hist[(size_t)ci][(size_t)q] = at; at += n; } }  // histogram -> write position
            parallel_chunks((size_t)J, [&](int ci, size_t j0, size_t j1) {
                std::vector<int32_t>& at = hist[(size_t)ci];
                for (size_t j = j0; j < j1; j++) { const int q = s->job_queue[j]; if (q >= 0) jobs_static[(size_t)at[(size_t)q]++] = (int32_t)j; }
            });
            // slots past the queued jobs (jobs without a queue leave the lists shorter than J): the values a zeroed / -1-filled array would hold there
            std::fill(jobs_static.begin() + (J > 0 ? job_off[Q] : 0), jobs_static.end(), 0); std::fill(slot_queue.begin() + (J > 0 ? job_off[Q] : 0), slot_queue.end(), -1);
        }
        const bool use_prio = cfg.plugins & KAI_PLUGIN_PRIORITY;
        parallel_chunks((size_t)Q, [&](int, size_t q0, size_t q1) {
            for (size_t q = q0; q < q1; q++) {
                std::sort(jobs_static.begin() + job_off[q], jobs_static.begin() + job_off[q + 1], [&](int l, int r) {
                    if (use_prio && s->job_priority[l] != s->job_priority[r]) return s->job_priority[l] > s->job_priority[r];
                    if (s->job_created_ns[l] != s->job_created_ns[r]) return s->job_created_ns[l] < s->job_created_ns[r];
                    return s->job_uid_rank[l] < s->job_uid_rank[r];
                });
                for (int i = job_off[q]; i < job_off[q + 1]; i++) slot_queue[i] = (int32_t)q;
            }
        }, 4);
        depth_order.resize(Q);
        for (int q = 0; q < Q; q++) depth_order[q] = q;
        std::stable_sort(depth_order.begin(), depth_order.end(), [&](int a, int b) { return depth[a] > depth[b]; });
        // fair-share levels: parents (incl. the virtual root) that have children, grouped by depth
        lvl_off.assign(1, 0); lvl_parents.clear();
        int maxd = 0; for (int q = 0; q < Q; q++) maxd = std::max(maxd, depth[q]);
        lvl_parents.push_back(Q); lvl_off.push_back(1);
        for (int d = 0; d <= maxd; d++) {
            for (int q = 0; q < Q; q++) if (depth[q] == d && child_off[q + 1] > child_off[q]) lvl_parents.push_back(q);
            if ((int)lvl_parents.size() > lvl_off.back()) lvl_off.push_back((int)lvl_parents.size());
        }
        n_levels = (int)lvl_off.size() - 1;
        lap(4);
        // proportion.createQueueResourceAttrs (plugins/proportion/proportion.go:307-345)
        shares.assign((size_t)std::max(Q, 1) * 3, QShare{});
        for (int q = 0; q < Q; q++) for (int k = 0; k < 3; k++) {
            QShare& x = shares[(size_t)q * 3 + k];
            double des = s->queue_deserved[(size_t)k * Q + q], lim = s->queue_limit[(size_t)k * Q + q];
            if (k == KAI_Q_MEM) { des = std::max(KAI_UNLIMITED, des * 1000000.0); lim = std::max(KAI_UNLIMITED, lim * 1000000.0); }
            x.deserved = des; x.max_allowed = lim; x.oqw = s->queue_oqw[(size_t)k * Q + q]; x.usage = s->queue_usage ? s->queue_usage[(size_t)k * Q + q] : 0.0;
        }
        if (int rc = build_topology(s, err)) return rc;
        lap(5);
        build_classes(cfg, s);
        lap(6);
        build_batch(cfg, s);
        lap(7);
        return 0;
    }

    void build_batch(const kai_config& cfg, const kai_snapshot_soa* s) {
        const int N = s->n_nodes, P = s->n_pods, Q = s->n_queues, R = s->n_res;
        q_height.assign(Q + 1, 0);
        for (int i = 0; i < Q; i++) { int q = depth_order[i], par = s->queue_parent[q] < 0 ? Q : s->queue_parent[q]; q_height[par] = std::max(q_height[par], q_height[q] + 1); }  // deepest first
        if (Q == 0) q_height[Q] = 1;
        n_heights = q_height[Q] + 1;
        h_off.assign(n_heights + 1, 0); h_nodes.assign(Q + 1, 0); shape = BatchShape{}; shape.n_heights = n_heights; shape.h_count.assign(n_heights, 0);
        for (int q = 0; q <= Q; q++) { h_off[q_height[q] + 1]++; shape.h_count[q_height[q]]++; }
        for (int h = 0; h < n_heights; h++) h_off[h + 1] += h_off[h];
        { std::vector<int32_t> fill(h_off.begin(), h_off.end() - 1); for (int q = 0; q <= Q; q++) h_nodes[fill[q_height[q]]++] = q; }
        q_islot.assign((size_t)Q + 1, -1); n_inner = 0;
        for (int i = n_heights > 1 ? h_off[1] : Q + 1; i <= Q; i++) { const int q = h_nodes[i]; if (q < Q) { q_islot[(size_t)q] = i - h_off[1]; n_inner = i - h_off[1] + 1; } }
        for (int q = 0; q < Q; q++) if (child_off[q + 1] == child_off[q]) shape.n_leaves++;
        // exact sums: per resource every quantity is a non-negative integer multiple of one power of two, and the totals stay below 2^53 units
        exact_sums = 1;
        for (int r = 0; r < R && exact_sums; r++) {
            // one pass per array on the host's cores: every value a non-negative integer below 2^63, the OR of all values (its trailing zeros = the common
            // power-of-two unit) and the 128-bit sums of the raw values (sum / unit = the sum of the units: every value is a multiple of the unit)
            struct Acc { bool ok = true; uint64_t bits = 0; unsigned __int128 sum = 0; };
            auto pass = [&](const double* v, size_t n) {
                std::vector<Acc> acc((size_t)chunk_count(n));
                parallel_chunks(n, [&](int ci, size_t i0, size_t i1) {
                    Acc a;
                    for (size_t i = i0; i < i1 && a.ok; i++) { const double x = v[i]; if (!(x >= 0) || x >= 9.2e18 || x != (double)(uint64_t)x) { a.ok = false; break; } a.bits |= (uint64_t)x; a.sum += (uint64_t)x; }  // (0 <= x < 2^63: the truncation is floor)
                    acc[(size_t)ci] = a;
                });
                Acc t; for (const Acc& a : acc) { t.ok = t.ok && a.ok; t.bits |= a.bits; t.sum += a.sum; }
                return t;
            };
            const Acc an = pass(s->node_allocatable + (size_t)r * N, (size_t)N), ap = pass(s->pod_req + (size_t)r * P, (size_t)P);
            if (!an.ok || !ap.ok) { exact_sums = 0; break; }
            const uint64_t bits = an.bits | ap.bits;
            if (!bits) continue;
            const int tz = __builtin_ctzll(bits);
            const unsigned __int128 lim = (unsigned __int128)1 << 52;
            if ((an.sum >> tz) >= lim || (ap.sum >> tz) >= lim) exact_sums = 0;
        }
        batch_ok = cfg.engine_mode == 0 && R <= 4 && n_heights <= 16 && exact_sums;
    }

    // Topology domain tree and sub-group tree in the engine's indexing (plugins/topology/topology_plugin.go:57-110,
    // api/podgroup_info/subgroup_info/subgroupset.go).  Children of a domain are appended in the caller's node-index order, as the
    // oracle does (the reference ranges a Go map; the order only matters below a level that sortTree has sorted).
    int build_topology(const kai_snapshot_soa* s, std::string& err) {
        const int N = s->n_nodes, J = s->n_jobs, S = s->n_podsets;
        auto fail = [&](const char* m) { err = m; return (int)KAI_ERR_INVALID_ARG; };
        T = s->n_topologies; TL = s->n_topo_levels; D = s->n_domains;
        if (T < 0 || TL < 0 || D < 0) return fail("negative topology dimension");
        topo_level_off.assign(T + 1, 0);
        for (int t = 0; t <= T && T > 0; t++) topo_level_off[t] = s->topo_level_off[t];
        if (T > 0 && topo_level_off[T] != TL) return fail("topo_level_off / n_topo_levels mismatch");
        node_domain.assign((size_t)std::max(TL, 1) * std::max(N, 1), -1);
        for (int l = 0; l < TL; l++) for (int i = 0; i < N; i++) { int d = s->node_domain[(size_t)l * N + perm[i]]; if (d >= D) return fail("node_domain out of range"); node_domain[(size_t)l * N + i] = d; }
        dom_level.assign(D + T, -1); dom_topo.assign(D + T, 0); dom_parent.assign(D + T, -1); dom_id_rank.assign(D + T, 0);
        for (int d = 0; d < D; d++) {
            int gl = s->domain_level[d], t = -1;
            for (int x = 0; x < T; x++) if (gl >= topo_level_off[x] && gl < topo_level_off[x + 1]) t = x;
            if (t < 0) return fail("domain_level out of range");
            dom_topo[d] = t; dom_level[d] = gl - topo_level_off[t]; dom_parent[d] = s->domain_parent[d] >= 0 ? s->domain_parent[d] : D + t; dom_id_rank[d] = s->domain_id_rank[d];
        }
        for (int t = 0; t < T; t++) { dom_topo[D + t] = t; dom_level[D + t] = -1; dom_parent[D + t] = -1; }
        std::vector<std::vector<int32_t>> kids(D + T);
        for (int t = 0; t < T; t++) {
            const int L = topo_level_off[t + 1] - topo_level_off[t];
            for (int o = 0; o < N; o++) {  // caller's node order
                if (L == 0 || s->node_domain[(size_t)topo_level_off[t] * N + o] < 0) continue;
                int child = -1;
                for (int l = L - 1; l >= 0; l--) {
                    int d = s->node_domain[(size_t)(topo_level_off[t] + l) * N + o];
                    if (d < 0) return fail("node_domain: a node of a topology must have a domain at every level");
                    if (child >= 0 && std::find(kids[d].begin(), kids[d].end(), child) == kids[d].end()) kids[d].push_back(child);
                    child = d;
                }
                if (std::find(kids[D + t].begin(), kids[D + t].end(), child) == kids[D + t].end()) kids[D + t].push_back(child);
            }
        }
        dom_child_off.assign(D + T + 1, 0); dom_children.clear();
        for (int d = 0; d < D + T; d++) { dom_child_off[d] = (int)dom_children.size(); dom_children.insert(dom_children.end(), kids[d].begin(), kids[d].end()); }
        dom_child_off[D + T] = (int)dom_children.size();
        if (dom_children.empty()) dom_children.push_back(0);
        // sub-group tree
        groups_default = !(s->n_groups > 0);
        if (s->n_groups > 0) {
            G = s->n_groups;
            g_job.assign(s->group_job, s->group_job + G); g_parent.assign(s->group_parent, s->group_parent + G); g_name_rank.assign(s->group_name_rank, s->group_name_rank + G);
            g_topo.assign(s->group_topology, s->group_topology + G); g_req.assign(s->group_required_level, s->group_required_level + G); g_pref.assign(s->group_preferred_level, s->group_preferred_level + G);
            j_root_group.assign(s->job_root_group, s->job_root_group + J);
            s_group.assign(s->podset_group, s->podset_group + S); s_topo.assign(s->podset_topology, s->podset_topology + S);
            s_req.assign(s->podset_required_level, s->podset_required_level + S); s_pref.assign(s->podset_preferred_level, s->podset_preferred_level + S);
        } else {
            // no sub-group tree in the snapshot: one root group per job, no topology anywhere — identity tables, written on the host's cores (J and S are ~4·10^5 at config 5);
            // the checks and the child lists below have nothing to find in them (every parent -1, every topology -1, podset_job range-checked by build())
            G = J;
            g_job.resize((size_t)J); g_parent.resize((size_t)J); g_name_rank.resize((size_t)J); g_topo.resize((size_t)J); g_req.resize((size_t)J); g_pref.resize((size_t)J); j_root_group.resize((size_t)J);
            g_child_off.resize((size_t)G + 1); j_has_topology.resize((size_t)std::max(J, 1));
            parallel_chunks((size_t)J, [&](int, size_t j0, size_t j1) {
                for (size_t j = j0; j < j1; j++) { g_job[j] = (int32_t)j; j_root_group[j] = (int32_t)j; g_parent[j] = -1; g_name_rank[j] = 0; g_topo[j] = -1; g_req[j] = -1; g_pref[j] = -1; g_child_off[j] = 0; j_has_topology[j] = 0; }
            }, 65536);
            g_child_off[(size_t)G] = 0; if (J == 0) j_has_topology[0] = 0;
            s_group.resize((size_t)S); s_topo.resize((size_t)S); s_req.resize((size_t)S); s_pref.resize((size_t)S);
            parallel_chunks((size_t)S, [&](int, size_t k0, size_t k1) { for (size_t k = k0; k < k1; k++) { s_group[k] = s->podset_job[k]; s_topo[k] = -1; s_req[k] = -1; s_pref[k] = -1; } }, 65536);
            g_children.assign(1, 0);
            return 0;
        }
        for (int g = 0; g < G; g++) { if (g_parent[g] >= G || g_job[g] < 0 || g_job[g] >= J) return fail("bad group table"); if (g_topo[g] >= T) return fail("group_topology out of range"); }
        for (int k = 0; k < S; k++) { if (s_group[k] < 0 || s_group[k] >= G) return fail("bad podset_group"); if (s_topo[k] >= T) return fail("podset_topology out of range"); }
        g_child_off.assign(G + 1, 0);  // children of a group in ascending group index (counting sort: no vector per group)
        for (int g = 0; g < G; g++) if (g_parent[g] >= 0) g_child_off[g_parent[g] + 1]++;
        for (int g = 0; g < G; g++) g_child_off[g + 1] += g_child_off[g];
        g_children.assign((size_t)std::max(g_child_off[G], 1), 0);
        if (g_child_off[G] > 0) { std::vector<int32_t> fillg(g_child_off.begin(), g_child_off.end() - 1); for (int g = 0; g < G; g++) if (g_parent[g] >= 0) g_children[fillg[g_parent[g]]++] = g; }
        j_has_topology.assign(std::max(J, 1), 0);
        for (int g = 0; g < G; g++) if (g_topo[g] != -1) j_has_topology[g_job[g]] = 1;
        for (int k = 0; k < S; k++) if (s_topo[k] != -1) j_has_topology[s->podset_job[k]] = 1;
        for (int g = 0; g < G; g++) if (g_parent[g] >= 0) j_has_topology[g_job[g]] = 1;  // nested sub-group sets also take the general path
        return 0;
    }

    // Scan classes: pods with the same request vector and static-predicate class see every node identically, so the device
    // keeps one arg-max index per class (kai_engine.hpp class_key).  A class is admitted only when its key order provably equals
    // the reference's f64 score order:
    //   bin-pack on resource r: every node quantity and every pod request of r is an integer <= 2^30 (then distinct
    //     Idle+Releasing values give pack scores at least 9·2^-30 apart, far above the rounding of the score sum), and for the
    //     CPU resource every node has Allocatable > 0 (a zero-capacity node scores 0, not by its free amount);
    //   spread on r: the divisor (gpu.count label or Allocatable) is an integer in [Allocatable, 2^22] so the ratio stays in [0,1]
    //     and distinct ratios stay distinct after the sum.
    // The KAI_CMAX most frequent admissible classes among PENDING pods are indexed; other pods use the brute-force scan.
    // Shared GPUs (round 6): a pod that asks for a fraction (or MiB) of one device is in no class — its fit, predicates and score read the node's GPU groups (brute-force scan) —
    // and does not count in the integer guard of the GPU column; the other pods' classes stay indexed, their keys carry the gpusharingorder bit (kai_engine.hpp key_shared_layout).
    // MIG rows keep every scan on brute force (the caller turns the index off).
    const SharedPods* shared_pods = nullptr;  // set by the caller before build() when the snapshot has shared-GPU requests
    void build_classes(const kai_config& cfg, const kai_snapshot_soa* s) {
        const int N = s->n_nodes, P = s->n_pods, R = s->n_res;
        const uint8_t* sh = (shared_pods && shared_pods->any) ? shared_pods->shared.data() : nullptr;
        par_fill(pod_scls, (size_t)P, -1); classes.clear(); all_tracked = 1;
        // the staged job path needs "fits on Idle+Releasing" == "fits on Idle" for every node: nothing releasing, nothing pipelined
        fast_ok = cfg.engine_mode == 2 ? 0 : 1;
        auto integral = [](double v, double lim) { return v >= 0 && v <= lim && v == (double)(int64_t)v; };  // (v <= lim < 2^63: the truncation is floor for v >= 0)
        bool ok_res[2] = {true, true};  // [0] = CPU, [1] = GPU as placement resource
        const int rr[2] = {KAI_RES_CPU, KAI_RES_GPU};
        for (int t = 0; t < 2; t++) {
            int r = rr[t]; int strat = t == 0 ? cfg.cpu_strategy : cfg.gpu_strategy;
            for (int n = 0; n < N && ok_res[t]; n++) {
                double a = s->node_allocatable[(size_t)r * N + n];
                if (strat == KAI_SPREAD) {
                    double cnt = a;
                    if (r == KAI_RES_GPU && s->node_gpu_count && s->node_gpu_count[n] >= 0) cnt = (double)s->node_gpu_count[n]; else if (r == KAI_RES_GPU) cnt = (double)(int64_t)a;
                    if (!integral(a, 4194304.0) || !(cnt >= a) || cnt > 4194304.0) ok_res[t] = false;
                } else {
                    if (!integral(a, 1073741824.0)) ok_res[t] = false;
                    if (r == KAI_RES_CPU && a == 0) ok_res[t] = false;
                }
            }
        }
        {   // per pod, one pass on the host's cores: a releasing / pipelined pod anywhere, a CPU or GPU request that is no integer <= 2^30
            const int K = chunk_count((size_t)P);
            std::vector<unsigned char> flags((size_t)K, 0);
            const double* cpu = s->pod_req + (size_t)KAI_RES_CPU * P; const double* gpu = s->pod_req + (size_t)KAI_RES_GPU * P;
            parallel_chunks((size_t)P, [&](int ci, size_t p0, size_t p1) {
                unsigned char f = 0;
                for (size_t p = p0; p < p1; p++) {
                    if (s->pod_status[p] & (KAI_POD_RELEASING | KAI_POD_PIPELINED)) f |= 1;
                    if (!integral(cpu[p], 1073741824.0)) f |= 2;
                    if (!(sh && sh[p]) && !integral(gpu[p], 1073741824.0)) f |= 4;
                }
                flags[(size_t)ci] = f;
            });
            for (unsigned char f : flags) { if (f & 1) fast_ok = 0; if (f & 2) ok_res[0] = false; if (f & 4) ok_res[1] = false; }
        }
        if (cfg.engine_mode == 1) { all_tracked = 0; return; }
        struct Key { double req[KAI_MAX_RES]; int32_t pc; bool operator<(const Key& o) const { int c = std::memcmp(req, o.req, sizeof req); return c ? c < 0 : pc < o.pc; } };
        // class ids in order of first appearance over the pods.  Chunks of the pod range are classified on the host's cores (a chunk's keys in ITS order of first
        // appearance; a pod usually repeats the key of the pod before it, so that one is compared first), then merged chunk by chunk: the ids come out as in one pass
        std::map<Key, int> ids; std::vector<Key> keys; std::vector<int64_t> freq; raw_vector<int32_t> pod_cls((size_t)P);  // (every element is written by the classification below)
        {
            const int K = chunk_count((size_t)P);
            struct Local { std::vector<Key> keys; std::vector<int64_t> freq; bool untracked = false; };
            std::vector<Local> loc((size_t)K);
            auto make_key = [&](size_t p, Key& k) { std::memset(&k, 0, sizeof k); for (int r = 0; r < R; r++) { double v = s->pod_req[(size_t)r * P + p]; k.req[r] = v == 0 ? 0.0 : v; } k.pc = s->pod_class ? s->pod_class[p] : 0; };  // folds -0.0
            parallel_chunks((size_t)P, [&](int ci, size_t p0, size_t p1) {
                Local& L = loc[(size_t)ci]; std::map<Key, int> lid; int last = -1;
                for (size_t p = p0; p < p1; p++) {
                    Key k; make_key(p, k);
                    int id;
                    if (last >= 0 && std::memcmp(&L.keys[(size_t)last], &k, sizeof(Key)) == 0) id = last;
                    else { auto it = lid.find(k); if (it == lid.end()) { id = (int)L.keys.size(); lid[k] = id; L.keys.push_back(k); L.freq.push_back(0); } else id = it->second; }
                    last = id; pod_cls[p] = id;  // chunk-local id for now
                    if (s->pod_status[p] == KAI_POD_PENDING) { if (sh && sh[p]) L.untracked = true; else L.freq[(size_t)id]++; }
                }
            });
            std::vector<std::vector<int>> to_global((size_t)K);
            for (int ci = 0; ci < K; ci++) {
                Local& L = loc[(size_t)ci]; to_global[(size_t)ci].resize(L.keys.size());
                if (L.untracked) all_tracked = 0;
                for (size_t i = 0; i < L.keys.size(); i++) {
                    auto it = ids.find(L.keys[i]);
                    int id; if (it == ids.end()) { id = (int)keys.size(); ids[L.keys[i]] = id; keys.push_back(L.keys[i]); freq.push_back(0); } else id = it->second;
                    to_global[(size_t)ci][i] = id; freq[(size_t)id] += L.freq[i];
                }
            }
            parallel_chunks((size_t)P, [&](int ci, size_t p0, size_t p1) { const std::vector<int>& g = to_global[(size_t)ci]; for (size_t p = p0; p < p1; p++) pod_cls[p] = g[(size_t)pod_cls[p]]; });
        }
        std::vector<int> order(keys.size()); for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return freq[a] > freq[b]; });
        std::vector<int> remap(keys.size(), -1);
        for (int id : order) {
            const Key& k = keys[id];
            bool cpu_only = !(k.req[KAI_RES_GPU] > 0);
            bool admissible = !(cfg.plugins & KAI_PLUGIN_NODEPLACEMENT) || ok_res[cpu_only ? 0 : 1];
            if (sh && (cfg.plugins & KAI_PLUGIN_NODEPLACEMENT) && (cfg.plugins & KAI_PLUGIN_GPUSHARINGORDER) && (cpu_only ? cfg.cpu_strategy : cfg.gpu_strategy) == KAI_SPREAD) admissible = false;  // no room for the sharing bit above a spread key
            if (freq[id] == 0) continue;  // only classes that have pending pods are ever queried by the allocate action
            if (!admissible || (int)classes.size() >= KAI_CMAX) { all_tracked = 0; continue; }
            ClassRec cr; std::memset(&cr, 0, sizeof cr);
            for (int r = 0; r < KAI_MAX_RES; r++) cr.req[r] = k.req[r];
            cr.pod_class = k.pc; cr.cpu_only = cpu_only;
            bool be = !(k.req[KAI_RES_GPU] > 0.01) && !(k.req[KAI_RES_CPU] >= 10.0 || k.req[KAI_RES_MEM] >= 10.0 * 1024 * 1024);
            for (int r = KAI_RES_PODS; r < R; r++) if (k.req[r] >= 10.0) be = false;
            cr.best_effort = be; cr.r_place = cpu_only ? KAI_RES_CPU : KAI_RES_GPU; cr.strategy = cpu_only ? cfg.cpu_strategy : cfg.gpu_strategy;
            remap[id] = (int)classes.size(); classes.push_back(cr);
        }
        parallel_chunks((size_t)P, [&](int, size_t p0, size_t p1) { for (size_t p = p0; p < p1; p++) pod_scls[p] = (sh && sh[p]) ? -1 : remap[(size_t)pod_cls[p]]; });
        int NB = (N + KAI_BLOCK - 1) / KAI_BLOCK, NSB = (NB + 63) / 64;
        if (NSB > KAI_NSB_MAX) { classes.clear(); par_fill(pod_scls, (size_t)P, -1); all_tracked = 0; }  // beyond the LDS level: brute force
    }
};

}  // namespace kai
