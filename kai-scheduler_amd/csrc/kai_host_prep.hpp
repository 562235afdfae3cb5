// kai_host_prep.hpp — index structures derived from a kai_snapshot_soa on the host.
//
// Pure re-orderings of the input (no scheduling arithmetic): each job's pods in TaskOrderFn order, the queue
// tree as CSR with a virtual root, per-queue job lists, queues by depth, fair-share levels, and the
// proportion plugin's per-queue quota records.  Shared by kai_core.hip (uploads them to HBM) and by
// tests/host_sim (which debugs the engine's control flow without a GPU).
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "kai_engine.hpp"

namespace kai {

struct HostPrep {
    std::vector<int32_t> sorted, child_off, children, depth, job_off, jobs_by_queue, depth_order, lvl_off, lvl_parents;
    std::vector<QShare> shares;
    int n_levels = 0;

    // returns 0 or KAI_ERR_INVALID_ARG with err set
    int build(const kai_config& cfg, const kai_snapshot_soa* s, std::string& err) {
        const int P = s->n_pods, J = s->n_jobs, Q = s->n_queues;
        auto fail = [&](const char* m) { err = m; return (int)KAI_ERR_INVALID_ARG; };
        // each job's pods in TaskOrderFn order (framework/session_plugins.go:244-260 + plugins/taskorder/task_order.go:28-63)
        sorted.resize(P);
        for (int p = 0; p < P; p++) sorted[p] = p;
        const bool taskorder = cfg.plugins & KAI_PLUGIN_TASKORDER;
        for (int j = 0; j < J; j++) {
            int b = s->job_first_pod[j], n = s->job_n_pods[j];
            if (b < 0 || n < 0 || b + n > P) return fail("job pod range out of bounds");
            std::sort(sorted.begin() + b, sorted.begin() + b + n, [&](int l, int r) {
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
        // queue children CSR with a virtual root at index Q (cache/cluster_info/queue.go:95-103)
        child_off.assign(Q + 2, 0); children.assign(std::max(Q, 1), 0); depth.assign(Q, 0);
        for (int q = 0; q < Q; q++) { int par = s->queue_parent[q]; if (par < -1 || par >= Q || par == q) return fail("bad queue_parent"); child_off[(par < 0 ? Q : par) + 1]++; }
        for (int i = 0; i < Q + 1; i++) child_off[i + 1] += child_off[i];
        { std::vector<int32_t> fill(child_off.begin(), child_off.end() - 1); for (int q = 0; q < Q; q++) { int par = s->queue_parent[q]; children[fill[par < 0 ? Q : par]++] = q; } }
        for (int q = 0; q < Q; q++) { int d = 0; for (int x = s->queue_parent[q]; x >= 0; x = s->queue_parent[x]) { if (++d > Q) return fail("queue cycle"); } depth[q] = d; }
        // per-queue job lists (leaf job heaps live in these regions)
        job_off.assign(Q + 1, 0); jobs_by_queue.assign(std::max(J, 1), 0);
        for (int j = 0; j < J; j++) { int q = s->job_queue[j]; if (q >= Q) return fail("bad job_queue"); if (q >= 0) job_off[q + 1]++; }
        for (int q = 0; q < Q; q++) job_off[q + 1] += job_off[q];
        { std::vector<int32_t> fill(job_off.begin(), job_off.end() - 1); for (int j = 0; j < J; j++) { int q = s->job_queue[j]; if (q >= 0) jobs_by_queue[fill[q]++] = j; } }
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
        // proportion.createQueueResourceAttrs (plugins/proportion/proportion.go:307-345)
        shares.assign((size_t)std::max(Q, 1) * 3, QShare{});
        for (int q = 0; q < Q; q++) for (int k = 0; k < 3; k++) {
            QShare& x = shares[(size_t)q * 3 + k];
            double des = s->queue_deserved[(size_t)k * Q + q], lim = s->queue_limit[(size_t)k * Q + q];
            if (k == KAI_Q_MEM) { des = std::max(KAI_UNLIMITED, des * 1000000.0); lim = std::max(KAI_UNLIMITED, lim * 1000000.0); }
            x.deserved = des; x.max_allowed = lim; x.oqw = s->queue_oqw[(size_t)k * Q + q]; x.usage = s->queue_usage ? s->queue_usage[(size_t)k * Q + q] : 0.0;
        }
        return 0;
    }
};

}  // namespace kai
