// kai_batch.hpp — the batch ("plan / fill / apply") execution of the allocate action for gfx950.
//
// Why.  The reference's allocate loop (actions/allocate/allocate.go:46-77) is sequential: pop the best job of the queue tree, place its gang,
// update the shares, pop again.  Executed literally that is one dependent chain (kai_engine.hpp runs it on one lane).  But the chain has
// structure:
//   * QUEUE SIDE.  The state a queue node x of the job-order tree contributes to the pop order (job_order_by_queue.go:61-245) — its shares and
//     the job its comparator key is read through — changes only when a job of x's own subtree is popped.  So, given the outcome of every
//     attempt, the sequence of jobs popped from x's subtree does not depend on what the rest of the tree does, and the parent's choice among its
//     children is a k-way merge of their key sequences: "pop the child whose current key is smallest" over sequences that are NOT monotone equals a
//     sort by each sequence's running maximum.  The whole pop order of a scheduling cycle is therefore a bottom-up pass of segmented scans and
//     merges over the tree — data parallel over all queued jobs — instead of 10^5 dependent heap operations.
//   * NODE SIDE.  A placement needs the arg-max node of its class (kai_engine.hpp class_key) and changes one node; the fill kernel walks the
//     planned order with ONE wavefront that keeps the three-level class index itself (lane = node inside a block, lane = class for the upper levels),
//     so a decision costs a few hundred cycles and no cross-wave hand-off.
//   * The plan needs outcomes (does the gang fit, does the queue-capacity gate pass) before they happen.  Gates are a function of the subtree's own
//     shares and are computed exactly inside the plan; node fit is PREDICTED (a job fails iff one of its classes has no fitting node left at plan
//     time — exact for the rest of the action because free resources only shrink during allocate) and VERIFIED by the fill kernel, which stops at
//     the first job whose outcome differs.  Everything up to and including that job is exactly what the sequential loop would have done; the next
//     round re-plans from there.  A cycle of BASELINE config 5 (78.7 k pops, 163 k placement decisions) takes a few dozen rounds.
//
// The lazy-reorder protocol of the reference is reproduced, not idealised: an inner node's key is read through the CURRENT TOP of its children
// heap, i.e. through the child it popped from last, not through its true best job (getBestJobFromNode, job_order_by_queue.go:309-318, with
// Fix(0) deferred to the next visit :194-217).  `sp` below ("stale-path job") is that job; cur_sp carries it from round to round.
//
// Qualification (k_batch_qualify + HostPrep): allocate action, proportion plugin, class index covering every pending pod, nothing releasing or
// pipelined in the session, infinite queue depth, R <= 4, share quantities that add exactly in any order, every queued job "regular" (one pod-set,
// no topology constraint, below minAvailable, its tasks-to-allocate chunk = all its pending pods so it is never pushed back, no nominated node),
// and sibling queues whose static tie-break (queue_order.go:214-240) is a strict total order.  Anything else runs on the sequential engine
// (kai_engine.hpp) with identical results.
//
// Kernel bodies are written against kai_simt.hpp so that tests/host_sim can run them on the CPU (debug aid, no GPU in the dev container).
#pragma once
#include "kai_engine.hpp"
#include "kai_simt.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define KAI_OPAQUE_F64(x) asm volatile("" : "+v"(x))
#else
#define KAI_OPAQUE_F64(x) (void)0
#endif

namespace kai {

// ------------------------------------------------------------------------------------------------------ keys
KAI_HD bool pk_less(const PlanKey& a, const PlanKey& b) {
    if (a.w0 != b.w0) return a.w0 < b.w0;
    if (a.w1 != b.w1) return a.w1 < b.w1;
    if (a.w2 != b.w2) return a.w2 < b.w2;
    return a.w3 < b.w3;
}
KAI_HD uint64_t pk_orderable(double d) {
    union { double d; uint64_t u; } cv; cv.d = d;
    return (cv.u & 0x8000000000000000ull) ? ~cv.u : (cv.u | 0x8000000000000000ull);
}
KAI_HD int plan_cmp_q(double a, double b) {  // resource_quantities.go:80-97
    if (a == KAI_UNLIMITED) return b == KAI_UNLIMITED ? 0 : 1;
    if (b == KAI_UNLIMITED) return -1;
    return a > b ? 1 : a < b ? -1 : 0;
}

// Operands of queue_order.GetQueueOrderResult (plugins/proportion/queue_order/queue_order.go:19-73) for a queue holding `alloc` and looking at a job that asks for `req`, as one
// ascending key, and the capacity gates of ONE queue of the chain (capacity_policy/max_allowed_check.go:20-66, quota_check.go:27-77) — same f64 expressions as Engine::queue_key /
// dominant_share_l.  The plan computes MANY keys of ONE queue node: what they read of the node (its three QShare records, its priority) is loaded once into a PlanNodeConst — through
// the context's pointers the compiler could not hoist those loads over the scans' stores.
struct PlanNodeConst { double fair[3], deserved[3], max_allowed[3], allocatable[3], alloc_eff[3]; int32_t prio, pad; };  // allocatable: qs_allocatable; alloc_eff: ... with "unlimited" replaced by the cluster total
KAI_HD PlanNodeConst plan_node_const(const KaiCtx& c, int q, double t0, double t1, double t2) {
    PlanNodeConst n; n.prio = c.q_prio[q]; n.pad = 0;
    for (int k = 0; k < 3; k++) {
        const QShare s = c.q_share[(size_t)q * 3 + k];
        n.fair[k] = s.fair; n.deserved[k] = s.deserved; n.max_allowed[k] = s.max_allowed;
        const double a = qs_allocatable(s);
        n.allocatable[k] = a; n.alloc_eff[k] = a == KAI_UNLIMITED ? (k == 0 ? t0 : k == 1 ? t1 : t2) : a;
    }
    return n;
}
KAI_HD PlanKey plan_key_c(const PlanNodeConst& n, const double* alloc, const double* req, int32_t srank) {
    bool over = true, starved = true, viol = false;
    double dwj = 0.0, dnj = 0.0;
    for (int k = 0; k < 3; k++) {
        if (n.fair[k] >= alloc[k]) over = false;
        const double with_job = alloc[k] + req[k];
        if (plan_cmp_q(with_job, n.deserved[k]) > 0) starved = false;
        if (n.allocatable[k] == 0 && with_job > 0) viol = true;
        const double allocatable = n.alloc_eff[k];
        const double vw = allocatable == 0 ? with_job * 1000 : with_job / allocatable;
        const double vn = allocatable == 0 ? alloc[k] * 1000 : alloc[k] / allocatable;
        dwj = kmax(dwj, vw); dnj = kmax(dnj, vn);
    }
    PlanKey r;
    r.w0 = ((uint64_t)over << 34) | ((uint64_t)!starved << 33) | ((uint64_t)(uint32_t)((int64_t)0x7fffffff - (int64_t)n.prio) << 1) | (uint64_t)viol;
    r.w1 = pk_orderable(dwj); r.w2 = pk_orderable(dnj); r.w3 = (uint64_t)(uint32_t)srank;
    return r;
}
KAI_HD bool plan_gate_fails_c(const PlanNodeConst& n, const double* alloc, const double* alloc_np, const double* req, bool np) {
    for (int k = 0; k < 3; k++) {
        if (req[k] == 0) continue;
        if (n.max_allowed[k] != KAI_UNLIMITED && n.max_allowed[k] < alloc[k] + req[k]) return true;
        if (np && n.deserved[k] != KAI_UNLIMITED && n.deserved[k] < alloc_np[k] + req[k]) return true;
    }
    return false;
}
// static part of the queue order between two siblings (steps 7, 8 of queue_order.go:19-73): true = l first
KAI_HD bool plan_static_before(const KaiCtx& c, int lq, int rq) {
    const QShare* L = &c.q_share[(size_t)lq * 3]; const QShare* R = &c.q_share[(size_t)rq * 3];
    bool l_le = true, r_le = true;
    for (int k = 0; k < 3; k++) { int cmp = plan_cmp_q(qs_allocatable(L[k]), qs_allocatable(R[k])); if (cmp > 0) l_le = false; if (cmp < 0) r_le = false; }
    if (!r_le && l_le) return true;
    if (!l_le && r_le) return false;
    return c.q_created[lq] < c.q_created[rq];
}
// number of keys of [keys, keys+n) that are smaller than k (the keys are a running maximum, hence non-decreasing)
KAI_HD int plan_lower_bound(KAI_GP(const PlanKey) keys, int n, const PlanKey& k) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; const PlanKey m = keys[mid]; if (pk_less(m, k)) lo = mid + 1; else hi = mid; }
    return lo;
}

// The class key (kai_engine.hpp class_key_regs) from a node record, with Releasing = 0 and the static predicates read from okmask.
// creq / cflags: the class's request vector and bits (1 = CPU-only request, 2 = placement resource is the GPU, 4 = spread strategy).
KAI_HD uint64_t class_key_rec(uint32_t plugins, int R, const double* creq, uint32_t cflags, int kidx, const NodeRec& s) {
    bool fit = true;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < 4; r++) {
        if (r >= R) continue;
        const double rq = creq[r];
        if (r >= KAI_RES_PODS && !(rq > 0)) continue;
        if (rq > s.idle[r]) fit = false;
    }
    if (!fit) return 0;
    if ((plugins & KAI_PLUGIN_PREDICATES) && !(s.idle[KAI_RES_PODS] > 0)) return 0;
    if (!((s.okmask >> kidx) & 1ull)) return 0;
    uint64_t key = 0;
    if (plugins & KAI_PLUGIN_NODEAVAILABILITY) key |= 1ull << 63;  // fits on Idle alone: nothing is releasing on this path
    if ((plugins & KAI_PLUGIN_RESOURCETYPE) && (cflags & 1u) && s.cpu_node) key |= 1ull << 62;
    uint64_t v = 1;
    if (plugins & KAI_PLUGIN_NODEPLACEMENT) {
        const bool gpu = cflags & 2u;
        // the two candidates as opaque register values: left visible, the compiler folds "select of two loads" into one load at a selected
        // address, which takes the whole record out of registers into scratch memory (a memory round trip per key)
        double ig = s.idle[KAI_RES_GPU], ic = s.idle[KAI_RES_CPU]; KAI_OPAQUE_F64(ig); KAI_OPAQUE_F64(ic);
        const double cur = gpu ? ig : ic;
        if (cflags & 4u) {
            double cg = s.cnt_gpu, cc = s.cnt_cpu; KAI_OPAQUE_F64(cg); KAI_OPAQUE_F64(cc);
            const double count = gpu ? cg : cc;
            const double place = count == 0 ? 0.0 : cur / count;
            union { double d; uint64_t u; } cv; cv.d = place;
            v = cv.u + 1;
        } else {
            v = (uint64_t)((((int64_t)1 << 53) - 1) - (int64_t)(int32_t)cur);  // a bin-pack class is indexed only with integer quantities <= 2^30 (HostPrep::build_classes): one v_cvt_i32_f64
        }
    }
    return key | v;
}
KAI_HD uint32_t class_flags(const ClassRec& k) { return (k.cpu_only ? 1u : 0u) | (k.r_place == KAI_RES_GPU ? 2u : 0u) | (k.strategy == KAI_SPREAD ? 4u : 0u); }
// the node record of node n from the session's node arrays (the static predicates of every scan class folded into okmask)
KAI_HD NodeRec make_node_rec(const KaiCtx& c, int n) {
    NodeRec r; r.idle[0] = r.idle[1] = r.idle[2] = r.idle[3] = 0; r.cnt_gpu = 0; r.cnt_cpu = 0; r.okmask = 0; r.cpu_node = 0; r.pad = 0;
    if (n >= c.N) return r;
    for (int k = 0; k < 4; k++) r.idle[k] = k < c.R ? c.n_idle[(size_t)k * c.N + n] : 0.0;
    const uint32_t f = c.n_flags[n]; const int nc = c.n_class[n], lbl = c.n_gpu_count[n];
    const double ag = c.n_alloc[(size_t)KAI_RES_GPU * c.N + n], ac = c.n_alloc[(size_t)KAI_RES_CPU * c.N + n];
    r.cnt_gpu = lbl >= 0 ? (double)lbl : (double)(int64_t)ag; r.cnt_cpu = ac;
    r.cpu_node = (!(f & KAI_NODE_MIG_ENABLED) && ag <= 0 && !(f & KAI_NODE_HAS_DRA_GPUS)) ? 1u : 0u;
    for (int k = 0; k < c.C; k++) {
        const ClassRec& cr = c.cls[k]; bool ok = true;
        if (c.plugins & KAI_PLUGIN_PREDICATES) {
            if (!cr.cpu_only) { if (f & KAI_NODE_HAS_DRA_GPUS) ok = false; if ((f & KAI_NODE_MIG_ENABLED) && (f & KAI_NODE_MIG_MIXED)) ok = false; }
            if (f & KAI_NODE_NOT_READY) ok = false;
            if (!c.class_fit[(size_t)cr.pod_class * c.n_node_classes + nc]) ok = false;
            if (c.restrict_nodes) { if (!cr.cpu_only) { if (!(f & KAI_NODE_GPU_WORKER)) ok = false; } else if (!(f & KAI_NODE_CPU_WORKER)) ok = false; }
        }
        if (ok) r.okmask |= 1ull << k;
    }
    return r;
}

}  // namespace kai
