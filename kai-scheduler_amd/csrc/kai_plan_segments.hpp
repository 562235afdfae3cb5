// kai_plan_segments.hpp — k_plan_scan for LONG streams, cut into segments that run on different workgroups.
//
// k_plan_scan (kai_batch_kernels.hpp) gives a queue node's merged stream to ONE workgroup: shares along the stream (prefix sums), the node's own gate (the first job it turns away
// ends the stream), the key before each pop (a running maximum).  Config 5's eight top-level queues hold 38 k positions each: eight workgroups of a 256-CU chip worked 0.42 ms per
// full plan on them (`profiles/r06l_*`), bound by the gate's and the keys' f64 divisions on eight compute units.  Here the stream of a node is cut into segments of KPS_SEG positions,
// one workgroup each, and the three scans become what they are across segments — a sum of the segments before, a minimum over all segments, a maximum of the segments before:
//   k_seg_sum    per segment: the sums of its jobs' resources                                                        -> sg_tot
//   k_seg_gate   per segment: shares before each position = node + segments before + local prefix; the gate; the first job turned away (atomic minimum per node) -> d_ab, sg_fb
//   k_seg_keys   per segment: the stream's end V from that minimum; flags of the jobs turned away; the keys and their running maximum inside the segment   -> pk (local), sg_key
//   k_seg_max    per segment: the maximum of the segments before into its keys                                      -> pk
// Sums in another order are exact on this path (HostPrep::batch_units), minimum and maximum are order-free: the outputs are those of k_plan_scan bit for bit (the emulator tests run
// both on the same plans).  The shares pass 2 of k_plan_scan re-sums "with the final flags" need no second scan: the gate only re-flags jobs at or before the first OK job it turns
// away (position f); of those only f itself was in a sum, and only the key behind it (position f + 1 = V) sees it — its shares are the stored ones minus that job's resources.
// Reference: the shares / gate / key arithmetic is kai_batch.hpp's (plugins/proportion/..., queue_order.go — cited there); this file only changes who computes which position.
#pragma once
#include "kai_batch.hpp"

namespace kai {

// (KPS_T threads x KPS_E positions = KPS_SEG positions per segment: kai_batch_types.hpp)
struct SegNode { int x, s, slot0, eb, kb, sumV, V0, compl_all; };
// the node and segment of this workgroup (bid = node's index in its height * segs + segment), the node's stream length before the gate
KW_BODY bool kps_node(const KaiCtx& c, int height, int segs, int32_t& s_sum, int32_t& s_inc, SegNode& n, bool need_v0) {
    const BatchCtx& b = c.bt;
    const int i = kw::bid() / segs, tid = kw::tid(), T = kw::bdim();
    n.s = kw::bid() % segs;
    const int idx = b.h_off[height] + i;
    if (idx >= b.h_off[height + 1]) return false;  // (the same for the whole workgroup)
    n.x = b.h_nodes[idx];
    if (n.x == c.Q) return false;                  // the virtual root's stream is not scanned (k_plan_scan sets its length)
    if (n.s * KPS_SEG > b.q_cnt[n.x]) return false;  // beyond what the node's stream can hold this round (the grid is sized by an upper bound)
    n.eb = b.q_ebase[n.x]; n.kb = b.q_kbase[n.x]; n.slot0 = n.eb / KPS_SEG + i;  // slots of different nodes never meet: regions are disjoint, floor((a+b)/S) >= floor(a/S) + ceil(b/S) - 1
    n.sumV = 0; n.V0 = 0; n.compl_all = 1;
    if (!need_v0) return true;
    if (tid == 0) { s_sum = 0; s_inc = 0; }
    kw::sync();
    { int sv = 0, inc = 0;
      for (int k = c.q_child_off[n.x] + tid; k < c.q_child_off[n.x + 1]; k += T) { const int ch = c.q_children[k]; sv += b.q_valid[ch]; if (!b.q_complete[ch]) inc = 1; }
      if (sv) kw::atomic_add(&s_sum, sv); if (inc) kw::atomic_add(&s_inc, 1); }
    kw::sync();
    n.sumV = s_sum; n.compl_all = s_inc ? 0 : 1;
    n.V0 = b.q_sent[n.x] < n.sumV ? b.q_sent[n.x] : n.sumV;
    return true;
}

KW_BODY void kb_seg_sum(const KaiCtx& c, RoundParams rp, int segs) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    KW_SHARED PlanScanLds L; KW_SHARED int32_t s_sum; KW_SHARED int32_t s_inc;
    constexpr int E = KPS_E;
    SegNode n; if (!kps_node(c, rp.height, segs, s_sum, s_inc, n, true)) return;
    const int tid = kw::tid();
    if (n.s == 0 && tid == 0) b.sg_fb[n.x] = 0x7fffffff;
    if (n.s * KPS_SEG >= n.V0) return;  // no job in this segment
    const int t0 = n.s * KPS_SEG + tid * E;
    double tsum[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < E; j++) {
        const int t = t0 + j; if (t >= n.V0) break;
        const int meta = b.d_meta[n.eb + t]; const bool ok = (meta & 3) == BF_OK, np = (meta >> 2) & 1;
        if (ok) for (int k = 0; k < 3; k++) { const double r = b.d_res[(size_t)(n.eb + t) * 3 + k]; tsum[k] += r; if (np) tsum[3 + k] += r; }
    }
    double incl[6], tot[6];
    kb_block_scan_add<6>(L, tsum, incl, tot);
    if (tid == 0) for (int k = 0; k < 6; k++) b.sg_tot[(size_t)(n.slot0 + n.s) * 6 + k] = tot[k];
}

KW_BODY void kb_seg_gate(const KaiCtx& c, RoundParams rp, int segs) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    KW_SHARED PlanScanLds L; KW_SHARED int32_t s_sum; KW_SHARED int32_t s_inc;
    constexpr int E = KPS_E;
    SegNode n; if (!kps_node(c, rp.height, segs, s_sum, s_inc, n, true)) return;
    if (n.s * KPS_SEG > n.V0) return;  // (position V0, behind the last job, still gets its shares: the key the node competes with after its last pop)
    const int tid = kw::tid(), lane = kw::lane();
    double alloc[3], anp[3];
    for (int k = 0; k < 3; k++) { alloc[k] = c.q_share[(size_t)n.x * 3 + k].allocated; anp[k] = c.q_share[(size_t)n.x * 3 + k].allocated_np; }
    for (int sp = 0; sp < n.s; sp++) for (int k = 0; k < 3; k++) { alloc[k] += b.sg_tot[(size_t)(n.slot0 + sp) * 6 + k]; anp[k] += b.sg_tot[(size_t)(n.slot0 + sp) * 6 + 3 + k]; }
    const PlanNodeConst nc = plan_node_const(c, n.x, c.st->total[0], c.st->total[1], c.st->total[2]);
    const int t0 = n.s * KPS_SEG + tid * E;
    int meta[E]; double res[E][3], pre[E][6], tsum[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < E; j++) {
        const int t = t0 + j; const bool is_elem = t < n.V0;
        meta[j] = is_elem ? (int)b.d_meta[n.eb + t] : (int)BF_GATE;
        for (int k = 0; k < 3; k++) res[j][k] = is_elem ? b.d_res[(size_t)(n.eb + t) * 3 + k] : 0.0;
        const bool ok = (meta[j] & 3) == BF_OK, np = (meta[j] >> 2) & 1;
        for (int k = 0; k < 6; k++) pre[j][k] = tsum[k];
        for (int k = 0; k < 3; k++) { tsum[k] += ok ? res[j][k] : 0.0; tsum[3 + k] += (ok && np) ? res[j][k] : 0.0; }
    }
    double incl[6], tot[6];
    kb_block_scan_add<6>(L, tsum, incl, tot);
    int fb = 0x7fffffff;  // this thread's first OK job the node turns away
    for (int j = 0; j < E; j++) {
        const int t = t0 + j, flag = meta[j] & 3; const bool np = (meta[j] >> 2) & 1;
        if (t > n.V0) break;
        double ab[3], abn[3];
        for (int k = 0; k < 3; k++) { ab[k] = alloc[k] + (incl[k] - tsum[k]) + pre[j][k]; abn[k] = anp[k] + (incl[3 + k] - tsum[3 + k]) + pre[j][3 + k]; }
        for (int k = 0; k < 3; k++) b.d_ab[(size_t)(n.kb + t) * 3 + k] = ab[k];
        const bool gate = t < n.V0 && flag != BF_GATE && plan_gate_fails_c(nc, ab, abn, res[j], np);
        if (gate) { b.d_meta[n.eb + t] = (uint8_t)(meta[j] | 8); if (flag == BF_OK && fb == 0x7fffffff) fb = t; }
    }
    const uint64_t wm = kw::wave_max_u64(fb == 0x7fffffff ? 0ull : (uint64_t)(0x7fffffff - fb));  // the wave's smallest fb
    if (lane == 0 && wm) kw::atomic_min((int32_t*)&b.sg_fb[n.x], 0x7fffffff - (int)wm);
}

KW_BODY void kb_seg_keys(const KaiCtx& c, RoundParams rp, int segs) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    KW_SHARED PlanScanLds L; KW_SHARED int32_t s_sum; KW_SHARED int32_t s_inc;
    constexpr int E = KPS_E;
    SegNode n; if (!kps_node(c, rp.height, segs, s_sum, s_inc, n, true)) return;
    const int tid = kw::tid();
    const int f = b.sg_fb[n.x];
    const int V = f == 0x7fffffff ? n.V0 : f + 1;  // the first OK job the node turns away ends its valid stream: everything behind it would be ordered under wrong shares
    const bool complete = n.compl_all && V == n.sumV;
    const int nk = complete ? V : V + 1;
    if (n.s == 0 && tid == 0) { b.q_valid[n.x] = V; b.q_nk[n.x] = nk; b.q_complete[n.x] = complete ? 1 : 0; }
#if !defined(__HIPCC__)
    if (n.s == 0 && tid == 0 && std::getenv("KAI_PLAN_SEG_DEBUG")) std::fprintf(stderr, "seg node %d: V0 %d first turned away %d (segment %d) -> V %d keys %d\n", n.x, n.V0, f == 0x7fffffff ? -1 : f, f == 0x7fffffff ? -1 : f / KPS_SEG, V, nk);
#endif
    if (n.s * KPS_SEG >= nk) return;
    const PlanNodeConst nc = plan_node_const(c, n.x, c.st->total[0], c.st->total[1], c.st->total[2]);
    const int srank = b.q_srank[n.x];
    const int t0 = n.s * KPS_SEG + tid * E;
    PlanKey key[E]; bool kv[E];
    PlanKey tm; tm.w0 = tm.w1 = tm.w2 = tm.w3 = 0; bool tmv = false;
    for (int j = 0; j < E; j++) {
        const int t = t0 + j; const bool is_key = t < nk;
        kv[j] = is_key; key[j].w0 = key[j].w1 = key[j].w2 = key[j].w3 = 0;
        if (t < V && (b.d_meta[n.eb + t] & 8)) b.e_flag[b.el_leaf[n.eb + t]] = BF_GATE;  // turned away under exact shares (a dead job turned away here counts as a gate failure)
        if (is_key) {
            double ab[3], rq[3];
            for (int k = 0; k < 3; k++) { ab[k] = b.d_ab[(size_t)(n.kb + t) * 3 + k]; rq[k] = b.d_spres[(size_t)(n.kb + t) * 3 + k]; }
            if (f != 0x7fffffff && t == f + 1) for (int k = 0; k < 3; k++) ab[k] -= b.d_res[(size_t)(n.eb + f) * 3 + k];  // job f was summed as placed; the node turned it away
            key[j] = plan_key_c(nc, ab, rq, srank);
            if (tmv && pk_less(key[j], tm)) key[j] = tm;  // running maximum inside the thread
            tm = key[j]; tmv = true;
        }
    }
    PlanKey none; none.w0 = none.w1 = none.w2 = none.w3 = 0;
    PlanKey pm, last; bool have_pm, have_last;
    kb_block_excl_max(L, tm, tmv, none, false, pm, have_pm, last, have_last);
    for (int j = 0; j < E; j++) if (kv[j]) {
        const int t = t0 + j;
        PlanKey out = key[j]; if (have_pm && pk_less(out, pm)) out = pm;
        b.pk[n.kb + t] = out; b.sp[n.kb + t] = b.d_spj[n.kb + t]; b.k_owner[n.kb + t] = n.x;
    }
    if (tid == 0) { b.sg_key[n.slot0 + n.s] = last; b.sg_kvalid[n.slot0 + n.s] = have_last ? 1 : 0; }
}

KW_BODY void kb_seg_max(const KaiCtx& c, RoundParams rp, int segs) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    KW_SHARED int32_t s_sum; KW_SHARED int32_t s_inc;
    constexpr int E = KPS_E;
    SegNode n; if (!kps_node(c, rp.height, segs, s_sum, s_inc, n, false)) return;
    const int nk = b.q_nk[n.x];
    if (n.s == 0 || n.s * KPS_SEG >= nk) return;
    PlanKey carry; carry.w0 = carry.w1 = carry.w2 = carry.w3 = 0; bool have = false;
    for (int sp = 0; sp < n.s; sp++) if (b.sg_kvalid[n.slot0 + sp]) { const PlanKey o = b.sg_key[n.slot0 + sp]; if (!have || pk_less(carry, o)) { carry = o; have = true; } }
    if (!have) return;
    const int t0 = n.s * KPS_SEG + kw::tid() * E;
    for (int j = 0; j < E; j++) { const int t = t0 + j; if (t >= nk) break; const PlanKey k = b.pk[n.kb + t]; if (pk_less(k, carry)) b.pk[n.kb + t] = carry; }
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(KPS_T) k_seg_sum(KaiCtx c, RoundParams rp, int segs) { kb_seg_sum(c, rp, segs); }
__global__ void __launch_bounds__(KPS_T) k_seg_gate(KaiCtx c, RoundParams rp, int segs) { kb_seg_gate(c, rp, segs); }
__global__ void __launch_bounds__(KPS_T) k_seg_keys(KaiCtx c, RoundParams rp, int segs) { kb_seg_keys(c, rp, segs); }
__global__ void __launch_bounds__(KPS_T) k_seg_max(KaiCtx c, RoundParams rp, int segs) { kb_seg_max(c, rp, segs); }
#endif

}  // namespace kai
