// kai_batch_kernels.hpp — kernel bodies of the batch path (see kai_batch.hpp for the algorithm), written against kai_simt.hpp.
// Under hipcc every body gets a __global__ entry point; tests/host_sim runs the same bodies on the lock-step emulator.
#pragma once
#include "kai_batch.hpp"

namespace kai {

constexpr int KB_INF = 0x7fffffff;
#ifdef KAI_FILL_PROF  // section clocks of the fill wave (diagnostic build only: every read of the clock stalls the wave)
#define KB_T(var) const int64_t var = kw::clock()
#define KB_ACC(slot, t0) f.cy[slot] += kw::clock() - (t0)
#else
#define KB_T(var)
#define KB_ACC(slot, t0)
#endif
constexpr int KB_APPLY_INNER = 1024;  // inner queue nodes whose shares k_apply_jobs sums per workgroup in LDS (48 KB); the nodes beyond take the atomics directly
#if defined(__HIPCC__)
constexpr int KB_PLAN_SCAN_THREADS = 1024; // workgroup of k_plan_scan (a multiple of 64, at most 1024): the scan is latency bound (f64 divisions of the keys), more wavefronts hide more of it
constexpr int KB_PLAN_SCAN_ELEMS = 4;      // positions of a stream per thread and step
constexpr int KB_PLAN_SETUP_THREADS = 1024; // the one workgroup of k_plan_setup (passes over the Q + 1 queue nodes: config 5's 2 185 in 3 steps instead of 9)
#else
constexpr int KB_PLAN_SETUP_THREADS = 128;
constexpr int KB_PLAN_SCAN_THREADS = 128;  // the emulator runs every thread as a fiber: two waves exercise the same code
constexpr int KB_PLAN_SCAN_ELEMS = 3;      // (an odd count: positions, threads and waves fall out of step)
#endif
constexpr int KB_PLACED_MAX = 1024;  // tasks of one gang the fill kernel can roll back (larger chunks do not qualify)

KW_BODY bool kb_is_leaf(const KaiCtx& c, int q) { return q < c.Q && c.q_child_off[q + 1] == c.q_child_off[q]; }
KW_BODY int kb_parent(const KaiCtx& c, int q) { int p = c.q_parent[q]; return p < 0 ? c.Q : p; }
// rounds without the host (RoundCtl): a round that was enqueued ahead of the loop's end has nothing to do — every kernel of a round starts with this
KW_BODY bool kb_round_off(const BatchCtx& b) { return b.dev_loop && b.ctl->done; }

// ------------------------------------------------------------------------------------------------------ per action
// static sibling order (queue_order.go:214-240): rank = number of siblings that sort before q; flagged when the relation is not a strict
// total order on the sibling set (a tournament is transitive iff its scores are all different)
KW_BODY void kb_static_rank(const KaiCtx& c) {
    const int q = kw::bid() * kw::bdim() + kw::tid();
    if (q >= c.Q) return;
    const BatchCtx& b = c.bt;
    const int par = kb_parent(c, q), b0 = c.q_child_off[par], b1 = c.q_child_off[par + 1];
    int wins = 0; bool bad = false;
    for (int i = b0; i < b1; i++) { int s = c.q_children[i]; if (s == q) continue; bool sq = plan_static_before(c, s, q), qs = plan_static_before(c, q, s); if (sq == qs) bad = true; if (sq) wins++; }
    b.q_srank[q] = wins;
    if (bad) kw::atomic_add((int32_t*)&b.qual[1], 1);
}
KW_BODY void kb_static_check(const KaiCtx& c) {
    const int q = kw::bid() * kw::bdim() + kw::tid();
    if (q >= c.Q) return;
    const BatchCtx& b = c.bt;
    const int par = kb_parent(c, q), b0 = c.q_child_off[par], b1 = c.q_child_off[par + 1], mine = b.q_srank[q];
    for (int i = b0; i < b1; i++) { int s = c.q_children[i]; if (s != q && b.q_srank[s] == mine) { kw::atomic_add((int32_t*)&b.qual[1], 1); break; } }
    b.cur_sp[q] = -1;
}
// per queued job: is it "regular" (see kai_batch.hpp)?  + the scan classes of its chunk, + class of every task in chunk order
KW_BODY void kb_qualify(const KaiCtx& c) {
    const int j = kw::bid() * kw::bdim() + kw::tid();
    if (j >= c.J) return;
    const BatchCtx& b = c.bt;
    b.j_clsmask[j] = 0; b.j_ucls[j] = -1;
    const int st = c.j_state[j];
    if (st == 3) return;  // not queued
    kw::atomic_add((int32_t*)&b.qual[2], 1);
    bool ok = st == 0 && c.j_n_ps[j] == 1 && !c.j_has_topology[j] && c.j_tta_valid[j] && c.j_tta_n[j] == c.j_n_pending[j] && c.j_tta_n[j] >= 1 && c.j_tta_n[j] <= KB_PLACED_MAX;
    if (ok && c.s_pipelined[c.j_first_ps[j]] != 0) ok = false;
    uint64_t mask = 0;
    if (ok) {
        const int first = c.j_first_pod[j], nt = c.j_tta_n[j]; const bool nominated = c.plugins & KAI_PLUGIN_NOMINATEDNODE;
        for (int i = 0; i < nt; i++) {
            const int p = c.tta[first + i], k = c.p_scls[p];
            if (k < 0 || c.p_status[p] != KAI_POD_PENDING || c.p_on_node[p] >= 0 || (nominated && c.p_nominated[p] >= 0)) { ok = false; break; }
            mask |= 1ull << k; b.t_cls[first + i] = k;
        }
    }
    if (!ok) { kw::atomic_add((int32_t*)&b.qual[0], 1); return; }
    kw::atomic_max((int32_t*)&b.qual[3], c.j_tta_n[j]);  // the largest gang (a node-sharded group compares it with the offers per class)
    b.j_clsmask[j] = mask; b.j_ucls[j] = (mask & (mask - 1)) == 0 ? __builtin_ctzll(mask) : -1;
}
// node records of the fill kernel from the session's node arrays (nothing is releasing on this path: the host checked)
KW_BODY void kb_build_nrec(const KaiCtx& c) {
    const int n = kw::bid() * kw::bdim() + kw::tid();
    if (n >= c.NB * KAI_BLOCK) return;
    c.bt.nrec[n] = make_node_rec(c, n);
}

// ------------------------------------------------------------------------------------------------------ plan: setup
// candidate counts (a leaf offers its next h_leaf jobs), stream regions of every node in the pools.  One workgroup.
KW_BODY void kb_plan_setup(const KaiCtx& c, RoundParams rp) {
    if (kb_round_off(c.bt)) return;
    if (c.bt.dev_loop) rp.h_leaf = c.bt.ctl->H;  // how far this plan looks: the loop's state lives on the device
    const BatchCtx& b = c.bt;
    const int T = kw::bdim(), t = kw::tid(), Q = c.Q;
    for (int q = t; q <= Q; q += T) {
        int cnt = 0;
        if (kb_is_leaf(c, q)) { int rem = c.lq_end[q] - c.lq_cur[q]; cnt = rem < rp.h_leaf ? rem : rp.h_leaf; if (cnt < 0) cnt = 0; }
        b.q_cnt[q] = cnt; b.q_sent[q] = KB_INF; b.q_valid[q] = 0; b.q_nk[q] = 0; b.q_complete[q] = 1; b.q_taken[q] = 0;
    }
    kw::sync();
    for (int h = 1; h < b.n_h; h++) {
        for (int i = b.h_off[h] + t; i < b.h_off[h + 1]; i += T) {
            const int x = b.h_nodes[i]; int cnt = 0;
            for (int k = c.q_child_off[x]; k < c.q_child_off[x + 1]; k++) cnt += b.q_cnt[c.q_children[k]];
            b.q_cnt[x] = cnt;
        }
        kw::sync();
    }
    // exclusive scan over h_nodes order: element region = cnt + children (room for the sentinels), key region = cnt + 1
    KW_SHARED int s_part[64]; KW_SHARED int s_carry[2];
    if (t == 0) { s_carry[0] = 0; s_carry[1] = 0; }
    kw::sync();
    const int lane = kw::lane(), wave = t >> 6, nw = (T + 63) >> 6;
    for (int base = 0; base <= Q; base += T) {
        const int i = base + t; int ve = 0, vk = 0, x = -1;
        if (i <= Q) { x = b.h_nodes[i]; ve = b.q_cnt[x] + (c.q_child_off[x + 1] - c.q_child_off[x]); vk = b.q_cnt[x] + 1; }
        int se = kw::wave_scan_add(ve), sk = kw::wave_scan_add(vk);
        if (lane == 63) { s_part[wave] = se; s_part[32 + wave] = sk; }
        kw::sync();
        int oe = s_carry[0], ok = s_carry[1];
        for (int w = 0; w < wave; w++) { oe += s_part[w]; ok += s_part[32 + w]; }
        if (x >= 0) { b.q_ebase[x] = oe + se - ve; b.q_kbase[x] = ok + sk - vk; }
        kw::sync();
        if (t == T - 1) { int te = 0, tk = 0; for (int w = 0; w < nw; w++) { te += s_part[w]; tk += s_part[32 + w]; } s_carry[0] += te; s_carry[1] += tk; }
        kw::sync();
    }
    if (t == 0) { b.plan_tot[0] = s_carry[0]; b.plan_tot[1] = s_carry[1]; }
}

// inclusive running maximum of a PlanKey over the lanes, seeded with `carry` (valid when have_carry)
KW_BODY PlanKey kb_wave_scan_max(PlanKey v, bool valid, PlanKey carry, bool have_carry) {
    if (have_carry && (!valid || pk_less(v, carry))) { v = carry; valid = true; }
#if defined(__HIPCC__)
    // the DPP scan of kai_wave.hpp (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31) over (valid, key): nine DPP moves per step instead of nine ds_bpermute round trips
#define KB_MAX_STEP(CTRL, RM) { PlanKey o; o.w0 = kw::dpp_mov64<CTRL, RM>(v.w0); o.w1 = kw::dpp_mov64<CTRL, RM>(v.w1); o.w2 = kw::dpp_mov64<CTRL, RM>(v.w2); o.w3 = kw::dpp_mov64<CTRL, RM>(v.w3); \
                                const int ov = kw::dpp_mov32<CTRL, RM>((int)valid); if (ov && (!valid || pk_less(v, o))) { v = o; valid = true; } }
    KB_MAX_STEP(0x111, 0xf) KB_MAX_STEP(0x112, 0xf) KB_MAX_STEP(0x114, 0xf) KB_MAX_STEP(0x118, 0xf) KB_MAX_STEP(0x142, 0xa) KB_MAX_STEP(0x143, 0xc)
#undef KB_MAX_STEP
#else
    const int lane = kw::lane();
    for (int d = 1; d < 64; d <<= 1) {
        PlanKey o; o.w0 = kw::shfl_up(v.w0, d); o.w1 = kw::shfl_up(v.w1, d); o.w2 = kw::shfl_up(v.w2, d); o.w3 = kw::shfl_up(v.w3, d);
        const int ov = kw::shfl_up((int)valid, d);
        if (lane >= d && ov && (!valid || pk_less(v, o))) { v = o; valid = true; }
    }
#endif
    return v;
}

// ------------------------------------------------------------------------------------------------------ plan: leaves
// One wavefront per leaf queue: walks the leaf's next candidates in JobOrderFn order, decides the leaf's own capacity gate exactly
// (sequential in effect: a gate failure changes the shares every later job sees — resolved per 64-job chunk by re-scanning from the first
// failure), predicts node fit from the dead classes, and emits the key the leaf competes with before each pop.
KW_BODY void kb_plan_leaf(const KaiCtx& c, RoundParams rp) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    const int q = kw::bid() * (kw::bdim() >> 6) + (kw::tid() >> 6), lane = kw::lane();
    if (q >= c.Q || !kb_is_leaf(c, q)) return;
    const int V = b.q_cnt[q], rem = c.lq_end[q] - c.lq_cur[q];
    const bool complete = V == rem;
    const int nk = V + (complete ? 0 : 1);
    const int off = c.q_job_off[q] + c.lq_cur[q], eb = b.q_ebase[q], kb = b.q_kbase[q], srank = b.q_srank[q];
    double alloc[3], anp[3];
    for (int k = 0; k < 3; k++) { alloc[k] = c.q_share[(size_t)q * 3 + k].allocated; anp[k] = c.q_share[(size_t)q * 3 + k].allocated_np; }
    const uint64_t dead_mask = b.dead_mask[0];
    const PlanNodeConst nc = plan_node_const(c, q, c.st->total[0], c.st->total[1], c.st->total[2]);  // what the gate and the keys read of the leaf, loaded once
    PlanKey run; run.w0 = run.w1 = run.w2 = run.w3 = 0; bool have_run = false;
    for (int base = 0; base < nk; base += 64) {
        const int i = base + lane;
        const bool is_elem = i < V, is_key = i < nk;
        const int job = is_key ? c.lq_sorted[off + i] : -1;
        double res[3] = {0, 0, 0}; bool np = false, dead = false;
        if (job >= 0) { for (int k = 0; k < 3; k++) res[k] = c.j_tta_res[(size_t)job * 4 + k]; np = !c.j_preempt[job]; dead = (b.j_clsmask[job] & dead_mask) != 0;
                        const int uc = b.j_ucls[job]; if (uc >= 0 && b.cls_cap[uc] < c.j_tta_n[job]) dead = true; }  // a gang of one class larger than what the cluster holds of it
        const bool assumed = is_elem && !dead; bool gate = false;
        double ab[3], abn[3], tot[3], totn[3];
        for (int k = 0; k < 3; k++) {  // every job assumed placed unless its classes are dead: exact up to the first job the leaf's own gate turns away
            const double d = assumed ? res[k] : 0.0, dn = (assumed && np) ? res[k] : 0.0;
            const double s = kw::wave_scan_add(d), sn = kw::wave_scan_add(dn);
            ab[k] = alloc[k] + (s - d); abn[k] = anp[k] + (sn - dn);
            tot[k] = kw::shfl(s, 63); totn[k] = kw::shfl(sn, 63);
        }
        gate = is_elem && plan_gate_fails_c(nc, ab, abn, res, np);
        const uint64_t bad = kw::ballot(assumed && gate);
        if (bad) {
            // A gate failure changes the shares every later job of the chunk sees.  From the first one on the chunk is settled job by job on uniform values (every lane computes the same
            // running sums; lane l keeps what job l met): ~40 instructions per job, where re-scanning the chunk once per failure — what this loop did until round 6 — cost six wave scans
            // per failure, and a leaf at its limit fails on every job (config 5: 70 k of 307 k planned jobs are gated, in the leaves with limits; k_plan_leaf was 0.3 ms of a 1.9 ms plan)
            const int f = __builtin_ctzll(bad);
            double ra[3], rn[3];
            for (int k = 0; k < 3; k++) { ra[k] = kw::bcast(ab[k], f); rn[k] = kw::bcast(abn[k], f); }
            const uint64_t em = kw::ballot(is_elem), dm = kw::ballot(dead), npm = kw::ballot(np);
            for (int l = f; l < 64 && ((em >> l) & 1ull); l++) {
                double r[3]; for (int k = 0; k < 3; k++) r[k] = kw::bcast(res[k], l);
                const bool npl = (npm >> l) & 1ull, dl = (dm >> l) & 1ull;
                const bool g = plan_gate_fails_c(nc, ra, rn, r, npl);
                if (lane == l) { for (int k = 0; k < 3; k++) { ab[k] = ra[k]; abn[k] = rn[k]; } gate = g; }
                if (!dl && !g) for (int k = 0; k < 3; k++) { ra[k] += r[k]; if (npl) rn[k] += r[k]; }
            }
            if (!is_elem) for (int k = 0; k < 3; k++) { ab[k] = ra[k]; abn[k] = rn[k]; }  // (the key behind the chunk's last job)
            for (int k = 0; k < 3; k++) { tot[k] = ra[k] - alloc[k]; totn[k] = rn[k] - anp[k]; }  // (exact: HostPrep::batch_units)
        }
        if (is_elem) { b.e_job[eb + i] = job; b.e_flag[eb + i] = gate ? BF_GATE : dead ? BF_DEAD : BF_OK; b.e_grank[eb + i] = KB_INF; }
        PlanKey key; key.w0 = key.w1 = key.w2 = key.w3 = 0;
        if (is_key) key = plan_key_c(nc, ab, res, srank);
        key = kb_wave_scan_max(key, is_key, run, have_run);
        if (is_key) { b.pk[kb + i] = key; b.sp[kb + i] = job; b.k_owner[kb + i] = q; }
        run.w0 = kw::shfl(key.w0, 63); run.w1 = kw::shfl(key.w1, 63); run.w2 = kw::shfl(key.w2, 63); run.w3 = kw::shfl(key.w3, 63); have_run = true;
        for (int k = 0; k < 3; k++) { alloc[k] += tot[k]; anp[k] += totn[k]; }
    }
    if (lane == 0) { b.q_valid[q] = V; b.q_nk[q] = nk; b.q_complete[q] = complete ? 1 : 0; }
}

// ------------------------------------------------------------------------------------------------------ plan: merge
// One thread per key slot of a child whose parent has height rp.height: its rank in the parent's merged stream = its index in its own stream
// + the number of smaller running-maximum keys in every sibling stream (keys of different siblings never tie: w3 is the sibling rank).
KW_BODY void kb_plan_rank(const KaiCtx& c, RoundParams rp) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    const int g = kw::bid() * kw::bdim() + kw::tid();
    if (g >= rp.n_slots || g >= b.plan_tot[1]) return;  // key slots in use this round (the host's upper bound / what k_plan_setup laid out)
    const int cq = b.k_owner[g];
    if (cq < 0 || cq >= c.Q) return;
    const int i = g - b.q_kbase[cq];
    if (i < 0 || i >= b.q_nk[cq]) return;
    const int x = kb_parent(c, cq);
    if (b.q_height[x] != rp.height) return;
    const PlanKey key = b.pk[g];
    int rank = i;
    // the binary searches in the siblings' streams, eight siblings in lock-step: a step's eight loads are independent and in flight together (one search after the other is a chain of
    // ~12 dependent L2 round trips per sibling: r06l, 0.23 ms of a full plan and 0.1 ms of every short one)
    const int c0 = c.q_child_off[x], c1 = c.q_child_off[x + 1];
    if (c1 - c0 <= 3) {  // a narrow tree: one or two siblings, searched one after the other (eight loads per step would be seven too many)
        for (int k = c0; k < c1; k++) { const int s2 = c.q_children[k]; if (s2 == cq) continue; const int n = b.q_nk[s2]; if (n) rank += plan_lower_bound(b.pk + b.q_kbase[s2], n, key); }
    } else
    for (int k0 = c0; k0 < c1; k0 += 8) {
        int lo[8], hi[8], kbs[8];
        for (int j = 0; j < 8; j++) {
            const int k = k0 + j < c1 ? k0 + j : c1 - 1; const int s2 = c.q_children[k];
            lo[j] = 0; hi[j] = (k0 + j < c1 && s2 != cq) ? b.q_nk[s2] : 0; kbs[j] = b.q_kbase[s2];
        }
        for (bool more = true; more;) {
            more = false;
            PlanKey m[8]; int mid[8];
            for (int j = 0; j < 8; j++) { mid[j] = (lo[j] + hi[j]) >> 1; m[j] = b.pk[kbs[j] + (lo[j] < hi[j] ? mid[j] : 0)]; }  // (unconditional loads: a finished search re-reads its stream's first key)
            for (int j = 0; j < 8; j++) { const bool act = lo[j] < hi[j], less = pk_less(m[j], key); lo[j] = (act && less) ? mid[j] + 1 : lo[j]; hi[j] = (act && !less) ? mid[j] : hi[j]; more = more || lo[j] < hi[j]; }
        }
        for (int j = 0; j < 8; j++) rank += lo[j];
    }
    const int V = b.q_valid[cq], cap = b.q_cnt[x] + (c.q_child_off[x + 1] - c.q_child_off[x]);
    if (rank >= cap) return;  // beyond a sentinel: never used
    const int pos = b.q_ebase[x] + rank;
    if (i < V) {
        b.el_leaf[pos] = kb_is_leaf(c, cq) ? b.q_ebase[cq] + i : b.el_leaf[b.q_ebase[cq] + i];
        b.el_ck[pos] = g + 1; b.el_next[pos] = (i + 1 < b.q_nk[cq]) ? 1 : 0;
    } else {  // the key after the child's last planned element: the parent's stream is valid up to here
        b.el_leaf[pos] = -1; b.el_ck[pos] = g + 1; b.el_next[pos] = 0;
        kw::atomic_min((int32_t*)&b.q_sent[x], rank);
    }
}

// One thread per position of the merged streams of the nodes of height rp.height: what the node's scan reads of the position — the element's job resources and flag, and the
// resources of the stale-path job the node's key before that pop is read through — gathered here chip-wide, so that the scan (ONE workgroup per node, walking its stream chunk by
// chunk) reads coalesced arrays instead of waiting out three dependent loads per element and chunk (r05u: 1.4 of the 1.9 ms of a 300 k-element plan were k_plan_scan).
KW_BODY void kb_plan_gather(const KaiCtx& c, RoundParams rp) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    const int p = kw::bid() * kw::bdim() + kw::tid();
    if (p >= b.plan_tot[0]) return;
    int lo = b.h_off[rp.height], hi = b.h_off[rp.height + 1];  // nodes of this height, their regions ascending in the pools (k_plan_setup lays them out in h_nodes order)
    if (lo >= hi || p < b.q_ebase[b.h_nodes[lo]]) return;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (b.q_ebase[b.h_nodes[mid]] <= p) lo = mid; else hi = mid; }
    const int x = b.h_nodes[lo], eb = b.q_ebase[x], kb = b.q_kbase[x], t = p - eb;
    if (x == c.Q) return;
    int tot = 0;  // keys the children's streams hold = positions of the merged stream the rank kernel has written
    for (int k = c.q_child_off[x]; k < c.q_child_off[x + 1]; k++) tot += b.q_nk[c.q_children[k]];
    const int cap = b.q_cnt[x] + (c.q_child_off[x + 1] - c.q_child_off[x]);
    if (t >= tot || t >= cap) return;
    const int e = b.el_leaf[p], job = e >= 0 ? b.e_job[e] : -1; const int flag = e >= 0 ? (int)b.e_flag[e] : (int)BF_GATE;
    double res[3] = {0, 0, 0}; int np = 0;
    if (job >= 0) { for (int k = 0; k < 3; k++) res[k] = c.j_tta_res[(size_t)job * 4 + k]; np = c.j_preempt[job] ? 0 : 1; }
    for (int k = 0; k < 3; k++) b.d_res[(size_t)p * 3 + k] = res[k];
    b.d_meta[p] = (uint8_t)(flag | (np << 2));
    if (t > b.q_cnt[x]) return;  // the node's key region holds q_cnt + 1 slots (its element region one per child more): no key beyond
    int spj;  // (kb_plan_scan pass 2 explains the three cases)
    if (t == 0) { const int cs = b.cur_sp[x]; spj = cs >= 0 ? cs : b.sp[b.el_ck[eb] - 1]; }
    else if (b.el_next[p - 1]) spj = b.sp[b.el_ck[p - 1]];
    else spj = b.sp[b.el_ck[p] - 1];
    b.d_spj[kb + t] = spj;
#if !defined(__HIPCC__) && defined(KAI_PLAN_DEBUG)
    if (std::getenv("KAI_PLAN_DEBUG_X") && x == std::atoi(std::getenv("KAI_PLAN_DEBUG_X"))) std::fprintf(stderr, "gather h %d x %d p %d eb %d kb %d t %d tot %d cap %d: el_ck[p] %d el_next[p-1] %d -> spj %d\n", rp.height, x, p, eb, kb, t, tot, cap, b.el_ck[p], t ? (int)b.el_next[p - 1] : -1, spj);
#endif
    for (int k = 0; k < 3; k++) b.d_spres[(size_t)(kb + t) * 3 + k] = spj >= 0 ? c.j_tta_res[(size_t)spj * 4 + k] : 0.0;
}

// One wavefront per inner node of height rp.height (the virtual root included): shares along its merged stream (segmented scan), its own
// capacity gate (the first job it turns away ends the node's valid stream: everything after it would be ordered under wrong shares), the
// stale-path job and the key of the node before each of its pops, as a running maximum.
// Workgroup-wide scans of the stream of one queue node: thread t of the workgroup holds element base + t.  Per-wave scans, the waves' totals (or last
// keys) through LDS, one barrier pair per batch of scans.
struct PlanScanLds { double tot[6][16]; PlanKey wkey[16]; int32_t wvalid[16]; int32_t wbad[16]; };
template <int NV>
KW_BODY void kb_block_scan_add(PlanScanLds& L, const double* d, double* incl, double* total) {  // NV independent inclusive sums over the workgroup's threads
    const int w = kw::tid() >> 6, nw = kw::bdim() >> 6, lane = kw::lane();
    for (int k = 0; k < NV; k++) { incl[k] = kw::wave_scan_add(d[k]); if (lane == 63) L.tot[k][w] = incl[k]; }
    kw::sync();
    for (int k = 0; k < NV; k++) { double pre = 0, tot = 0; for (int i = 0; i < nw; i++) { const double v = L.tot[k][i]; if (i < w) pre += v; tot += v; } incl[k] += pre; total[k] = tot; }
    kw::sync();  // the totals are read: the next batch may overwrite them
}
KW_BODY PlanKey kb_block_scan_max(PlanScanLds& L, PlanKey v, bool valid, PlanKey carry, bool have_carry, PlanKey& last) {  // inclusive running maximum, seeded with carry; last = its value at the last thread
    const int w = kw::tid() >> 6, nw = kw::bdim() >> 6, lane = kw::lane();
    PlanKey none; none.w0 = none.w1 = none.w2 = none.w3 = 0;
    v = kb_wave_scan_max(v, valid, none, false);
    const uint64_t anyv = kw::ballot(valid);
    const bool mine = anyv && lane >= __builtin_ctzll(anyv ? anyv : 1);  // lanes from the wave's first valid element on hold a maximum
    if (lane == 63) { L.wkey[w] = v; L.wvalid[w] = anyv ? 1 : 0; }
    kw::sync();
    PlanKey pre = carry; bool have = have_carry;
    for (int i = 0; i < w; i++) if (L.wvalid[i]) { const PlanKey o = L.wkey[i]; if (!have || pk_less(pre, o)) { pre = o; have = true; } }
    PlanKey out = v; bool ov = mine;
    if (have && (!ov || pk_less(out, pre))) { out = pre; ov = true; }
    PlanKey fin = carry; bool hf = have_carry;
    for (int i = 0; i < nw; i++) if (L.wvalid[i]) { const PlanKey o = L.wkey[i]; if (!hf || pk_less(fin, o)) { fin = o; hf = true; } }
    last = fin;
    kw::sync();
    return out;
}
// EXCLUSIVE running maximum over the workgroup's threads, seeded with carry: pre = the maximum of carry and the values of the threads before this one (have_pre: there is one);
// last = the maximum over carry and every thread's value (valid when any of them is)
KW_BODY void kb_block_excl_max(PlanScanLds& L, PlanKey v, bool valid, PlanKey carry, bool have_carry, PlanKey& pre, bool& have_pre, PlanKey& last, bool& have_last) {
    const int w = kw::tid() >> 6, nw = kw::bdim() >> 6, lane = kw::lane();
    PlanKey none; none.w0 = none.w1 = none.w2 = none.w3 = 0;
    const PlanKey inc = kb_wave_scan_max(v, valid, none, false);  // inclusive, meaningful from the wave's first valid lane on
    const uint64_t anyv = kw::ballot(valid);
    if (lane == 63) { L.wkey[w] = inc; L.wvalid[w] = anyv ? 1 : 0; }
    PlanKey ex; ex.w0 = kw::shfl_up(inc.w0, 1); ex.w1 = kw::shfl_up(inc.w1, 1); ex.w2 = kw::shfl_up(inc.w2, 1); ex.w3 = kw::shfl_up(inc.w3, 1);
    const bool ex_valid = (anyv & ((1ull << lane) - 1)) != 0;  // a valid lane in front of this one in its wave
    kw::sync();
    pre = carry; have_pre = have_carry;
    for (int i = 0; i < w; i++) if (L.wvalid[i]) { const PlanKey o = L.wkey[i]; if (!have_pre || pk_less(pre, o)) { pre = o; have_pre = true; } }
    last = pre; have_last = have_pre;
    for (int i = w; i < nw; i++) if (L.wvalid[i]) { const PlanKey o = L.wkey[i]; if (!have_last || pk_less(last, o)) { last = o; have_last = true; } }
    if (ex_valid && (!have_pre || pk_less(pre, ex))) { pre = ex; have_pre = true; }
    kw::sync();
}
// One WORKGROUP per queue node of height rp.height (the streams of the top levels hold tens of thousands of elements), KB_PLAN_SCAN_ELEMS consecutive positions of the stream per
// thread and step: a thread sums its own positions serially and the workgroup scans the threads' totals, so a step of T·E positions costs the shuffles and barriers of ONE scan over T
// values (r06c: the 38 k-position streams of config 5's eight top-level queues took 1.03 ms of a 1.9 ms plan at one position per thread).  Sums in another order are exact on this
// path (HostPrep::batch_units).
KW_BODY void kb_plan_scan(const KaiCtx& c, RoundParams rp) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    KW_SHARED PlanScanLds L; KW_SHARED int32_t s_sum; KW_SHARED int32_t s_incomplete;
    constexpr int E = KB_PLAN_SCAN_ELEMS;
    const int idx = b.h_off[rp.height] + kw::bid(), lane = kw::lane(), tid = kw::tid(), T = kw::bdim(), w = tid >> 6, nw = T >> 6;
    if (idx >= b.h_off[rp.height + 1]) return;  // the same for the whole workgroup
    const int x = b.h_nodes[idx];
    if (tid == 0) { s_sum = 0; s_incomplete = 0; }
    kw::sync();
    { int sv = 0, inc = 0;
      for (int k = c.q_child_off[x] + tid; k < c.q_child_off[x + 1]; k += T) { const int s = c.q_children[k]; sv += b.q_valid[s]; if (!b.q_complete[s]) inc = 1; }
      if (sv) kw::atomic_add(&s_sum, sv); if (inc) kw::atomic_add(&s_incomplete, 1); }
    kw::sync();
    const int sumV = s_sum; const int compl_all = s_incomplete ? 0 : 1;
    int V = b.q_sent[x] < sumV ? b.q_sent[x] : sumV;
    const int eb = b.q_ebase[x], kb = b.q_kbase[x];
    if (x == c.Q) { if (tid == 0) { b.q_valid[x] = V; b.q_complete[x] = (compl_all && V == sumV) ? 1 : 0; b.q_nk[x] = 0; } return; }
    double alloc0[3], anp0[3];
    for (int k = 0; k < 3; k++) { alloc0[k] = c.q_share[(size_t)x * 3 + k].allocated; anp0[k] = c.q_share[(size_t)x * 3 + k].allocated_np; }
    const PlanNodeConst nc = plan_node_const(c, x, c.st->total[0], c.st->total[1], c.st->total[2]);  // what the gate and the keys read of the node, loaded once
    // pass 1: the node's own gate along the stream; stops at the first job it turns away
    {
        double alloc[3] = {alloc0[0], alloc0[1], alloc0[2]}, anp[3] = {anp0[0], anp0[1], anp0[2]};
        for (int base = 0; base < V; base += T * E) {
            const int t0 = base + tid * E;
            int meta[E]; double res[E][3], pre[E][6], tsum[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < E; j++) {  // (gathered by k_plan_gather) — the thread's own positions, each with the sums of the ones before it
                const int t = t0 + j; const bool is_elem = t < V;
                meta[j] = is_elem ? (int)b.d_meta[eb + t] : (int)BF_GATE;
                for (int k = 0; k < 3; k++) res[j][k] = is_elem ? b.d_res[(size_t)(eb + t) * 3 + k] : 0.0;
                const bool ok = (meta[j] & 3) == BF_OK, np = (meta[j] >> 2) & 1;
                for (int k = 0; k < 6; k++) pre[j][k] = tsum[k];
                for (int k = 0; k < 3; k++) { tsum[k] += ok ? res[j][k] : 0.0; tsum[3 + k] += (ok && np) ? res[j][k] : 0.0; }
            }
            double incl[6], tot[6];
            kb_block_scan_add<6>(L, tsum, incl, tot);
            bool gate[E]; int fb = 0x7fffffff;  // this thread's first position the node turns away (a job that would have been placed)
            for (int j = 0; j < E; j++) {
                const int t = t0 + j, flag = meta[j] & 3; const bool np = (meta[j] >> 2) & 1;
                double ab[3], abn[3];
                for (int k = 0; k < 3; k++) { ab[k] = alloc[k] + (incl[k] - tsum[k]) + pre[j][k]; abn[k] = anp[k] + (incl[3 + k] - tsum[3 + k]) + pre[j][3 + k]; }
                gate[j] = t < V && flag != BF_GATE && plan_gate_fails_c(nc, ab, abn, res[j], np);
                if (gate[j] && flag == BF_OK && fb == 0x7fffffff) fb = tid * E + j;
            }
            const uint64_t wm = kw::wave_max_u64(fb == 0x7fffffff ? 0ull : (uint64_t)(0x7fffffff - fb));  // the wave's smallest fb
            if (lane == 0) L.wbad[w] = wm ? 0x7fffffff - (int)wm : 0x7fffffff;
            kw::sync();
            int f = 0x7fffffff; for (int i = 0; i < nw; i++) if (L.wbad[i] < f) f = L.wbad[i];
            kw::sync();
            for (int j = 0; j < E; j++) if (gate[j] && tid * E + j <= f) { const int t = t0 + j; b.e_flag[b.el_leaf[eb + t]] = BF_GATE; b.d_meta[eb + t] = (uint8_t)(BF_GATE | (meta[j] & 4)); }  // positions before f hold exact shares: a dead job turned away here counts as a gate failure
            if (f != 0x7fffffff) { V = base + f + 1; break; }
            for (int k = 0; k < 3; k++) { alloc[k] += tot[k]; anp[k] += tot[3 + k]; }
        }
    }
    kw::fence(); kw::sync();
    const bool complete = compl_all && V == sumV;
    const int nk = complete ? V : V + 1;
    const int srank = b.q_srank[x];
    // pass 2: shares with the final flags, stale-path job and key before each pop
    {
        double alloc[3] = {alloc0[0], alloc0[1], alloc0[2]};
        PlanKey run; run.w0 = run.w1 = run.w2 = run.w3 = 0; bool have_run = false;
        for (int base = 0; base < nk; base += T * E) {
            const int t0 = base + tid * E;
            double res[E][3], pre[E][3], tsum[3] = {0, 0, 0};
            for (int j = 0; j < E; j++) {
                const int t = t0 + j; const bool is_elem = t < V;
                const int flag = is_elem ? (int)b.d_meta[eb + t] & 3 : (int)BF_GATE;
                for (int k = 0; k < 3; k++) { res[j][k] = (is_elem && flag == BF_OK) ? b.d_res[(size_t)(eb + t) * 3 + k] : 0.0; pre[j][k] = tsum[k]; tsum[k] += res[j][k]; }
            }
            double incl[3], tot[3];
            kb_block_scan_add<3>(L, tsum, incl, tot);
            // the keys of the thread's positions and their running maximum inside the thread.  The stale-path job (gathered by k_plan_gather): before the node's first pop its job of
            // the last round (cur_sp) or its first child's; after a pop from a child that still has a key, that child's next stale-path job; else the child popped from is gone and the
            // node's heap top is its best remaining child
            PlanKey key[E]; bool kv[E]; int spj[E];
            PlanKey tm; tm.w0 = tm.w1 = tm.w2 = tm.w3 = 0; bool tmv = false;
            for (int j = 0; j < E; j++) {
                const int t = t0 + j; const bool is_key = t < nk;
                kv[j] = is_key; spj[j] = is_key ? b.d_spj[kb + t] : -1;
                key[j].w0 = key[j].w1 = key[j].w2 = key[j].w3 = 0;
                if (is_key) {
                    double ab[3], rq[3];
                    for (int k = 0; k < 3; k++) { ab[k] = alloc[k] + (incl[k] - tsum[k]) + pre[j][k]; rq[k] = b.d_spres[(size_t)(kb + t) * 3 + k]; }
                    key[j] = plan_key_c(nc, ab, rq, srank);
                    if (tmv && pk_less(key[j], tm)) key[j] = tm;  // running maximum inside the thread
                    tm = key[j]; tmv = true;
                }
            }
            PlanKey pm, last; bool have_pm, have_last;
            kb_block_excl_max(L, tm, tmv, run, have_run, pm, have_pm, last, have_last);
            for (int j = 0; j < E; j++) if (kv[j]) {
                const int t = t0 + j;
                PlanKey out = key[j]; if (have_pm && pk_less(out, pm)) out = pm;
                b.pk[kb + t] = out; b.sp[kb + t] = spj[j]; b.k_owner[kb + t] = x;
            }
            run = last; have_run = have_last;
            for (int k = 0; k < 3; k++) alloc[k] += tot[k];
        }
    }
    if (tid == 0) { b.q_valid[x] = V; b.q_nk[x] = nk; b.q_complete[x] = complete ? 1 : 0; }
}
// the planned global order: one thread per position of the virtual root's valid stream
KW_BODY void kb_plan_emit(const KaiCtx& c) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    const int t = kw::bid() * kw::bdim() + kw::tid();
    if (t < 64 && b.cap_on) b.cls_cap[t] = 0;  // (the plan's leaves have read it: ready for the next round's sum)
    if (t >= b.q_valid[c.Q]) return;
    const int e = b.el_leaf[b.q_ebase[c.Q] + t], job = b.e_job[e];
    const int flag = b.e_flag[e];
    b.e_grank[e] = t; b.g_job[t] = job; b.g_flag[t] = (uint8_t)flag;
    b.g_first[t] = c.j_first_pod[job]; b.g_nt[t] = flag == BF_GATE ? 0 : c.j_tta_n[job]; b.g_ucls[t] = b.j_ucls[job];
}

// ------------------------------------------------------------------------------------------------------ fill
// ONE wavefront walks the planned order and places every task: arg-max node of the task's scan class out of the three-level class
// index (block maxima [C][NB] in LDS — or in HBM when they do not fit —, super-block maxima and class tops in LDS), node update, index
// maintenance, all inside this wave: lane = node of the current 64-node block (its records stay in registers) for the block level,
// lane = class for the upper levels.  What bounds it is the number of instructions one wave issues per decision, so the hot path keeps its
// operands in registers (class table: lane k holds class k; broadcasts are v_readlane, not LDS round trips) and touches memory only for the
// 16-byte index entries and the 32-byte node update.
struct FillLds {
    int32_t placed_node[KB_PLACED_MAX]; int32_t placed_cls[KB_PLACED_MAX];
};
enum { FM_PLAIN = 0, FM_SHARDED = 2 };  // compile-time modes of the fill kernel: the session's own nodes / the virtual cluster of a node-sharded group (floors)
// block level of the index: 16-byte entries in LDS, or the session's HBM arrays (sum1_key / sum1_node) when C x NB entries do not fit
KW_BODY void idx_get(KW_LDS_PTR(IdxE) p, int i, uint64_t& key, int& node) { key = p[i].key; node = p[i].node; }  // member-wise: no struct copies across address spaces
KW_BODY void idx_set(KW_LDS_PTR(IdxE) p, int i, uint64_t key, int node) { p[i].key = key; p[i].node = node; }
struct L1Lds {
    KW_LDS_PTR(IdxE) e; int NB;
    KW_BODY void get(int k, int blk, uint64_t& key, int& node) const { idx_get(e, k * NB + blk, key, node); }
    KW_BODY void set(int k, int blk, uint64_t key, int node) const { idx_set(e, k * NB + blk, key, node); }
    KW_BODY void done() const {}
};
struct L1Hbm {
    KAI_GP(uint64_t) key; KAI_GP(int32_t) node; int NB;
    KW_BODY void get(int k, int blk, uint64_t& ky, int& nd) const { ky = key[(size_t)k * NB + blk]; nd = node[(size_t)k * NB + blk]; }
    KW_BODY void set(int k, int blk, uint64_t ky, int nd) const { key[(size_t)k * NB + blk] = ky; node[(size_t)k * NB + blk] = nd; }
    KW_BODY void done() const { kw::fence_wg(); }  // entries written by one lane are read by other lanes of this wave later on
};
// plugin bits the class key reads; SPEC instantiations of the fill kernel have them all set and R == 4 as compile-time facts
constexpr uint32_t KB_KEY_PLUGINS = KAI_PLUGIN_PREDICATES | KAI_PLUGIN_NODEAVAILABILITY | KAI_PLUGIN_RESOURCETYPE | KAI_PLUGIN_NODEPLACEMENT;
struct FillState {
    int bcur, sbcur; NodeRec rec;          // the current block: lane i holds node bcur*64 + i
    double creq[4]; uint32_t cflags;       // lane k: scan class k
    uint64_t topk; int topn;               // lane k: arg-max key of class k over the cluster and its node
    uint64_t c1k, c2k; int c1n, c2n; bool c1_dirty, c2_dirty;  // lane k: class k's index entry of the current block / super-block (written back when the wave moves on)
    int pend_n, pend_cls; uint64_t pend_key;  // node whose index entries are behind its record (see kb_fill_place), the class that is filling it, that class's top key
    KW_LDS_PTR(IdxE) l2;
    uint32_t plugins; int R, C, NB, NSB;
    int64_t n_loads, n_r1, n_r2, n_r3;
    int64_t cy[4];  // KAI_FILL_PROF: place (top, block, record) / block level / super-block level / class tops
};
template <bool SPEC> KW_BODY uint32_t kb_plugins(const FillState& f) { return SPEC ? KB_KEY_PLUGINS : f.plugins; }
template <bool SPEC> KW_BODY int kb_nres(const FillState& f) { return SPEC ? 4 : f.R; }
template <int MODE, bool SPEC>
KW_BODY uint64_t kb_lane_key(const FillState& f, int kk) {  // key of this lane's node for class kk (kk the same in every lane)
    double rq[4]; for (int r = 0; r < 4; r++) rq[r] = kw::bcast(f.creq[r], kk);
    return class_key_rec(kb_plugins<SPEC>(f), kb_nres<SPEC>(f), rq, kw::bcast(f.cflags, kk), kk, f.rec);
}
template <class L1>
KW_BODY void kb_fill_writeback(FillState& f, const L1& l1) {
    const int lane = kw::lane();
    if (f.bcur >= 0 && f.c1_dirty && lane < f.C) l1.set(lane, f.bcur, f.c1k, f.c1n);
    if (f.sbcur >= 0 && f.c2_dirty && lane < f.C) idx_set(f.l2, lane * f.NSB + f.sbcur, f.c2k, f.c2n);
    f.c1_dirty = false; f.c2_dirty = false;
    l1.done();
}
template <class L1>
KW_BODY void kb_fill_load_block(const KaiCtx& c, FillState& f, const L1& l1, int blk) {
    if (blk == f.bcur) return;
    const int lane = kw::lane(), sb = blk >> 6;
    if (f.bcur >= 0 && f.c1_dirty && lane < f.C) l1.set(lane, f.bcur, f.c1k, f.c1n);
    f.c1_dirty = false;
    kw::fence_wg();  // the record stores of this wave to a block it may be coming back to (and, in HBM, its index entries)
    f.rec = c.bt.nrec[(size_t)blk * KAI_BLOCK + lane]; f.bcur = blk; f.n_loads++;
    f.c1k = 0; f.c1n = 0;
    if (lane < f.C) l1.get(lane, blk, f.c1k, f.c1n);
    if (sb != f.sbcur) {
        if (f.sbcur >= 0 && f.c2_dirty && lane < f.C) idx_set(f.l2, lane * f.NSB + f.sbcur, f.c2k, f.c2n);
        f.c2_dirty = false; f.c2k = 0; f.c2n = 0;
        if (lane < f.C) idx_get(f.l2, lane * f.NSB + sb, f.c2k, f.c2n);
        f.sbcur = sb;
    }
}
// node n (in the current block) changed: bring the three index levels up to date for every class
template <int MODE, bool SPEC, class L1>
KW_BODY void kb_fill_node_changed(FillState& f, const L1& l1, int n) {
    const int lane = kw::lane(), blk = n >> 6, ln = n & 63, sb = blk >> 6;
    KB_T(t_l1);
    NodeRec rn;
    for (int r = 0; r < 4; r++) rn.idle[r] = kw::bcast(f.rec.idle[r], ln);
    rn.cnt_gpu = kw::bcast(f.rec.cnt_gpu, ln); rn.cnt_cpu = kw::bcast(f.rec.cnt_cpu, ln); rn.okmask = kw::bcast(f.rec.okmask, ln); rn.cpu_node = kw::bcast(f.rec.cpu_node, ln); rn.pad = 0;
    const bool act = lane < f.C; const int k = act ? lane : 0;
    // ---- level 1 (block): lane k holds class k's entry of this block
    const uint64_t kap = act ? class_key_rec(kb_plugins<SPEC>(f), kb_nres<SPEC>(f), f.creq, f.cflags, k, rn) : 0;
    const uint64_t o1k = f.c1k; const int o1n = f.c1n;
    uint64_t n1k = o1k; int n1n = o1n; bool need = false;
    if (act) {
        if (o1k != 0 && o1n == n) { if (kap >= o1k) n1k = kap; else need = true; }  // n was the block's best: a raised key keeps it there, a lowered one needs the others
        else if (key_better(kap, n, o1k, o1n)) { n1k = kap; n1n = n; }
    }
    uint64_t todo = kw::ballot(need);
    while (todo) {
        const int kk = __builtin_ctzll(todo); todo &= todo - 1;
        uint64_t key = kb_lane_key<MODE, SPEC>(f, kk); int bn = blk * KAI_BLOCK + lane;
        kw::wave_argmax_first(key, bn);
        if (lane == kk) { n1k = key; n1n = bn; }
        f.n_r1++;
    }
    const bool ch1 = act && (n1k != o1k || n1n != o1n);
    if (ch1) { f.c1k = n1k; f.c1n = n1n; f.c1_dirty = true; }
    KB_ACC(1, t_l1);
    if (!kw::ballot(ch1)) return;  // no block maximum moved: the upper levels stand
    KB_T(t_l2);
    // ---- level 2 (super-block of 64 blocks): lane k holds class k's entry of this super-block
    const uint64_t o2k = f.c2k; const int o2n = f.c2n;
    uint64_t n2k = o2k; int n2n = o2n; need = false;
    if (ch1) {
        if (o2k != 0 && (o2n >> 6) == blk) { if (n1k != 0 && n1k >= o2k) { n2k = n1k; n2n = n1n; } else need = true; }
        else if (key_better(n1k, n1n, o2k, o2n)) { n2k = n1k; n2n = n1n; }
    }
    todo = kw::ballot(need);
    while (todo) {
        const int kk = __builtin_ctzll(todo); todo &= todo - 1;
        const int e = sb * 64 + lane;
        uint64_t key = 0; int bn = KB_INF;
        if (e < f.NB) l1.get(kk, e, key, bn);
        const uint64_t pk1 = kw::bcast(n1k, kk); const int pn1 = kw::bcast(n1n, kk);
        if (e == blk) { key = pk1; bn = pn1; }  // this block's entry lives in lane kk's registers
        kw::wave_argmax_first(key, bn);
        if (lane == kk) { n2k = key; n2n = bn; }
        f.n_r2++;
    }
    const bool ch2 = act && (n2k != o2k || n2n != o2n);
    if (ch2) { f.c2k = n2k; f.c2n = n2n; f.c2_dirty = true; }
    KB_ACC(2, t_l2);
    if (!kw::ballot(ch2)) return;
    KB_T(t_l3);
    // ---- level 3 (class top, in registers)
    const uint64_t o3k = f.topk; const int o3n = f.topn;
    uint64_t n3k = o3k; int n3n = o3n; need = false;
    if (ch2) {
        if (o3k != 0 && (o3n >> 12) == sb) { if (n2k != 0 && n2k >= o3k) { n3k = n2k; n3n = n2n; } else need = true; }
        else if (key_better(n2k, n2n, o3k, o3n)) { n3k = n2k; n3n = n2n; }
    }
    todo = kw::ballot(need);
    while (todo) {
        const int kk = __builtin_ctzll(todo); todo &= todo - 1;
        uint64_t key = 0; int bn = KB_INF;
        if (lane < f.NSB) idx_get(f.l2, kk * f.NSB + lane, key, bn);
        const uint64_t pk2 = kw::bcast(n2k, kk); const int pn2 = kw::bcast(n2n, kk);
        if (lane == sb) { key = pk2; bn = pn2; }  // this super-block's entry lives in lane kk's registers
        kw::wave_argmax_first(key, bn);
        if (lane == kk) { n3k = key; n3n = bn; }
        f.n_r3++;
    }
    f.topk = n3k; f.topn = n3n;
    KB_ACC(3, t_l3);
}
// Statement.Allocate's node side (NodeInfo.addTaskResources, node_info.go:457-493) / its undo, on the record of node n (current block)
KW_BODY void kb_fill_update_rec(const KaiCtx& c, FillState& f, int n, int kcls, double sign) {
    double rq[4]; for (int r = 0; r < 4; r++) rq[r] = kw::bcast(f.creq[r], kcls);
    if (kw::lane() == (n & 63)) {
        for (int r = 0; r < 4; r++) { if (r >= f.R) continue; const double v = rq[r]; if (v == 0) continue; f.rec.idle[r] = f.rec.idle[r] - sign * v; }
        for (int r = 0; r < 4; r++) c.bt.nrec[n].idle[r] = f.rec.idle[r];
    }
}
template <int MODE, bool SPEC, class L1>
KW_BODY void kb_fill_flush(FillState& f, const L1& l1) { if (f.pend_n >= 0) { kb_fill_node_changed<MODE, SPEC>(f, l1, f.pend_n); f.pend_n = -1; } }
// One task of scan class kcls: the node it goes to, or -1.  The index is brought up to date LAZILY: while consecutive tasks of one class keep
// landing on the node that class's top pointed to — its key for the class did not drop below the top key the index holds, so no other node can
// have overtaken it — only the node's record changes; the index entries of every class follow in one step when another class is asked for,
// when the node stops being the class's best, or when the round ends.
// node-sharded fill (mode 2): may the class's best candidate (tk, tn) be used?  Only while it beats the best node the ranks did NOT offer.
KW_BODY bool kb_beats_floor(const KaiCtx& c, int kcls, uint64_t tk, int tn) {
    const BatchCtx& b = c.bt;
    const uint64_t fk = b.floors[kcls].key; const int fn = b.floors[kcls].node;
    return fk == 0 || key_better(tk, b.vmap[tn], fk, fn);
}
// One task of scan class kcls: the node it goes to, -1 = no node fits, -2 (node-sharded group) = ask the ranks again.  The index is brought up to date
// LAZILY: while consecutive tasks of one class keep landing on the node that class's top pointed to — its key for the class did not drop below the top key
// the index holds, so no other node can have overtaken it — only the node's record changes; the index entries of every class follow in one step when
// another class is asked for, when the node stops being the class's best, or when the round ends.
template <int MODE, bool SPEC, class L1>
KW_BODY int kb_fill_place(const KaiCtx& c, FillState& f, const L1& l1, int kcls) {
    // A class whose top is empty has no fitting node — and keeps having none: during allocate a node's free resources only shrink (a rollback returns to a state that had no more than
    // when the top was computed), so a top that is behind the pending node's record can only be too GOOD, never too bad.  Answered without catching the index up: a third of config 3's
    // attempted gangs are such (18 128 of 30 913), and the flush they used to force also ended the lazy follow of the class that was filling the pending node.
    if (MODE != FM_SHARDED && kw::bcast(f.topk, kcls) == 0) return -1;
    if (f.pend_n >= 0) {
        if (kcls == f.pend_cls) {
            const int ln = f.pend_n & 63;
            const uint64_t mine = kb_lane_key<MODE, SPEC>(f, kcls);
            const uint64_t kap = kw::bcast(mine, ln);
            bool stay = kap != 0 && kap >= f.pend_key;
            if (MODE == FM_SHARDED) stay = stay && kb_beats_floor(c, kcls, kap, f.pend_n);
            if (stay) { kb_fill_update_rec(c, f, f.pend_n, kcls, 1.0); return f.pend_n; }
        }
        kb_fill_flush<MODE, SPEC>(f, l1);
    }
    KB_T(t_p);
    const uint64_t tk = kw::bcast(f.topk, kcls); const int tn = kw::bcast(f.topn, kcls);
    if (MODE == FM_SHARDED) {  // no candidate at all is an answer only when no rank holds anything back for the class
        if (tk == 0) return c.bt.floors[kcls].key == 0 ? -1 : -2;
        if (!kb_beats_floor(c, kcls, tk, tn)) return -2;
    }
    if (tk == 0) return -1;
    kb_fill_load_block(c, f, l1, tn >> 6);
    kb_fill_update_rec(c, f, tn, kcls, 1.0);
    f.pend_n = tn; f.pend_cls = kcls; f.pend_key = tk;
    KB_ACC(0, t_p);
    return tn;
}
template <int MODE, bool SPEC, class L1>
KW_BODY void kb_fill_run(const KaiCtx& c, RoundParams rp, FillLds& L, FillState& f, const L1& l1, bool l1_in_lds) {
    const BatchCtx& b = c.bt;
    const int lane = kw::lane(), C = f.C, NB = f.NB, NSB = f.NSB;
    const bool sharded = MODE == FM_SHARDED;
    const int64_t tstart = kw::clock();
    KAI_GP(uint64_t) home_k = c.sum1_key; KAI_GP(int32_t) home_n = c.sum1_node;  // HBM home of the block level (the virtual cluster's in a node-sharded group)
    if (l1_in_lds) for (int i = lane; i < C * NB; i += 64) l1.set(i / NB, i % NB, home_k[i], home_n[i]);
    kw::sync();
    f.topk = 0; f.topn = KB_INF;
    for (int k = 0; k < C; k++) {  // upper levels from the block level
        for (int sb = 0; sb < NSB; sb++) {
            const int e = sb * 64 + lane;
            uint64_t key = 0; int bn = KB_INF;
            if (e < NB) l1.get(k, e, key, bn);
            kw::wave_argmax_first(key, bn);
            if (lane == 0) idx_set(f.l2, k * NSB + sb, key, bn);
        }
        kw::sync();
        uint64_t key = 0; int bn = KB_INF;
        if (lane < NSB) idx_get(f.l2, k * NSB + lane, key, bn);
        kw::wave_argmax_first(key, bn);
        if (lane == k) { f.topk = key; f.topn = bn; }
    }
    kw::sync();
    const int V = rp.mode != 1 ? b.q_valid[c.Q] : 0;  // mode 1: index levels and dead classes only (before the first plan)
    int64_t decisions = 0, attempted = 0, committed = 0, rollbacks = 0, ops = 0; int n_done = rp.start, mismatch = 0, floor_stop = 0;
    for (int base = rp.start; base < V && !mismatch && !floor_stop; base += 64) {
        const int gi = base + lane;
        const int my_flag = gi < V ? b.g_flag[gi] : BF_GATE, my_first = gi < V ? b.g_first[gi] : 0, my_nt = gi < V ? b.g_nt[gi] : 0, my_ucls = gi < V ? b.g_ucls[gi] : 0;
        const int cnt = V - base < 64 ? V - base : 64;
        for (int jj = 0; jj < cnt; jj++) {
            const int flag = kw::bcast(my_flag, jj), first = kw::bcast(my_first, jj), nt = kw::bcast(my_nt, jj), ucls = kw::bcast(my_ucls, jj);
            const int opoff = (int)ops + rp.ops0, stmtoff = (int)committed + rp.stmt0; const int64_t dec0 = decisions;  // ops0 / stmt0: what earlier launches of this round committed
            bool ok = flag != BF_GATE; int placed = 0;
            if (flag != BF_GATE) {
                for (int tb = 0; tb < nt && ok; tb += 64) {
                    int my_cls = ucls;
                    if (ucls < 0) my_cls = tb + lane < nt ? b.t_cls[first + tb + lane] : 0;  // a gang of several scan classes: its task list
                    const int tc = nt - tb < 64 ? nt - tb : 64;
                    for (int ti = 0; ti < tc; ti++) {
                        const int kcls = ucls >= 0 ? ucls : kw::bcast(my_cls, ti);
                        decisions++;
                        const int tn = kb_fill_place<MODE, SPEC>(c, f, l1, kcls);
                        if (tn == -2) { floor_stop = 1; ok = false; break; }
                        if (tn < 0) { ok = false; break; }
                        if (lane == 0) { L.placed_node[placed] = tn; L.placed_cls[placed] = kcls; b.t_node[first + placed] = sharded ? b.vmap[tn] : tn; }
                        placed++;
                    }
                }
                if (!ok) {  // Statement.Rollback: the undone operations in reverse order (a gang that placed nothing leaves nothing to undo — and the index as it is)
                    if (placed || floor_stop) {
                        kb_fill_flush<MODE, SPEC>(f, l1);
                        kw::sync();
                        for (int i = placed - 1; i >= 0; i--) {
                            const int n = L.placed_node[i];
                            kb_fill_load_block(c, f, l1, n >> 6); kb_fill_update_rec(c, f, n, L.placed_cls[i], -1.0); kb_fill_node_changed<MODE, SPEC>(f, l1, n);
                        }
                    }
                    if (floor_stop) { decisions = dec0; break; }  // the gang is taken back untouched: the next exchange starts with it
                    rollbacks += 2;
                } else { committed++; ops += nt; }
            }
            attempted++; n_done = base + jj + 1;
            if (lane == 0) { b.g_out[base + jj] = ok ? BF_OK : BF_DEAD; b.g_opoff[base + jj] = opoff; b.g_stmt[base + jj] = stmtoff; }
            if ((flag == BF_OK) != ok) { mismatch = 1; break; }
        }
    }
    kb_fill_flush<MODE, SPEC>(f, l1);
    kb_fill_writeback(f, l1);
    kw::sync();
    if (l1_in_lds) for (int i = lane; i < C * NB; i += 64) { uint64_t ky; int nd; l1.get(i / NB, i % NB, ky, nd); home_k[i] = ky; home_n[i] = nd; }
    const uint64_t dead = kw::ballot(lane < C && f.topk == 0 && (!sharded || b.floors[lane < C ? lane : 0].key == 0));  // sharded: out of candidates is not out of nodes
    if (lane == 0) {
        FillStatus s; s.n_done = n_done; s.mismatch = mismatch; s.all_dead = (C > 0 && dead == (C >= 64 ? ~0ull : ((1ull << C) - 1))) ? 1 : 0; s.planned = V; s.floor_stop = floor_stop; s.pad = 0;
        s.decisions = decisions; s.attempted = attempted; s.committed = committed; s.rollbacks = rollbacks; s.ops = ops; s.dead_mask = dead;
        s.cycles_total = kw::clock() - tstart; s.cycles_load = f.cy[0]; s.cycles_update = f.cy[1]; s.cycles_rescan = f.cy[2] + f.cy[3];
        s.block_loads = f.n_loads; s.rescans1 = f.n_r1; s.rescans2 = f.n_r2; s.rescans3 = f.n_r3;
        b.fs[0] = s; b.dead_mask[0] = dead;
    }
}
template <int MODE>
KW_BODY void kb_fill_mode(const KaiCtx& c, RoundParams rp, int l1_in_lds, FillLds& L, FillState& f, unsigned char* dyn, size_t off) {
    const bool spec = (c.plugins & KB_KEY_PLUGINS) == KB_KEY_PLUGINS && c.R == 4;  // the default plugin tier: the class key folds to its shortest form
    if (l1_in_lds) { L1Lds l1; l1.e = (KW_LDS_PTR(IdxE))(dyn + off); l1.NB = f.NB; if (spec) kb_fill_run<MODE, true>(c, rp, L, f, l1, true); else kb_fill_run<MODE, false>(c, rp, L, f, l1, true); }
    else { L1Hbm l1; l1.key = c.sum1_key; l1.node = c.sum1_node; l1.NB = f.NB;
           if (spec) kb_fill_run<MODE, true>(c, rp, L, f, l1, false); else kb_fill_run<MODE, false>(c, rp, L, f, l1, false); }
}
// dynamic LDS: [super-block level C x NSB][block level C x NB when it fits]
KW_BODY void kb_fill_state(const KaiCtx& c, RoundParams rp, FillState& f) {
    const int lane = kw::lane();
    f.bcur = -1; f.sbcur = -1; f.c1k = f.c2k = 0; f.c1n = f.c2n = 0; f.c1_dirty = f.c2_dirty = false; f.n_loads = f.n_r1 = f.n_r2 = f.n_r3 = 0; f.pend_n = -1; f.pend_cls = 0; f.pend_key = 0; f.topk = 0; f.topn = KB_INF; for (int i = 0; i < 4; i++) f.cy[i] = 0;
    f.plugins = c.plugins; f.R = c.R; f.C = c.C; f.NB = c.NB; f.NSB = c.NSB;
    const bool virt = rp.mode == 2 || (rp.mode == 1 && c.bt.world > 1);
    if (virt) { f.NB = (c.bt.vstate[0] + KAI_BLOCK - 1) / KAI_BLOCK; f.NSB = (f.NB + 63) / 64; if (f.NSB < 1) f.NSB = 1; }  // the virtual cluster of a node-sharded group
    for (int r = 0; r < 4; r++) f.creq[r] = 0; f.cflags = 0;
    if (lane < c.C) { const ClassRec cr = c.cls[lane]; for (int r = 0; r < 4; r++) f.creq[r] = cr.req[r]; f.cflags = class_flags(cr); }
    f.rec = make_node_rec(c, c.N);  // an empty record until the first block is loaded
    f.l2 = (KW_LDS_PTR(IdxE))(kw::dyn_lds());
}
// ONE variant of the fill per kernel (device): compiled together into one kernel the eight variants cost ~800 SGPR spills and a 35 000-instruction body; apart, 72 - 110
// spills each (profiles/r04_fill_kernels_resource_usage.txt).  MODE: the session's own nodes / the virtual cluster of a node-sharded group; SPEC: the default plugin tier
// with R = 4 (the class key folds to its shortest form); L1L: block level of the index in LDS / in HBM.
template <int MODE, bool SPEC, bool L1L>
KW_BODY void kb_fill_variant(const KaiCtx& c, RoundParams rp) {
    if (kb_round_off(c.bt)) return;
    KW_SHARED FillLds L;
    FillState f; kb_fill_state(c, rp, f);
    if (L1L) { L1Lds l1; l1.e = (KW_LDS_PTR(IdxE))(kw::dyn_lds() + (size_t)c.C * f.NSB * sizeof(IdxE)); l1.NB = f.NB; kb_fill_run<MODE, SPEC>(c, rp, L, f, l1, true); }
    else { L1Hbm l1; l1.key = c.sum1_key; l1.node = c.sum1_node; l1.NB = f.NB; kb_fill_run<MODE, SPEC>(c, rp, L, f, l1, false); }
}
KW_BODY bool kb_fill_spec(const KaiCtx& c) { return (c.plugins & KB_KEY_PLUGINS) == KB_KEY_PLUGINS && c.R == 4; }
KW_BODY void kb_fill(const KaiCtx& c, RoundParams rp, int l1_in_lds) {  // every variant behind one entry: the emulator's form (tests/host_sim)
    if (kb_round_off(c.bt)) return;
    KW_SHARED FillLds L;
    const int lane = kw::lane();
    FillState f; f.bcur = -1; f.sbcur = -1; f.c1k = f.c2k = 0; f.c1n = f.c2n = 0; f.c1_dirty = f.c2_dirty = false; f.n_loads = f.n_r1 = f.n_r2 = f.n_r3 = 0; f.pend_n = -1; f.pend_cls = 0; f.pend_key = 0; f.topk = 0; f.topn = KB_INF; for (int i = 0; i < 4; i++) f.cy[i] = 0;
    f.plugins = c.plugins; f.R = c.R; f.C = c.C; f.NB = c.NB; f.NSB = c.NSB;
    const bool virt = rp.mode == 2 || (rp.mode == 1 && c.bt.world > 1);
    if (virt) { f.NB = (c.bt.vstate[0] + KAI_BLOCK - 1) / KAI_BLOCK; f.NSB = (f.NB + 63) / 64; if (f.NSB < 1) f.NSB = 1; }  // the virtual cluster of a node-sharded group
    for (int r = 0; r < 4; r++) f.creq[r] = 0; f.cflags = 0;
    if (lane < c.C) { const ClassRec cr = c.cls[lane]; for (int r = 0; r < 4; r++) f.creq[r] = cr.req[r]; f.cflags = class_flags(cr); }
    f.rec = make_node_rec(c, c.N);  // an empty record until the first block is loaded
    unsigned char* dyn = kw::dyn_lds();
    f.l2 = (KW_LDS_PTR(IdxE))(dyn);
    const size_t off = (size_t)c.C * f.NSB * sizeof(IdxE);
    if (rp.mode == 2) kb_fill_mode<FM_SHARDED>(c, rp, l1_in_lds, L, f, dyn, off);
    else kb_fill_mode<FM_PLAIN>(c, rp, l1_in_lds, L, f, dyn, off);
}

// ------------------------------------------------------------------------------------------------------ node-axis sharding
// Class index of a record array, block level only: one wavefront per 64-record block, every class (k_index_build on node records).
KW_BODY void kb_index_from_recs(const KaiCtx& c, KAI_GP(const NodeRec) recs, int n_recs, KAI_GP(uint64_t) l1k, KAI_GP(int32_t) l1n, int nb, int blk0, int blk1) {
    const int blk = blk0 + kw::bid() * (kw::bdim() >> 6) + (kw::tid() >> 6), lane = kw::lane();
    if (n_recs < 0) { n_recs = c.bt.vstate[0]; nb = (n_recs + KAI_BLOCK - 1) / KAI_BLOCK; blk1 = nb; }  // the virtual cluster: its size is on the device
    if (blk >= blk1) return;
    const int n = blk * KAI_BLOCK + lane;
    NodeRec rec = make_node_rec(c, c.N);
    if (n < n_recs) rec = recs[n];
    for (int k = 0; k < c.C; k++) {
        const ClassRec cr = c.cls[k]; double rq[4]; for (int r = 0; r < 4; r++) rq[r] = cr.req[r];
        uint64_t key = n < n_recs ? class_key_rec(c.plugins, c.R, rq, class_flags(cr), k, rec) : 0; int bn = n;
        kw::wave_argmax_first(key, bn);
        if (lane == 0) { l1k[(size_t)k * nb + blk] = key; l1n[(size_t)k * nb + blk] = bn; }
    }
}
// records of the nodes this rank does not own are dead (no class fits): its index and its offers then cover exactly its own slice
KW_BODY void kb_shard_mask_nrec(const KaiCtx& c) {
    const int n = kw::bid() * kw::bdim() + kw::tid();
    if (n >= c.NB * KAI_BLOCK) return;
    if (n < c.bt.n_lo || n >= c.bt.n_hi) c.bt.nrec[n].okmask = 0;
}
// class keys of the own nodes
KW_BODY void kb_shard_keys(const KaiCtx& c) {
    const BatchCtx& b = c.bt;
    const int i = kw::bid() * kw::bdim() + kw::tid(), nloc = b.n_hi - b.n_lo;
    if (i < (c.N + 31) / 32) b.cand_bits[i] = 0;
    if (i >= nloc * c.C) return;
    const int k = i / nloc, n = b.n_lo + i % nloc;
    const ClassRec cr = c.cls[k]; double rq[4]; for (int r = 0; r < 4; r++) rq[r] = cr.req[r];
    const NodeRec rec = b.nrec[n];
    b.sh_keys[(size_t)k * c.N + n] = class_key_rec(c.plugins, c.R, rq, class_flags(cr), k, rec);
}
// One workgroup per scan class: the K best own nodes in (key desc, node asc) order are marked as offered, the K+1st is the class's floor.
// Every thread owns the nodes n_lo + tid, + T, …; a round takes the workgroup's best of the threads' current bests, the thread it came from
// moves on to its next best (its nodes worse than the one just taken).
KW_BODY void kb_shard_select(const KaiCtx& c) {
    const BatchCtx& b = c.bt;
    const int k = kw::bid(), T = kw::bdim(), t = kw::tid(), lane = kw::lane(), wave = t >> 6, nw = (T + 63) >> 6;
    KW_SHARED uint64_t s_key[16]; KW_SHARED int s_node[16]; KW_SHARED uint64_t s_wk; KW_SHARED int s_wn;
    KAI_GP(const uint64_t) keys = b.sh_keys + (size_t)k * c.N;
    uint64_t myk = 0; int myn = KB_INF;
    for (int n = b.n_lo + t; n < b.n_hi; n += T) { const uint64_t ky = keys[n]; if (key_better(ky, n, myk, myn)) { myk = ky; myn = n; } }
    ShardHdr* hdr = (ShardHdr*)(unsigned char*)b.send;
    for (int round = 0; round <= b.shard_k; round++) {
        // best of the wave by (key desc, NODE asc): lanes do not hold ascending nodes here (a thread's current best may be its second node)
        uint64_t wk = kw::wave_max_u64(myk);
        const uint64_t inv = kw::wave_max_u64((myk == wk && wk != 0) ? (uint64_t)(0x7fffffff - myn) + 1 : 0);
        int wn = inv ? 0x7fffffff - (int)(inv - 1) : KB_INF;
        if (lane == 0) { s_key[wave] = wk; s_node[wave] = wn; }
        kw::sync();
        if (t == 0) { uint64_t bk = 0; int bn = KB_INF; for (int w = 0; w < nw; w++) if (key_better(s_key[w], s_node[w], bk, bn)) { bk = s_key[w]; bn = s_node[w]; } s_wk = bk; s_wn = bn; }
        kw::sync();
        const uint64_t bk = s_wk; const int bn = s_wn;
        if (round == b.shard_k || bk == 0) { if (t == 0) { hdr->floor[k].key = bk; hdr->floor[k].node = bk ? bn : KB_INF; hdr->floor[k].pad = 0; } break; }
        if (t == 0) { uint32_t* w = (uint32_t*)&b.cand_bits[bn >> 5]; kw::atomic_or32(w, 1u << (bn & 31)); }
        if (myn == bn) {  // mine was taken: my next best is my best node that sorts after it
            myk = 0; myn = KB_INF;
            for (int n = b.n_lo + t; n < b.n_hi; n += T) { const uint64_t ky = keys[n]; if (!key_better(bk, bn, ky, n)) continue; if (key_better(ky, n, myk, myn)) { myk = ky; myn = n; } }
        }
        kw::sync();
    }
}
// the offered nodes in ascending node order with their records (one workgroup: prefix sums over the bitmap words of the own range)
KW_BODY void kb_shard_compact(const KaiCtx& c) {
    const BatchCtx& b = c.bt;
    const int T = kw::bdim(), t = kw::tid(), lane = kw::lane(), wave = t >> 6, nw = (T + 63) >> 6;
    KW_SHARED int s_part[16]; KW_SHARED int s_carry;
    ShardHdr* hdr = (ShardHdr*)(unsigned char*)b.send;
    int32_t* out_node = (int32_t*)((unsigned char*)b.send + sizeof(ShardHdr));
    NodeRec* out_rec = (NodeRec*)((unsigned char*)b.send + sizeof(ShardHdr) + (size_t)b.shard_mmax * 4);
    if (t == 0) s_carry = 0;
    kw::sync();
    const int w0 = b.n_lo >> 5, w1 = (b.n_hi + 31) >> 5;
    for (int base = w0; base < w1; base += T) {
        const int wi = base + t; const uint32_t word = wi < w1 ? b.cand_bits[wi] : 0u; const int cnt = __builtin_popcount(word);
        const int incl = kw::wave_scan_add(cnt);
        if (lane == 63) s_part[wave] = incl;
        kw::sync();
        int off = s_carry; for (int w = 0; w < wave; w++) off += s_part[w];
        off += incl - cnt;
        uint32_t m = word;
        while (m) { const int bit = __builtin_ctz(m); m &= m - 1; const int n = wi * 32 + bit; if (off < b.shard_mmax) { out_node[off] = n; out_rec[off] = b.nrec[n]; } off++; }
        kw::sync();
        if (t == T - 1) { int tot = 0; for (int w = 0; w < nw; w++) tot += s_part[w]; s_carry += tot; }
        kw::sync();
    }
    if (t == 0) { hdr->count = s_carry < b.shard_mmax ? s_carry : b.shard_mmax; hdr->pad = 0; }
    for (int k = c.C + t; k < 64; k += T) { hdr->floor[k].key = 0; hdr->floor[k].node = KB_INF; hdr->floor[k].pad = 0; }
}
// after the all-gather: the virtual cluster = the ranks' offers one after the other (rank ranges ascend, offers ascend: global node order),
// padded with dead records to a multiple of 64, and the best floor per class.  One workgroup per rank message + one for the floors.
KW_BODY void kb_shard_vbuild(const KaiCtx& c) {
    const BatchCtx& b = c.bt;
    const int r = kw::bid(), T = kw::bdim(), t = kw::tid();
    int base = 0, total = 0;
    for (int q = 0; q < b.world; q++) { const ShardHdr* h = (const ShardHdr*)((const unsigned char*)b.recv + (size_t)q * b.msg_bytes); if (q < r) base += h->count; total += h->count; }
    if (r < b.world) {
        const unsigned char* msg = (const unsigned char*)b.recv + (size_t)r * b.msg_bytes;
        const ShardHdr* h = (const ShardHdr*)msg; const int32_t* nodes = (const int32_t*)(msg + sizeof(ShardHdr)); const NodeRec* recs = (const NodeRec*)(msg + sizeof(ShardHdr) + (size_t)b.shard_mmax * 4);
        for (int i = t; i < h->count; i += T) { b.vrec[base + i] = recs[i]; b.vmap[base + i] = nodes[i]; }
    } else {  // floors + padding
        for (int k = t; k < c.C; k += T) {
            uint64_t fk = 0; int fn = KB_INF;
            for (int q = 0; q < b.world; q++) { const ShardHdr* h = (const ShardHdr*)((const unsigned char*)b.recv + (size_t)q * b.msg_bytes); if (key_better(h->floor[k].key, h->floor[k].node, fk, fn)) { fk = h->floor[k].key; fn = h->floor[k].node; } }
            b.floors[k].key = fk; b.floors[k].node = fn; b.floors[k].pad = 0;
        }
        const int padded = (total + KAI_BLOCK - 1) / KAI_BLOCK * KAI_BLOCK;
        for (int i = total + t; i < padded; i += T) { b.vrec[i] = make_node_rec(c, c.N); b.vmap[i] = KB_INF; }
        if (t == 0) b.vstate[0] = total;
#if !defined(__HIPCC__)
        if (t == 0 && std::getenv("KAI_SHARD_TRACE")) { std::fprintf(stderr, "[shard r%d] total %d nodes:", b.rank, total); for (int i = 0; i < total && i < 12; i++) std::fprintf(stderr, " %d(gpu %.0f)", b.vmap[i], b.vrec[i].idle[2]); std::fprintf(stderr, " floor0 %llx/%d\n", (unsigned long long)b.floors[0].key, b.floors[0].node); }
#endif
    }
}
// what the virtual fill left on the nodes this rank owns goes back into its own records
KW_BODY void kb_shard_scatter(const KaiCtx& c, int total) {
    const BatchCtx& b = c.bt;
    const int i = kw::bid() * kw::bdim() + kw::tid();
    if (i >= total || i >= b.vstate[0]) return;  // the virtual cluster's size is on the device
    const int n = b.vmap[i];
    if (n >= b.n_lo && n < b.n_hi) b.nrec_home[n] = b.vrec[i];
}

// ------------------------------------------------------------------------------------------------------ apply
// Statement.Commit of every committed job of the executed prefix (framework/statement.go:536-575): pod state, committed operations in
// commit order, pod-set / job counters, node accounting, proportion event handlers up the queue chain (proportion.go:443-465).
// Quantities add exactly in any order (HostPrep::batch_units), so f64 atomics reproduce the sequential sums bit for bit.
KW_BODY void kb_apply_jobs(const KaiCtx& c, int64_t ops_base, int64_t stmt_base) {
    if (kb_round_off(c.bt)) return;
    if (c.bt.dev_loop) { ops_base = c.bt.ctl->ops_base; stmt_base = c.bt.ctl->stmt_base; }  // where the round's operations / Statements start in the action's output
    const BatchCtx& b = c.bt;
    const int tid = kw::tid(), T = kw::bdim(), t = kw::bid() * T + tid, n_done = b.fs[0].n_done;
    if (kw::bid() * T >= n_done) return;  // (the same for the whole workgroup)
    // The shares of the INNER queue nodes are summed per workgroup in LDS first: every committed job adds to every node of its chain, and the few top-level queues took tens of
    // thousands of f64 atomics each on one cache line (config 5's big round: 72 k jobs x 3 resources on 8 top-level queues; the kernel was bound by those lines, 0.27 ms).  Leaves
    // (thousands of them, a handful of jobs each) are added to directly.  Sums in another order are exact on this path (HostPrep::batch_units).
    KW_SHARED double s_acc[KB_APPLY_INNER * 6];
    const int n_in = b.n_inner < KB_APPLY_INNER ? b.n_inner : KB_APPLY_INNER;
    for (int i = tid; i < n_in * 6; i += T) s_acc[i] = 0.0;
    kw::sync();
    if (t < n_done && b.g_out[t] == BF_OK && b.g_flag[t] != BF_GATE) {
        const int j = b.g_job[t], first = c.j_first_pod[j], nt = c.j_tta_n[j], s = c.j_first_ps[j];
        double sum[3] = {0, 0, 0};
        for (int i = 0; i < nt; i++) {
            const int p = c.tta[first + i], n = b.t_node[first + i];
            kai_op o; o.seq = ops_base + b.g_opoff[t] + i; o.kind = KAI_OP_ALLOCATE; o.pod = p; o.node = n; o.job = j; o.stmt = (int32_t)(stmt_base + b.g_stmt[t]); o.pad = 0;
            c.out_ops[o.seq] = o;
            c.p_status[p] = KAI_POD_BINDING; c.p_node[p] = n; c.p_on_node[p] = n; c.p_on_node_status[p] = KAI_POD_ALLOCATED; c.p_accepted[p] = 1; c.p_virtual[p] = 1;
            for (int r = 0; r < c.R; r++) {
                const double v = c.p_req[(size_t)r * c.P + p]; if (v == 0) continue;
                kw::atomic_add((double*)&c.n_used[(size_t)r * c.N + n], v); kw::atomic_add((double*)&c.n_idle[(size_t)r * c.N + n], -v);
            }
            sum[0] += c.p_req[(size_t)KAI_RES_CPU * c.P + p]; sum[1] += c.p_req[(size_t)KAI_RES_MEM * c.P + p]; sum[2] += c.p_req[(size_t)KAI_RES_GPU * c.P + p];
        }
        c.s_active_alloc[s] += nt; c.s_active_used[s] += nt; c.j_n_pending[j] -= nt; c.j_tta_valid[j] = 0;
        for (int k = 0; k < 3; k++) c.j_allocated[(size_t)j * 4 + k] += sum[k];
        const bool np = !c.j_preempt[j];
        for (int q = c.j_queue[j]; q >= 0; q = c.q_parent[q]) {
            const int slot = b.q_islot[q];
            for (int k = 0; k < 3; k++) {
                if (sum[k] == 0) continue;
                if (slot >= 0 && slot < n_in) { kw::atomic_add(&s_acc[slot * 6 + k], sum[k]); if (np) kw::atomic_add(&s_acc[slot * 6 + 3 + k], sum[k]); }
                else { kw::atomic_add((double*)&c.q_share[(size_t)q * 3 + k].allocated, sum[k]); if (np) kw::atomic_add((double*)&c.q_share[(size_t)q * 3 + k].allocated_np, sum[k]); }
            }
        }
    }
    kw::sync();
    for (int i = tid; i < n_in * 6; i += T) {
        const double v = s_acc[i]; if (v == 0) continue;
        const int q = b.h_nodes[b.h_off[1] + i / 6], k = i % 3;
        if (i % 6 < 3) kw::atomic_add((double*)&c.q_share[(size_t)q * 3 + k].allocated, v); else kw::atomic_add((double*)&c.q_share[(size_t)q * 3 + k].allocated_np, v);
    }
}
// how far every queue node got inside the executed prefix: leaf cursors, stale-path jobs of the inner nodes
KW_BODY void kb_apply_nodes(const KaiCtx& c) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    const int x = kw::bid() * kw::bdim() + kw::tid();
    if (x >= c.Q) return;
    const int A = b.fs[0].n_done, V = b.q_valid[x], eb = b.q_ebase[x];
    const bool leaf = kb_is_leaf(c, x);
    int lo = 0, hi = V;  // elements of the node's stream inside the executed prefix: a prefix of the stream
    while (lo < hi) { const int mid = (lo + hi) >> 1; const int e = leaf ? eb + mid : b.el_leaf[eb + mid]; if (b.e_grank[e] < A) lo = mid + 1; else hi = mid; }
    if (lo == 0) return;
    if (leaf) c.lq_cur[x] += lo;
    else b.cur_sp[x] = lo < b.q_nk[x] ? b.sp[b.q_kbase[x] + lo] : -1;
}

// ------------------------------------------------------------------------------------------------------ rounds without the host
// What batch_allocate's loop did between two rounds on the host — read the fill's status, add it to the action's sums, move the output bases, decide how far the next plan looks
// (the rule and its measurements: kai_batch_driver.hpp batch_round_policy), stop when the queue is empty or no class fits anywhere — as ONE thread behind the round's apply kernels.
KW_BODY void kb_round_init(const KaiCtx& c, int remaining, int H0, int policy, int64_t ops_base, int64_t stmt_base) {  // behind the mode-1 fill: its verdict on the classes opens the loop
    if (kw::bid() != 0 || kw::tid() != 0) return;
    const BatchCtx& b = c.bt;
    RoundCtl r{};
    r.H = H0; r.remaining = remaining; r.policy = policy; r.ops_base = ops_base; r.stmt_base = stmt_base;
    if (remaining <= 0) r.done = 1;
    else if (b.fs[0].all_dead) { r.done = 1; r.drain = 1; }
    b.ctl[0] = r;
}
KW_BODY void kb_round_next(const KaiCtx& c) {
    if (kw::bid() != 0 || kw::tid() != 0) return;
    const BatchCtx& b = c.bt;
    RoundCtl r = b.ctl[0];
    if (r.done) return;
    const FillStatus fs = b.fs[0];
    if (fs.n_done <= 0) { r.fault = 1; r.done = 1; b.ctl[0] = r; return; }  // a round always executes at least one job
    r.rounds++; r.mismatches += fs.mismatch; r.planned += fs.planned; r.max_h = r.max_h > r.H ? r.max_h : r.H;
    r.decisions += fs.decisions; r.attempted += fs.attempted; r.committed += fs.committed; r.rollbacks += fs.rollbacks; r.ops += fs.ops;
    r.fill_cycles += fs.cycles_total; r.fill_load += fs.cycles_load; r.fill_update += fs.cycles_update; r.fill_rescan += fs.cycles_rescan;
    r.block_loads += fs.block_loads; r.rescans1 += fs.rescans1; r.rescans2 += fs.rescans2; r.rescans3 += fs.rescans3;
    r.last_h = r.H; r.last_planned = fs.planned; r.last_done = fs.n_done; r.last_mismatch = fs.mismatch; r.last_decisions = fs.decisions; r.last_steps = fs.rescans2; r.last_committed = fs.committed;
    if (fs.mismatch) { const int t = fs.n_done - 1; r.mm_flag = b.g_flag[t]; r.mm_out = b.g_out[t]; r.mm_nt = b.g_nt[t]; r.mm_cls = b.g_ucls[t]; }
    r.ops_base += fs.ops; r.stmt_base += fs.committed; r.remaining -= fs.n_done;
    r.H = kb_round_policy(r.policy, r.H, fs.mismatch != 0, fs.n_done, fs.planned);
    if (r.remaining <= 0) r.done = 1;
    else if (fs.all_dead) { r.done = 1; r.drain = 1; }
    b.ctl[0] = r;
}
// the action's sums into the engine's state (what the host added to a copy of it after its loop)
KW_BODY void kb_round_finish(const KaiCtx& c) {
    if (kw::bid() != 0 || kw::tid() != 0) return;
    const RoundCtl r = c.bt.ctl[0];
    c.st->decisions += r.decisions; c.st->jobs_attempted += r.attempted; c.st->jobs_committed += r.committed; c.st->rollbacks += r.rollbacks; c.st->out_len += r.ops; c.st->stmts += r.committed;
    c.st->index_queries += r.decisions; c.st->drain_pending = r.drain;
}

#if defined(__HIPCC__)
__global__ void k_batch_static_rank(KaiCtx c) { kb_static_rank(c); }
__global__ void k_batch_static_check(KaiCtx c) { kb_static_check(c); }
__global__ void k_batch_qualify(KaiCtx c) { kb_qualify(c); }
__global__ void k_batch_nrec(KaiCtx c) { kb_build_nrec(c); }
__global__ void __launch_bounds__(1024) k_plan_setup(KaiCtx c, RoundParams rp) { kb_plan_setup(c, rp); }
__global__ void k_plan_leaf(KaiCtx c, RoundParams rp) { kb_plan_leaf(c, rp); }
__global__ void k_plan_rank(KaiCtx c, RoundParams rp) { kb_plan_rank(c, rp); }
__global__ void k_plan_gather(KaiCtx c, RoundParams rp) { kb_plan_gather(c, rp); }
__global__ void __launch_bounds__(1024) k_plan_scan(KaiCtx c, RoundParams rp) { kb_plan_scan(c, rp); }
__global__ void k_plan_emit(KaiCtx c) { kb_plan_emit(c); }
template <int MODE, bool SPEC, bool L1L> __global__ void __launch_bounds__(64) k_fill(KaiCtx c, RoundParams rp) { kb_fill_variant<MODE, SPEC, L1L>(c, rp); }
__global__ void k_apply_jobs(KaiCtx c, long long ops_base, long long stmt_base) { kb_apply_jobs(c, (int64_t)ops_base, (int64_t)stmt_base); }
__global__ void k_apply_nodes(KaiCtx c) { kb_apply_nodes(c); }
__global__ void k_round_init(KaiCtx c, int remaining, int H0, int policy, long long ops_base, long long stmt_base) { kb_round_init(c, remaining, H0, policy, (int64_t)ops_base, (int64_t)stmt_base); }
__global__ void k_round_next(KaiCtx c) { kb_round_next(c); }
__global__ void k_round_finish(KaiCtx c) { kb_round_finish(c); }
__global__ void k_index_from_recs(KaiCtx c, const NodeRec* recs, int n_recs, uint64_t* l1k, int32_t* l1n, int nb, int blk0, int blk1) { kb_index_from_recs(c, (KAI_GP(const NodeRec))recs, n_recs, (KAI_GP(uint64_t))l1k, (KAI_GP(int32_t))l1n, nb, blk0, blk1); }
__global__ void k_shard_mask_nrec(KaiCtx c) { kb_shard_mask_nrec(c); }
__global__ void k_shard_keys(KaiCtx c) { kb_shard_keys(c); }
__global__ void k_shard_select(KaiCtx c) { kb_shard_select(c); }
__global__ void k_shard_compact(KaiCtx c) { kb_shard_compact(c); }
__global__ void k_shard_vbuild(KaiCtx c) { kb_shard_vbuild(c); }
__global__ void k_shard_scatter(KaiCtx c, int total) { kb_shard_scatter(c, total); }
#endif

}  // namespace kai

#include "kai_fill_buckets.hpp"
#include "kai_fill_counts.hpp"
#include "kai_fill_levels.hpp"
#include "kai_plan_segments.hpp"
