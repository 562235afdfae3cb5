// kai_fill_buckets.hpp — the fill of the batch path for bin-packed GPU classes: nodes bucketed by free devices, all of it in LDS.
//
// What the fill has to answer per task is OrderedNodesByTask + FittingNode (framework/session.go:201-283) for the task's scan class: the fitting node
// with the largest Σ NodeOrderFn, ties to the lowest name.  With the default plugin tier and the bin-pack strategy on the GPU (plugins/nodeplacement/
// pack.go:45-64: 9·(1 − (free − min)/(max − min)), strictly decreasing in the node's free devices) every fitting node carries the same nodeavailability
// and resourcetype points (a GPU class never earns the CPU-only bonus, resourcetype.go:29-41), so the order of the fitting nodes of a class is
//     (free devices ascending, name rank ascending)
// — what kai_batch.hpp::class_key_rec encodes as a 64-bit key.  Free devices are small integers, so instead of keys this kernel keeps, per number of
// free devices g = 1 .. L, the SET of nodes with exactly g free devices as a bitmap over the name ranks with two summary levels (64 words / 64·64 words
// per bit), all in LDS: 8 levels x 65 536 nodes = 64 KB.  The best node of a class that asks for q devices is the first set bit of the lowest non-empty
// level g >= q; placing a task moves one bit from level g to level g − q.  Per class the current best is cached (lane = class) and patched in O(1) after
// every placement (the node that changed is the only one whose key moved); only a class whose best node stops fitting looks its next one up — one lane per
// level walks summary → word, the lowest level that finds a node wins.  No node record is loaded, no key is computed, nothing goes to HBM but the task's
// node: ~10 LDS accesses per decision instead of the general kernel's ≈ 530 instructions.
//
// When is "fits ⇔ free devices >= q" exact?  The fit of FittingNode also compares CPU, memory, pods and every other tracked resource
// (resource_requirment.go:126-140) and predicates.go:264-285 wants a free pod slot.  k_bucket_build proves per node, before the action, that none of those
// can bind before the devices do: for every class c that may use the node and every resource r != GPU, req_c[r]·free0 <= idle0[r]·q_c (exact 128-bit
// integer products).  Then whatever sequence of tasks lands on the node consumes at most idle0[r]·(devices consumed)/free0 of r, and a task that still finds
// its devices finds its share of r (DESIGN.md §5.2b has the two-line proof).  Static predicates (class_fit, node readiness, worker labels, MIG / DRA rules)
// are the okmask of the node records: a per-class bitmap ANDed into the lookup, dropped for classes it would not change.  Anything else — a node that fails
// the proof, a spread or CPU-placed class, fractional or > 16 free devices, a node-sharded group — and the action runs on the general kernel (k_fill), with
// identical results (tests/test_batch_path.py runs both against the oracle).
#pragma once
#include "kai_batch.hpp"

namespace kai {

constexpr int KBK_GMAX = 16;  // levels the home arrays hold: free devices 1 .. 16

struct BucketMeta {        // written by k_bucket_build (zeroed by the host before it)
    int32_t bad;           // != 0: the action does not qualify for the bucket fill (bit 0 class shape, 1 free amount, 2 a resource may bind first)
    int32_t max_free;      // largest free amount of a live node
    int32_t live, pad;     // nodes some class may use
    int32_t ok_miss[64];   // class k: live nodes its static predicates turn away (0 = the class needs no bitmap of its own)
};
struct BucketParams { int32_t levels, nw, nw1, n_ok; int8_t okslot[64]; };  // okslot[k]: LDS slot of class k's static bitmap, -1 = none needed

// per 64-node block: bucket words, static-predicate words, the proof that only the devices can bind
KW_BODY void kb_bucket_build(const KaiCtx& c) {
    const BatchCtx& b = c.bt;
    const int w = kw::bid() * (kw::bdim() >> 6) + (kw::tid() >> 6), lane = kw::lane(), NW = c.NB;
    if (w >= NW) return;
    BucketMeta* meta = (BucketMeta*)b.bk_meta;
    int bad = 0;
    if (w == 0 && lane < c.C) {  // class shape: bin-packed on the GPU, a whole number of 1 .. 16 devices, one pod slot
        const ClassRec cr = c.cls[lane];
        const double qd = cr.req[KAI_RES_GPU];
        if (cr.cpu_only || cr.r_place != KAI_RES_GPU || cr.strategy != KAI_BINPACK || !(qd >= 1) || qd > KBK_GMAX || qd != (double)(int)qd) bad |= 1;
        if (c.R > KAI_RES_PODS && !(cr.req[KAI_RES_PODS] >= 1)) bad |= 1;  // the free-pod-slot predicate is implied only for classes that take a slot
    }
    const NodeRec rec = b.nrec[(size_t)w * KAI_BLOCK + lane];
    const bool live = rec.okmask != 0;
    int fr = 0;
    if (live) {
        const double f = rec.idle[KAI_RES_GPU];
        if (!(f >= 0) || f > KBK_GMAX || f != (double)(int)f) bad |= 2; else fr = (int)f;
        for (int r = 0; r < c.R && r < 4; r++) if (!(rec.idle[r] >= 0) || rec.idle[r] >= 9.2e18) bad |= 4;
    }
    if (live && fr >= 1 && !bad) {
        for (int k = 0; k < c.C; k++) {
            if (!((rec.okmask >> k) & 1ull)) continue;
            const int q = (int)c.cls[k].req[KAI_RES_GPU];
            if (q < 1) continue;  // (flagged above)
            for (int r = 0; r < c.R && r < 4; r++) {
                if (r == KAI_RES_GPU) continue;
                const double rq = c.cls[k].req[r];
                if (!(rq > 0)) continue;
                if (rq >= 9.2e18) { bad |= 4; continue; }
                const unsigned __int128 lhs = (unsigned __int128)(uint64_t)rq * (unsigned)fr, rhs = (unsigned __int128)(uint64_t)rec.idle[r] * (unsigned)q;
                if (lhs > rhs) bad |= 4;
            }
        }
    }
    const uint64_t livew = kw::ballot(live);
    int top = 0;
    for (int g = 1; g <= KBK_GMAX; g++) {
        const uint64_t word = kw::ballot(live && fr == g);
        if (word) top = g;
        if (lane == 0) b.bk_words[(size_t)(g - 1) * NW + w] = word;
    }
    for (int k = 0; k < c.C; k++) {
        const uint64_t okw = kw::ballot(live && ((rec.okmask >> k) & 1ull));
        if (lane == 0) { b.bk_ok[(size_t)k * NW + w] = okw; const int miss = __builtin_popcountll(livew & ~okw); if (miss) kw::atomic_add((int32_t*)&meta->ok_miss[k], miss); }
    }
    const uint64_t anybad = kw::ballot(bad != 0);
    if (anybad) { int all = 0; for (int l = 0; l < 64; l++) all |= kw::shfl(bad, l); if (lane == 0) kw::atomic_max((int32_t*)&meta->bad, all); }
    if (lane == 0) { if (top) kw::atomic_max((int32_t*)&meta->max_free, top); if (livew) kw::atomic_add((int32_t*)&meta->live, __builtin_popcountll(livew)); }
}

// a / b for 0 <= a <= 1024, 1 <= b <= 64 without the integer-division sequence (the fill wave's steps divide small counts): (a + 1/2) / b lies at least 1 / (2b) from an
// integer, far more than a float reciprocal is off by
KW_BODY int bk_div_small(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)b));
#else
    return (int)(((float)a + 0.5f) * (1.0f / (float)b));  // (the same arithmetic on the emulator, so that the CPU tests exercise it)
#endif
}
struct BkLds { int32_t placed_node[KB_PLACED_MAX]; int32_t placed_info[KB_PLACED_MAX]; };  // the running gang, for its rollback (info: class | level before the placement << 8)
// (Measured and dropped, r04l: the placements staged in LDS and written to t_node once per 64 jobs, so that the loop over jobs issues no global store — the fill was no
// faster: the wave does not wait on its stores.  r04m, clocks around the four parts of a step: best node + mask 360, task records 250, the move 310, tops + lookups 430 cycles.)
constexpr uint32_t BK_DEAD = 0xffffffffu;  // a class's best node as one ascending key, level << 20 | node (N <= 2^18 nodes: the summaries' reach); BK_DEAD: none
KW_BODY uint32_t bk_key(int g, int n) { return n < 0 ? BK_DEAD : ((uint32_t)g << 20) | (uint32_t)n; }
struct BkView {
    KW_LDS_PTR(uint64_t) gw; KW_LDS_PTR(uint64_t) s1; KW_LDS_PTR(uint64_t) ok;
    int NW, NW1, LV;
};
// LANE l OWNS LEVEL l + 1: its words gw[l][·], its first summary s1[l][·] (LDS) and its second summary s2 (a register of that lane).  No lane ever
// touches another lane's level, so the levels need no ordering between lanes at all.
//
// The first node (lowest name rank) of the lowest level >= from_g that a class with static bitmap `slot` may use: every lane walks its own level
// (summary register -> summary word -> node word: two dependent LDS reads), the lowest level that finds one wins.
KW_BODY void bk_find(const BkView& v, uint64_t s2, int slot, int from_g, int& og, int& on) {
    const int lane = kw::lane();
    int n = KB_INF;
    if (lane < v.LV && lane + 1 >= from_g) {
        uint64_t m2 = s2;
        while (m2 && n == KB_INF) {
            const int w1 = __builtin_ctzll(m2); m2 &= m2 - 1;
            uint64_t m1 = v.s1[lane * v.NW1 + w1];
            while (m1) {
                const int w = w1 * 64 + __builtin_ctzll(m1); m1 &= m1 - 1;
                uint64_t word = v.gw[lane * v.NW + w];
                if (slot >= 0) word &= v.ok[slot * v.NW + w];
                if (word) { n = w * 64 + __builtin_ctzll(word); break; }
            }
        }
    }
    const uint64_t m = kw::ballot(n != KB_INF);
    if (!m) { og = 0; on = -1; return; }
    const int l = __builtin_ctzll(m);
    og = l + 1; on = kw::bcast(n, l);
}
// Node n leaves level `from` and enters level `to` (0 = no level: a node without a free device fits no class).  Both are the same operation on the owning
// lane's word — toggle bit n (ds_xor_rtn_b64, one LDS round trip for both levels) — and a summary bit toggles exactly when the word below it became
// empty or stopped being empty.  Returns, in the lane of level `from`, that level's word after the removal.
// (bk_move_mask: the same for several nodes of ONE word at once — `bit` holds their bits; they all leave `from` and enter `to`)
KW_BODY uint64_t bk_move_mask(const BkView& v, uint64_t& s2, int& cnt, int w, uint64_t bit, int from, int to) {
    const int lane = kw::lane(), w1 = w >> 6;
    const bool isfrom = lane == from - 1, part = isfrom || lane == to - 1;
    uint64_t old = 0;
    if (part) { old = kw::lds_xor(&v.gw[lane * v.NW + w], bit); const int kb = __builtin_popcountll(bit); cnt += isfrom ? -kb : kb; }
    const uint64_t neww = old ^ bit;
    const bool flip1 = part && (isfrom ? neww == 0 : old == 0);
    if (kw::ballot(flip1)) {
        if (flip1) {
            const uint64_t bit1 = 1ull << (w & 63), o1 = kw::lds_xor(&v.s1[lane * v.NW1 + w1], bit1);
            if (isfrom ? (o1 ^ bit1) == 0 : o1 == 0) s2 ^= 1ull << w1;
        }
    }
    return neww;
}
KW_BODY uint64_t bk_move(const BkView& v, uint64_t& s2, int& cnt, int n, int from, int to) {
    const int lane = kw::lane(), w = n >> 6, w1 = w >> 6;
    const uint64_t bit = 1ull << (n & 63);
    const bool isfrom = lane == from - 1, part = isfrom || lane == to - 1;
    uint64_t old = 0;
    if (part) { old = kw::lds_xor(&v.gw[lane * v.NW + w], bit); cnt += isfrom ? -1 : 1; }
    const uint64_t neww = old ^ bit;
    const bool flip1 = part && (isfrom ? neww == 0 : old == 0);
    if (kw::ballot(flip1)) {  // (uniform) a word became empty or stopped being empty: its bit in the first summary toggles, and so on upwards
        if (flip1) {
            const uint64_t bit1 = 1ull << (w & 63), o1 = kw::lds_xor(&v.s1[lane * v.NW1 + w1], bit1);
            if (isfrom ? (o1 ^ bit1) == 0 : o1 == 0) s2 ^= 1ull << w1;
        }
    }
    return neww;
}

// workgroup of 256: all four wavefronts move the state between its HBM home and LDS, wavefront 0 walks the planned order
KW_BODY void kb_fill_buckets(const KaiCtx& c, RoundParams rp, BucketParams bp) {
    if (kb_round_off(c.bt)) return;
    KW_SHARED BkLds L;
    const BatchCtx& b = c.bt;
    const int tid = kw::tid(), T = kw::bdim(), lane = kw::lane(), C = c.C;
    BkView v; v.NW = bp.nw; v.NW1 = bp.nw1; v.LV = bp.levels;
    unsigned char* dyn = kw::dyn_lds();
    v.gw = (KW_LDS_PTR(uint64_t))dyn; v.s1 = v.gw + (size_t)v.LV * v.NW; v.ok = v.s1 + (size_t)v.LV * v.NW1 + KBK_GMAX;
    const int64_t tstart = kw::clock();
    for (int i = tid; i < v.LV * v.NW; i += T) v.gw[i] = b.bk_words[i];
    for (int k = 0; k < C; k++) { const int s = bp.okslot[k]; if (s < 0) continue; for (int i = tid; i < v.NW; i += T) v.ok[s * v.NW + i] = b.bk_ok[(size_t)k * v.NW + i]; }
    kw::sync();
    for (int i = tid; i < v.LV * v.NW1; i += T) {
        const int l = i / v.NW1, w1 = i % v.NW1; uint64_t m = 0;
        for (int j = 0; j < 64 && w1 * 64 + j < v.NW; j++) if (v.gw[l * v.NW + w1 * 64 + j]) m |= 1ull << j;
        v.s1[i] = m;
    }
    kw::sync();
    if (tid < 64) {
        uint64_t s2 = 0;  // lane l: second summary of level l + 1
        if (lane < v.LV) for (int j = 0; j < v.NW1; j++) if (v.s1[lane * v.NW1 + j]) s2 |= 1ull << j;
        int lvl_n = 0;    // lane l: nodes of level l + 1 (kept by every move; what a class's capacity is summed from)
        if (lane < v.LV) for (int j = 0; j < v.NW; j++) lvl_n += __builtin_popcountll(v.gw[lane * v.NW + j]);
        // lane k: class k — devices asked for, slot of its static bitmap, and its best node as ONE ascending key: level << 20 | node (a class's order over the nodes
        // it may use is (free devices, name rank) ascending; BK_DEAD = no node fits).  Lanes beyond the classes hold BK_DEAD and ask for "infinitely many" devices.
        const bool act = lane < C;
        int q = 0x7fffffff, okslot = -1; uint32_t top = BK_DEAD;
        if (act) { q = (int)c.cls[lane].req[KAI_RES_GPU]; okslot = bp.okslot[lane]; }
        const bool batched = rp.pad2 == 0;  // (RoundParams::pad2 = 1: one placement per step for every gang — A/B runs, tests)
        const bool plain = kw::ballot(act && okslot >= 0) == 0;  // no class carries a static bitmap of its own: one lookup answers every class that lost the same node
        for (int k = 0; k < C; k++) { int g, n; bk_find(v, s2, kw::bcast(okslot, k), kw::bcast(q, k), g, n); if (lane == k) top = bk_key(g, n); }
        const int V = rp.mode != 1 ? b.q_valid[c.Q] : 0;  // mode 1: dead classes only (before the first plan)
        int decisions = 0, attempted = 0, committed = 0, rollbacks = 0, ops = 0, finds = 0, steps = 0, n_done = rp.start, mismatch = 0;  // (of one launch: far below 2^31)
#ifdef KAI_FILL_PROF
        int64_t qcy[4] = {0, 0, 0, 0};  // per step: best node + mask / task records / the move / tops
        int64_t pcy[4] = {0, 0, 0, 0};  // per job: before its tasks / its steps incl. the stretch stores / after them (rollback, outputs); [3] the stretch stores alone
#endif
        for (int base = rp.start; base < V && !mismatch; base += 64) {
            const int gi = base + lane;
            const int my_flag = gi < V ? b.g_flag[gi] : BF_GATE, my_first = gi < V ? b.g_first[gi] : 0, my_nt = gi < V ? b.g_nt[gi] : 0, my_ucls = gi < V ? b.g_ucls[gi] : 0;
            const int cnt = V - base < 64 ? V - base : 64;
            int my_out = 0, my_opoff = 0, my_stmt = 0, n_out = 0;  // lane jj: what job jj of this stretch ended with (stored once per stretch, coalesced)
            for (int jj = 0; jj < cnt; jj++) {
#ifdef KAI_FILL_PROF
                const int64_t pt0 = kw::clock();
#endif
                const int flag = kw::bcast(my_flag, jj), first = kw::bcast(my_first, jj), nt = kw::bcast(my_nt, jj), ucls = kw::bcast(my_ucls, jj);
                const int opoff = ops + rp.ops0, stmtoff = committed + rp.stmt0;
                bool ok = flag != BF_GATE; int placed = 0;
#ifdef KAI_FILL_PROF
                const int64_t pt1 = kw::clock(); int64_t pt2 = pt1;
#endif
                if (flag != BF_GATE) {
                    for (int tb = 0; tb < nt && ok; tb += 64) {
                        if (ucls >= 0 && plain && batched) {
                            // A gang of ONE class, no static bitmaps: whole nodes per step.  The class's best node n holds g free devices, so it takes r = g / q of the gang's
                            // tasks one after the other (fewer free devices is a better key: it stays on top while it fits); then it holds g mod q < q and the class's next best is the
                            // next node of the same level — levels q .. g-1 were empty and still are — which again takes r tasks, and so on.  So a step moves the first k nodes of
                            // n's word at level g to level g mod q at once (one ds_xor per level with the k bits) and hands out k·r tasks; a remainder of fewer than r tasks goes to one
                            // node, which then stays at level g - rem·q.  Every other class sees exactly what k·r single placements would have left: the only new candidates are the
                            // moved nodes, all at one level, the first of them the best.
                            const int tc = nt - tb < 64 ? nt - tb : 64, qc = kw::bcast(q, ucls);
                            if (tb == 0 && nt > 1 && flag == BF_DEAD) {  // (the plan's predictions of "dead" are exact: free resources only shrink)
                                // Does the gang fit at all?  A node with g free devices holds g / q of its tasks and every placement takes exactly one such place away (g - q holds
                                // g / q - 1), so the placements above succeed exactly while Σ_levels (g / q) · nodes(g) lasts: a gang that asks for more than that places `cap`
                                // tasks, finds no node for the next one and is rolled back — the state it started from, cap + 1 decisions.  Decided here without placing anything.
                                int term = 0;
                                if (lane < v.LV && lane + 1 >= qc) term = bk_div_small(lane + 1, qc) * lvl_n;
                                int cap = 0;
                                for (int l = qc - 1; l < v.LV; l++) cap += kw::bcast(term, l);
                                if (cap < nt) { decisions += cap + 1; steps++; ok = false; break; }
                            }
                            int my_node = 0, my_info = 0, done = 0;
                            while (done < tc) {
#ifdef KAI_FILL_PROF
                                const int64_t q0 = kw::clock();
#endif
                                const uint32_t tk = kw::bcast(top, ucls);
                                if (tk == BK_DEAD) { decisions++; ok = false; break; }
                                const int n = (int)(tk & 0xfffffu), g = (int)(tk >> 20), r = bk_div_small(g, qc), rem = tc - done, w = n >> 6;
                                int k = 1, per = rem < r ? rem : r;
                                uint64_t mask = 1ull << (n & 63);
                                if (rem >= 2 * r) {  // more than one whole node: the set bits of n's word at its level, from n upwards (n is the level's first node)
                                    uint64_t word = v.gw[(g - 1) * v.NW + w];
                                    kw::lds_order();  // (every lane has read the word before its owner toggles it below; lock-step on the device, an order the emulator needs told)
                                    const int want = bk_div_small(rem, r); mask = 0;
                                    for (k = 0; k < want && word; k++) { mask |= word & (0 - word); word &= word - 1; }
                                }
#ifdef KAI_FILL_PROF
                                const int64_t q1 = kw::clock();
#endif
                                const int g2 = g - per * qc, step = k * per;
                                {   // lane done + t: task t of the step, on the (t / per)-th node of the mask, found at level g - (t mod per)·q (what the rollback needs)
                                    uint64_t m = mask; int x = lane - done;  // (x: this lane's task counted from the node the loop is at)
                                    for (int j = 0; j < k; j++, x -= per) { const int nj = (w << 6) + __builtin_ctzll(m); m &= m - 1; if (x >= 0 && x < per) { my_node = nj; my_info = ucls | ((g - x * qc) << 8); } }
                                }
                                decisions += step; done += step; steps++;
#ifdef KAI_FILL_PROF
                                const int64_t q2 = kw::clock();
#endif
                                const uint64_t neww = bk_move_mask(v, s2, lvl_n, w, mask, g, g2);
#ifdef KAI_FILL_PROF
                                const int64_t q3 = kw::clock();
#endif
                                const uint32_t cand = ((uint32_t)g2 << 20) | (uint32_t)n;
                                const bool mine = top == tk, fits2 = g2 >= q;
                                const bool need = mine && !fits2;
                                top = (fits2 && (mine || cand < top)) ? cand : top;
                                if (kw::ballot(need)) {
                                    const uint64_t rest = kw::bcast(neww, g - 1);
                                    uint32_t fk = ((uint32_t)g << 20) | (uint32_t)((n & ~63) + (rest ? __builtin_ctzll(rest) : 0));
                                    if (!rest) { int fg, fn; bk_find(v, s2, -1, g, fg, fn); fk = bk_key(fg, fn); finds++; }
                                    top = need ? fk : top;
                                }
#ifdef KAI_FILL_PROF
                                { const int64_t q4 = kw::clock(); qcy[0] += q1 - q0; qcy[1] += q2 - q1; qcy[2] += q3 - q2; qcy[3] += q4 - q3; }
#endif
                            }
#ifdef KAI_FILL_PROF
                            const int64_t ps0 = kw::clock();
#endif
                            if (lane < done) { b.t_node[first + tb + lane] = my_node; L.placed_node[tb + lane] = my_node; L.placed_info[tb + lane] = my_info; }
#ifdef KAI_FILL_PROF
                            pcy[3] += kw::clock() - ps0;
#endif
                            placed += done;
                            continue;
                        }
                        int my_cls = ucls;
                        if (ucls < 0) my_cls = tb + lane < nt ? b.t_cls[first + tb + lane] : 0;  // a gang of several scan classes: its task list
                        const int tc = nt - tb < 64 ? nt - tb : 64;
                        int my_node = 0, my_info = 0, done = 0;  // lane i: the i-th placement of this stretch of (at most 64) tasks
                        for (int ti = 0; ti < tc; ti++) {
                            const int kcls = ucls >= 0 ? ucls : kw::bcast(my_cls, ti);
                            decisions++; steps++;
                            const uint32_t tk = kw::bcast(top, kcls);
                            if (tk == BK_DEAD) { ok = false; break; }
                            const int n = (int)(tk & 0xfffffu), g = (int)(tk >> 20), g2 = g - kw::bcast(q, kcls);
                            const bool me = lane == ti;
                            my_node = me ? n : my_node; my_info = me ? (kcls | (g << 8)) : my_info;
                            done++;
                            const uint64_t neww = bk_move(v, s2, lvl_n, n, g, g2);
                            // the only node whose key moved is n (to cand): a class that had it on top keeps it while it still fits (fewer free devices = a better
                            // key), any other class takes it if it now beats that class's best
                            const uint32_t cand = ((uint32_t)g2 << 20) | (uint32_t)n;
                            bool okn = true;
                            if (!plain) okn = okslot < 0 || ((v.ok[(okslot < 0 ? 0 : okslot) * v.NW + (n >> 6)] >> (n & 63)) & 1ull);
                            const bool mine = top == tk, fits2 = g2 >= q;
                            const bool need = mine && !fits2;
                            top = (fits2 && (mine || (okn && cand < top))) ? cand : top;
                            uint64_t todo = kw::ballot(need);
                            if (todo) {
                                // A class that lost n had it at level g as the FIRST node of the lowest level it can use, and n went below what it asks for: its
                                // next best is the first node at level >= g.  Without static bitmaps that is one answer for all of them — and usually the next
                                // bit of the word n just left, which the move already returned.
                                if (plain) {
                                    const uint64_t rest = kw::bcast(neww, g - 1);
                                    uint32_t fk = ((uint32_t)g << 20) | (uint32_t)((n & ~63) + (rest ? __builtin_ctzll(rest) : 0));
                                    if (!rest) { int fg, fn; bk_find(v, s2, -1, g, fg, fn); fk = bk_key(fg, fn); finds++; }
                                    top = need ? fk : top;
                                } else while (todo) {
                                    const int kk = __builtin_ctzll(todo); todo &= todo - 1;
                                    int fg, fn; bk_find(v, s2, kw::bcast(okslot, kk), g, fg, fn); finds++;
                                    if (lane == kk) top = bk_key(fg, fn);
                                }
                            }
                        }
                        // the stretch's placements: the tasks' nodes for the apply kernels (one coalesced store) and the rollback list
                        if (lane < done) { b.t_node[first + tb + lane] = my_node; L.placed_node[tb + lane] = my_node; L.placed_info[tb + lane] = my_info; }
                        placed += done;
                    }
#ifdef KAI_FILL_PROF
                    pt2 = kw::clock();
#endif
                    if (!ok) {  // Statement.Rollback: the undone operations in reverse order, then every class's best from the restored sets
                        kw::lds_order();
                        for (int i = placed - 1; i >= 0; i--) {
                            const int n = L.placed_node[i], info = L.placed_info[i], gb = info >> 8;
                            (void)bk_move(v, s2, lvl_n, n, gb - kw::bcast(q, info & 0xff), gb);
                        }
                        if (placed) for (int k = 0; k < C; k++) { int g, n; bk_find(v, s2, kw::bcast(okslot, k), kw::bcast(q, k), g, n); finds++; if (lane == k) top = bk_key(g, n); }
                        rollbacks += 2;
                    } else { committed++; ops += nt; }
                }
#ifdef KAI_FILL_PROF
                { const int64_t pt3 = kw::clock(); pcy[0] += pt1 - pt0; pcy[1] += pt2 - pt1; pcy[2] += pt3 - pt2; }
#endif
                attempted++; n_done = base + jj + 1;
                { const bool me = lane == jj; my_out = me ? (ok ? BF_OK : BF_DEAD) : my_out; my_opoff = me ? opoff : my_opoff; my_stmt = me ? stmtoff : my_stmt; n_out = jj + 1; }
                if ((flag == BF_OK) != ok) { mismatch = 1; break; }
            }
            if (lane < n_out) { b.g_out[base + lane] = (uint8_t)my_out; b.g_opoff[base + lane] = my_opoff; b.g_stmt[base + lane] = my_stmt; }
        }
        const uint64_t dead = kw::ballot(act && top == BK_DEAD);
        if (lane == 0) {
            FillStatus s; s.n_done = n_done; s.mismatch = mismatch; s.all_dead = (C > 0 && dead == (C >= 64 ? ~0ull : ((1ull << C) - 1))) ? 1 : 0; s.planned = V; s.floor_stop = 0; s.pad = 0;
            s.decisions = decisions; s.attempted = attempted; s.committed = committed; s.rollbacks = rollbacks; s.ops = ops; s.dead_mask = dead;
            s.cycles_total = kw::clock() - tstart; s.cycles_load = 0; s.cycles_update = 0; s.cycles_rescan = 0;
            s.block_loads = 0;
#ifdef KAI_FILL_PROF
            s.cycles_load = qcy[0]; s.cycles_update = qcy[1]; s.cycles_rescan = qcy[2]; s.block_loads = qcy[3]; (void)pcy;
#endif
            s.rescans1 = finds; s.rescans2 = steps; s.rescans3 = 0;  // rescans2: steps of the fill wave (a step places whole nodes of a one-class gang, or one task)
            b.fs[0] = s; b.dead_mask[0] = dead;
        }
    }
    kw::sync();
    for (int i = tid; i < v.LV * v.NW; i += T) b.bk_words[i] = v.gw[i];
}

// Tasks of every scan class the cluster holds at the committed state, summed into cls_cap: per node the pods of the class that fit it one after the other (the smallest
// quotient idle / request over the class's resources; the sets by free devices give it as free / q), capped at 4096 per node.  It feeds a PREDICTION of the plan (a gang of one
// class that asks for more tasks than the cluster holds cannot fit — and stays unfit, free resources only shrink during allocate); the fill verifies every prediction.
// One wavefront per 64-node block.
KW_BODY void kb_class_capacity(const KaiCtx& c, int buckets, int levels) {
    if (kb_round_off(c.bt)) return;
    const BatchCtx& b = c.bt;
    const int w = kw::bid() * (kw::bdim() >> 6) + (kw::tid() >> 6), lane = kw::lane(), NW = c.NB;
    if (w >= NW) return;
    if (buckets) {  // lane = class: the block's word of every level against the class's static bitmap
        if (lane >= c.C) return;
        const int q = (int)c.cls[lane].req[KAI_RES_GPU]; if (q < 1) return;
        const uint64_t okw = b.bk_ok[(size_t)lane * NW + w];
        int sum = 0;
        for (int g = q; g <= levels; g++) sum += (g / q) * __builtin_popcountll(b.bk_words[(size_t)(g - 1) * NW + w] & okw);
        if (sum) kw::atomic_add((int32_t*)&b.cls_cap[lane], sum);
        return;
    }
    const NodeRec rec = b.nrec[(size_t)w * KAI_BLOCK + lane];
    for (int k = 0; k < c.C; k++) {
        int m = 0;
        if ((rec.okmask >> k) & 1ull) {
            double best = 4096.0;
            for (int r = 0; r < c.R && r < 4; r++) { const double rq = c.cls[k].req[r]; if (!(rq > 0)) continue; const double f = rec.idle[r] >= rq ? rec.idle[r] / rq : 0.0; if (f < best) best = f; }
            if ((c.plugins & KAI_PLUGIN_PREDICATES) && !(rec.idle[KAI_RES_PODS] > 0)) best = 0.0;
            m = (int)best;
        }
        const int s = kw::wave_scan_add(m), tot = kw::shfl(s, 63);
        if (lane == 0 && tot) kw::atomic_add((int32_t*)&b.cls_cap[k], tot);
    }
}

#if defined(__HIPCC__)
__global__ void k_class_capacity(KaiCtx c, int buckets, int levels) { kb_class_capacity(c, buckets, levels); }
__global__ void k_bucket_build(KaiCtx c) { kb_bucket_build(c); }
__global__ void __launch_bounds__(256) k_fill_buckets(KaiCtx c, RoundParams rp, BucketParams bp) { kb_fill_buckets(c, rp, bp); }
#endif

}  // namespace kai
