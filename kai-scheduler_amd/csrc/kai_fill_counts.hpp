// kai_fill_counts.hpp — the bucket fill (kai_fill_buckets.hpp) taken apart into two wavefronts that run side by side.
//
// kai_fill_buckets.hpp walks the planned order on ONE wavefront: per step it looks the best node up in the sets, moves its bit, patches every class's best, writes the
// tasks' nodes — ≈ 1 000 cycles per decision, all of it one dependency chain (profiles/r04q_*: 134 instructions per decision at 7.7 cycles each; one host core runs the same
// loop ≈ 7x faster, tests/host_sim/native_bucket_fill.hpp).  Most of that chain is not needed to DECIDE anything.  When no class carries a static bitmap of its own
// ("plain": every class may use every live node — BASELINE configs 2 and 5), which node a task lands on never feeds back into a decision: a class that asks for q devices takes
// the lowest non-empty level g >= q, and whether a gang fits, where its tasks go by LEVEL, what every later gang finds — all of it is a function of the levels'
// POPULATIONS alone.  So:
//
//   * wavefront 0, the counting machine: the planned order over cnt[g] = nodes with g free devices, 8 .. 16 integers in registers.  A gang of one class is decided before
//     anything moves (it fits iff Σ_g (g / q)·cnt[g] >= its tasks — the argument of kai_fill_buckets.hpp's capacity check, now the rule for every such gang), a gang of several
//     classes is simulated on a copy of the counts; job outcomes, decision counts and operation offsets come out exactly as the one-wave kernel produces them.  What it emits is a
//     stream of commands "the first k nodes of level g take `per` tasks each and move to level g2; their tasks are t_node[tbase ..)" through a ring in LDS.  It never touches a
//     bitmap and never rolls anything back: only commands of gangs that fit are published.
//   * wavefront 1, the set worker: executes the commands on the sets (lane = level as before: ds_xor moves, two summary levels, the first node of each level cached in its lane)
//     and writes the tasks' nodes.  No class state, no outcomes, no rollback.
//
// The chain that bounds the action is now the longer of the two, and each is a fraction of the old one.  Results are identical to k_fill_buckets (and through it to the oracle):
// tests/test_batch_path.py and tests/test_gpu_parity.py run the three fills against each other.  Everything that is not plain — static class bitmaps — and the
// one-placement-per-step debug mode stay on k_fill_buckets.
#pragma once
#include "kai_fill_buckets.hpp"

namespace kai {

constexpr int KFC_RING = 2048;  // commands the ring holds; a gang (<= KB_PLACED_MAX tasks) is written in full before it is published
struct FcCmd { int32_t lv, k, per, tbase; };  // lv = g | g2 << 8: the first k nodes of level g move to level g2 (0: no level), `per` tasks each; t_node[tbase ..) receives the nodes
struct FcMove { int32_t w, cmd; uint64_t mask; };  // hand-over from the worker of the upper levels to the worker of the lower ones: the nodes `mask` of word w enter command cmd's target level (cmd bit 30: its last entry)
struct FcLds { FcCmd ring[KFC_RING]; FcMove xring[KFC_RING]; int32_t cnt0[KBK_GMAX]; int32_t head, tail0, tail1, done, xhead, xtail, pad0, pad1; int64_t a_wait, b_idle[2], b_total[2]; };  // (the clocks: profiling)

KW_BODY void kb_fill_counts(const KaiCtx& c, RoundParams rp, BucketParams bp) {
    if (kb_round_off(c.bt)) return;
    KW_SHARED FcLds L;
    const BatchCtx& b = c.bt;
    const int tid = kw::tid(), T = kw::bdim(), lane = kw::lane(), C = c.C;
    BkView v; v.NW = bp.nw; v.NW1 = bp.nw1; v.LV = bp.levels;
    unsigned char* dyn = kw::dyn_lds();
    v.gw = (KW_LDS_PTR(uint64_t))dyn; v.s1 = v.gw + (size_t)v.LV * v.NW; v.ok = v.s1 + (size_t)v.LV * v.NW1 + KBK_GMAX;
    const int64_t tstart = kw::clock();
    if (tid < KBK_GMAX) L.cnt0[tid] = 0;
    if (tid == 0) { L.head = 0; L.tail0 = 0; L.tail1 = 0; L.done = 0; L.xhead = 0; L.xtail = 0; L.b_idle[0] = L.b_idle[1] = 0; L.b_total[0] = L.b_total[1] = 0; }
    for (int i = tid; i < v.LV * v.NW; i += T) v.gw[i] = b.bk_words[i];
    kw::sync();
    for (int i = tid; i < v.LV * v.NW1; i += T) {  // first summary level, and the levels' populations on the way
        const int l = i / v.NW1, w1 = i % v.NW1; uint64_t m = 0; int pop = 0;
        for (int j = 0; j < 64 && w1 * 64 + j < v.NW; j++) { const uint64_t x = v.gw[l * v.NW + w1 * 64 + j]; if (x) { m |= 1ull << j; pop += __builtin_popcountll(x); } }
        v.s1[i] = m;
        if (pop) kw::atomic_add((int32_t*)&L.cnt0[l], pop);
    }
    kw::sync();
    const int V = rp.mode != 1 ? b.q_valid[c.Q] : 0;  // mode 1: dead classes only (before the first plan)
    // two set workers from four levels on: wavefront 1 owns the levels 1 .. SPLIT, wavefront 2 the levels above (nodes only move DOWN the levels, so the upper worker never waits for the lower one)
    const int SPLIT = v.LV >= 4 ? (v.LV >= 8 ? v.LV / 2 - 1 : v.LV / 2) : v.LV;
    const bool two_workers = SPLIT < v.LV;
    if (tid < 64) {
        // ------------------------------------------------------------------ wavefront 0: the counting machine.  lane l: nodes of level l + 1; lane k: class k's request
        int cnt = lane < v.LV ? L.cnt0[lane] : 0;
        const bool act = lane < C;
        const int q = act ? (int)c.cls[lane].req[KAI_RES_GPU] : 0x7fffffff;
        uint32_t nz = (uint32_t)kw::ballot(cnt > 0);  // bit l: level l + 1 holds a node
        int decisions = 0, attempted = 0, committed = 0, rollbacks = 0, ops = 0, steps = 0, n_done = rp.start, mismatch = 0;
        int wp = 0, tail_seen = 0, pub = 0;  // commands written / the workers' progress as last read / commands published (per 64 commands, at the end of a stretch of jobs and before every wait: a release store per gang is a wait per gang)
        int64_t a_wait = 0;         // cycles this wavefront waited for room in the ring
#ifdef KAI_FILL_PROF
        int64_t pcy[4] = {0, 0, 0, 0};  // per job: decoding its parameters / deciding it and emitting its commands / its outcome; jobs
#endif
        auto tails_min = [&]() { const int t0 = kw::lds_load_acq(&L.tail0); if (!two_workers) return t0; const int t1 = kw::lds_load_acq(&L.tail1); return t0 < t1 ? t0 : t1; };  // a slot is free once BOTH workers have read it
        // the lowest non-empty level >= qc, 0 = none
        #define KFC_LEVEL_FOR(qc) ((nz >> ((qc) - 1)) ? (qc) + __builtin_ctz(nz >> ((qc) - 1)) : 0)
        // k nodes leave level g for level g2 (0: none): the counts and the non-empty mask, from values this lane already holds
        #define KFC_MOVE(g, g2, k, cg) do { if (lane == (g) - 1) cnt -= (k); if ((cg) == (k)) nz &= ~(1u << ((g) - 1)); if ((g2) >= 1) { if (lane == (g2) - 1) cnt += (k); nz |= 1u << ((g2) - 1); } } while (0)
        // the 64 jobs of a stretch: one per lane; the NEXT stretch's loads are issued before this stretch is walked (a wavefront that waits out four HBM loads per 64 jobs waits ~3 ms per C5 cycle)
        // (unconditional loads on a clamped index; the bound is applied when the values are used, so nothing waits for them before the walk)
        int nx_flag = 0, nx_first = 0, nx_nt = 0, nx_ucls = 0;
        if (V > rp.start) { const int gc = rp.start + lane < V ? rp.start + lane : V - 1; nx_flag = b.g_flag[gc]; nx_first = b.g_first[gc]; nx_nt = b.g_nt[gc]; nx_ucls = b.g_ucls[gc]; }
        for (int base = rp.start; base < V && !mismatch; base += 64) {
            const int gi = base + lane;
            const int my_flag = gi < V ? nx_flag : BF_GATE, my_first = gi < V ? nx_first : 0, my_nt = gi < V ? nx_nt : 0, my_ucls = gi < V ? nx_ucls : 0;
            { const int gc = gi + 64 < V ? gi + 64 : V - 1; nx_flag = b.g_flag[gc]; nx_first = b.g_first[gc]; nx_nt = b.g_nt[gc]; nx_ucls = b.g_ucls[gc]; }
            const int my_pack = my_flag | ((my_ucls + 1) << 2) | (my_nt << 12);  // (flag: 2 bits, class + 1: up to 64, tasks: up to KB_PLACED_MAX) — one readlane per job instead of three
            const int jn = V - base < 64 ? V - base : 64;
            // jobs the plan turned away at a capacity gate take no part in the fill: their outcome is written here, the walk below steps over them
            uint64_t todo = kw::ballot(gi < V && my_flag != BF_GATE);
            int my_out = BF_DEAD, my_opoff = 0, my_stmt = 0, n_out = jn;  // lane jj: what job jj of this stretch ended with (stored once per stretch, coalesced)
            attempted += jn; n_done = base + jn;
            while (todo) {
#ifdef KAI_FILL_PROF
                const int64_t pj0 = kw::clock();
#endif
                const int jj = __builtin_ctzll(todo); todo &= todo - 1;
                const int pack = kw::bcast(my_pack, jj), first = kw::bcast(my_first, jj);
                const int flag = pack & 3, ucls = ((pack >> 2) & 0x3ff) - 1, nt = pack >> 12;
                const int opoff = ops + rp.ops0, stmtoff = committed + rp.stmt0;
                bool ok = true;
#ifdef KAI_FILL_PROF
                const int64_t pj1 = kw::clock();
#endif
                if (ucls >= 0) {
                    // a gang of ONE class: it fits iff the levels hold enough places for it (a node of level g holds g / q of its tasks, every placement takes exactly one
                    // place away) — else it places `cap` tasks, finds no node for the next one and is rolled back: cap + 1 decisions, the state it started from
                    const int qc = kw::bcast(q, ucls);
                    // the usual gang: all of it fits on the class's best node (its lowest non-empty level g holds nt·q devices) — one command, no capacity sum, no divisions
                    const int g0 = KFC_LEVEL_FOR(qc), need0 = nt * qc;
                    if (g0 && need0 <= g0) {
                        if (wp - tail_seen >= KFC_RING) { kw::lds_store_rel(&L.head, wp); pub = wp; const int64_t w0 = kw::clock(); while (wp - tail_seen >= KFC_RING) { tail_seen = tails_min(); if (wp - tail_seen >= KFC_RING) kw::relax(); } a_wait += kw::clock() - w0; }
                        const int cg = kw::bcast(cnt, g0 - 1), g2 = g0 - need0;
                        if (lane == 0) { FcCmd cm; cm.lv = g0 | (g2 << 8); cm.k = 1; cm.per = nt; cm.tbase = first; L.ring[wp & (KFC_RING - 1)] = cm; }
                        wp++; steps++;
                        KFC_MOVE(g0, g2, 1, cg);
                        decisions += nt;
                        if (wp - pub >= 64) { kw::lds_store_rel(&L.head, wp); pub = wp; }
                    } else {
                    int cap;
                    if (nt == 1) cap = (nz >> (qc - 1)) ? 1 : 0;
                    else { int term = 0; if (lane < v.LV && lane + 1 >= qc) term = bk_div_small(lane + 1, qc) * cnt; cap = 0; for (int l = qc - 1; l < v.LV && cap < nt; l++) cap += kw::bcast(term, l); }
                    if (cap < nt) { decisions += cap + 1; ok = false; }
                    else {
                        int done = 0;
                        while (done < nt) {
                            if (wp - tail_seen >= KFC_RING) { kw::lds_store_rel(&L.head, wp); pub = wp; const int64_t w0 = kw::clock(); while (wp - tail_seen >= KFC_RING) { tail_seen = tails_min(); if (wp - tail_seen >= KFC_RING) kw::relax(); } a_wait += kw::clock() - w0; }
                            const int g = KFC_LEVEL_FOR(qc), r = bk_div_small(g, qc), rem = nt - done, cg = kw::bcast(cnt, g - 1);
                            int k = 1, per = rem;
                            if (rem >= r) { per = r; k = bk_div_small(rem, r); if (k > cg) k = cg; }
                            const int g2 = g - per * qc;
                            if (lane == 0) { FcCmd cm; cm.lv = g | (g2 << 8); cm.k = k; cm.per = per; cm.tbase = first + done; L.ring[wp & (KFC_RING - 1)] = cm; }
                            wp++; steps++;
                            KFC_MOVE(g, g2, k, cg);
                            done += k * per;
                        }
                        decisions += nt;
                        if (wp - pub >= 64) { kw::lds_store_rel(&L.head, wp); pub = wp; }
                    }
                    }
                } else {
                    // a gang of several scan classes: task by task on a copy of the counts; its commands stay unpublished until the last task has found its level
                    if (wp - tail_seen > KFC_RING - KB_PLACED_MAX) { kw::lds_store_rel(&L.head, wp); pub = wp; const int64_t w0 = kw::clock(); while (wp - tail_seen > KFC_RING - KB_PLACED_MAX) { tail_seen = tails_min(); if (wp - tail_seen > KFC_RING - KB_PLACED_MAX) kw::relax(); } a_wait += kw::clock() - w0; }
                    const int cnt_s = cnt; const uint32_t nz_s = nz; const int wp_s = wp;
                    for (int tb = 0; tb < nt && ok; tb += 64) {
                        const int my_cls = tb + lane < nt ? b.t_cls[first + tb + lane] : 0;
                        const int tc = nt - tb < 64 ? nt - tb : 64;
                        for (int ti = 0; ti < tc; ti++) {
                            const int qc = kw::bcast(q, kw::bcast(my_cls, ti));
                            decisions++;
                            const int g = KFC_LEVEL_FOR(qc);
                            if (!g) { ok = false; break; }
                            const int cg = kw::bcast(cnt, g - 1), g2 = g - qc;
                            if (lane == 0) { FcCmd cm; cm.lv = g | (g2 << 8); cm.k = 1; cm.per = 1; cm.tbase = first + tb + ti; L.ring[wp & (KFC_RING - 1)] = cm; }
                            wp++; steps++;
                            KFC_MOVE(g, g2, 1, cg);
                        }
                    }
                    if (ok) { if (wp - pub >= 64) { kw::lds_store_rel(&L.head, wp); pub = wp; } }
                    else { cnt = cnt_s; nz = nz_s; wp = wp_s; }  // Statement.Rollback: nothing was published
                }
#ifdef KAI_FILL_PROF
                const int64_t pj2 = kw::clock();
#endif
                if (ok) { committed++; ops += nt; } else rollbacks += 2;
                { const bool me = lane == jj; my_out = me ? (ok ? BF_OK : BF_DEAD) : my_out; my_opoff = me ? opoff : my_opoff; my_stmt = me ? stmtoff : my_stmt; }
                if ((flag == BF_OK) != ok) { mismatch = 1; n_done = base + jj + 1; n_out = jj + 1; attempted -= jn - (jj + 1); break; }
#ifdef KAI_FILL_PROF
                { const int64_t pj3 = kw::clock(); pcy[0] += pj1 - pj0; pcy[1] += pj2 - pj1; pcy[2] += pj3 - pj2; pcy[3]++; }
#endif
            }
            if (lane < n_out) { b.g_out[base + lane] = (uint8_t)my_out; b.g_opoff[base + lane] = my_opoff; b.g_stmt[base + lane] = my_stmt; }
            if (wp != pub) { kw::lds_store_rel(&L.head, wp); pub = wp; }
        }
        #undef KFC_LEVEL_FOR
        #undef KFC_MOVE
        kw::lds_store_rel(&L.head, wp);
        kw::lds_store_rel(&L.done, 1);
        const uint64_t dead = kw::ballot(act && (q > 32 || (nz >> (q - 1)) == 0));
        if (lane == 0) {
            FillStatus s; s.n_done = n_done; s.mismatch = mismatch; s.all_dead = (C > 0 && dead == (C >= 64 ? ~0ull : ((1ull << C) - 1))) ? 1 : 0; s.planned = V; s.floor_stop = 0; s.pad = 0;
            s.decisions = decisions; s.attempted = attempted; s.committed = committed; s.rollbacks = rollbacks; s.ops = ops; s.dead_mask = dead;
            s.cycles_total = kw::clock() - tstart; s.cycles_load = a_wait; s.cycles_update = 0; s.cycles_rescan = 0; s.block_loads = 0;  // (cycles_update / cycles_rescan, block_loads / rescans1: the workers' idle and total clocks, added below)
            s.rescans1 = 0; s.rescans2 = steps; s.rescans3 = 0;
#ifdef KAI_FILL_PROF
            s.cycles_load = pcy[0]; s.cycles_update = pcy[1]; s.cycles_rescan = pcy[2]; s.rescans3 = pcy[3];
#endif
  // rescans2: commands (a command moves the first k nodes of a level)
            b.fs[0] = s; b.dead_mask[0] = dead;
        }
    } else if (tid < 128 || (tid < 192 && two_workers)) {
        // ------------------------------------------------------------------ wavefronts 1 and 2: the set workers.  lane l owns level l + 1 — its words, its summaries, its first node — in
        // the worker that owns that level.  Both read every command.  The worker that owns a command's SOURCE level removes the nodes and writes the tasks' nodes; if it also owns the
        // target level it inserts them in the same ds_xor, else (source above SPLIT, target at or below) it hands (word, mask) over through xring and the lower worker inserts them when it
        // reaches that command — every level sees its removals and insertions in command order.
        const int me = tid < 128 ? 0 : 1, own_lo = me == 0 ? 1 : SPLIT + 1, own_hi = me == 0 ? SPLIT : v.LV;
        const bool mine = lane + 1 >= own_lo && lane + 1 <= own_hi;
        uint64_t s2 = 0;
        if (mine) for (int j = 0; j < v.NW1; j++) if (v.s1[lane * v.NW1 + j]) s2 |= 1ull << j;
        int dummy = 0, firstn = KB_INF;  // firstn: the lowest name rank of this lane's level (KB_INF: the level is empty)
        auto own_first = [&]() { if (!s2) return KB_INF; const int w1 = __builtin_ctzll(s2); const uint64_t m1 = v.s1[lane * v.NW1 + w1]; const int w = w1 * 64 + __builtin_ctzll(m1); return w * 64 + __builtin_ctzll(v.gw[lane * v.NW + w]); };
        if (mine) firstn = own_first();
        int tail = 0, finds = 0, xp = 0, xseen = 0; int64_t b_idle = 0; const int64_t b_start = kw::clock();  // xp: hand-over entries written (upper worker) / read (lower worker)
        for (;;) {
            const int head = kw::lds_load_acq(&L.head);
            if (tail == head) {
                if (kw::lds_load_acq(&L.done) && tail == kw::lds_load_acq(&L.head)) break;
                const int64_t i0 = kw::clock(); while (kw::lds_load_acq(&L.head) == tail && !kw::lds_load_acq(&L.done)) kw::relax(); b_idle += kw::clock() - i0;
                continue;
            }
            FcCmd nxt = L.ring[tail & (KFC_RING - 1)];
            for (; tail < head; tail++) {
                const FcCmd cm = nxt; nxt = L.ring[(tail + 1) & (KFC_RING - 1)];  // (the next command's read is in flight while this one runs; a slot beyond `head` is read and not used)
                const int g = cm.lv & 0xff, g2 = cm.lv >> 8, per = cm.per; int left = cm.k, tb = cm.tbase;
                const bool own_g = g >= own_lo && g <= own_hi, own_g2 = g2 >= own_lo && g2 <= own_hi;
                if (own_g) {
                    const int to = own_g2 ? g2 : 0;  // (a target level of the other worker: removed here, inserted there)
                    while (left > 0) {
                        const int n = kw::bcast(firstn, g - 1), w = n >> 6;
                        int m = 1; uint64_t mask = 1ull << (n & 63);
                        if (left > 1) {  // several nodes: the set bits of n's word at its level, from n upwards (n is the level's first node)
                            uint64_t word = v.gw[(g - 1) * v.NW + w];
                            kw::lds_order();  // (every lane has read the word before its owner toggles it below)
                            m = __builtin_popcountll(word); mask = word;
                            if (m > left) { m = left; mask = 0; for (int j = 0; j < m; j++) { mask |= word & (0 - word); word &= word - 1; } }
                            for (int t0 = 0; t0 < m * per; t0 += 64) {  // task t of this word's share sits on the (t / per)-th node of the mask
                                const int t = t0 + lane;
                                if (t < m * per) { uint64_t mm = mask; for (int j = bk_div_small(t, per); j > 0; j--) mm &= mm - 1; b.t_node[tb + t] = (w << 6) + __builtin_ctzll(mm); }
                            }
                        } else if (lane < per) b.t_node[tb + lane] = n;  // one node (the usual command): its tasks all sit on n (per <= 16)
                        const uint64_t neww = bk_move_mask(v, s2, dummy, w, mask, g, to);
                        if (lane == g - 1) { if (neww) firstn = (w << 6) + __builtin_ctzll(neww); else { firstn = own_first(); finds++; } }
                        if (lane == to - 1 && n < firstn) firstn = n;
                        tb += m * per; left -= m;
                        if (!own_g2 && g2 >= 1) {  // hand the nodes over to the worker of the lower levels
                            if (xp - xseen >= KFC_RING) { while (xp - xseen >= KFC_RING) { xseen = kw::lds_load_acq(&L.xtail); if (xp - xseen >= KFC_RING) kw::relax(); } }
                            if (lane == 0) { FcMove mv; mv.w = w; mv.cmd = tail | (left == 0 ? 1 << 30 : 0); mv.mask = mask; L.xring[xp & (KFC_RING - 1)] = mv; }
                            xp++;
                            kw::lds_store_rel(&L.xhead, xp);
                        }
                    }
                } else if (own_g2) {  // the upper worker removes this command's nodes: take them in as they arrive
                    for (bool last = false; !last;) {
                        if (xp == xseen) { const int64_t i0 = kw::clock(); while ((xseen = kw::lds_load_acq(&L.xhead)) == xp) kw::relax(); b_idle += kw::clock() - i0; }
                        const FcMove mv = L.xring[xp & (KFC_RING - 1)]; xp++;
                        last = (mv.cmd >> 30) & 1;
                        (void)bk_move_mask(v, s2, dummy, mv.w, mv.mask, 0, g2);
                        const int n = (mv.w << 6) + __builtin_ctzll(mv.mask);
                        if (lane == g2 - 1 && n < firstn) firstn = n;
                        kw::lds_store_rel(&L.xtail, xp);
                    }
                }
            }
            kw::lds_store_rel(me == 0 ? &L.tail0 : &L.tail1, tail);
        }
        (void)finds;
        if (lane == 0) { L.b_idle[me] = b_idle; L.b_total[me] = kw::clock() - b_start; }
    }
    kw::sync();
#ifndef KAI_FILL_PROF
    if (tid == 0) { b.fs[0].cycles_update = L.b_idle[0]; b.fs[0].cycles_rescan = L.b_total[0]; b.fs[0].block_loads = L.b_idle[1]; b.fs[0].rescans1 = L.b_total[1]; }  // the lower / the upper worker: idle and total clocks
#endif
    for (int i = tid; i < v.LV * v.NW; i += T) b.bk_words[i] = v.gw[i];
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(256) k_fill_counts(KaiCtx c, RoundParams rp, BucketParams bp) { kb_fill_counts(c, rp, bp); }
#endif

}  // namespace kai
