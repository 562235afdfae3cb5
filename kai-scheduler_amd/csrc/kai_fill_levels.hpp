// kai_fill_levels.hpp — the fill of kai_fill_counts.hpp with the sets spread over ONE WAVEFRONT PER LEVEL and a counting machine that decides a stretch's dead gangs in bulk.
//
// kai_fill_counts.hpp split the bucket fill into a counting machine (the planned order over cnt[g] = nodes with g free devices) and two set workers that execute its commands on the
// LDS-resident sets.  Measured on BASELINE config 5 (profiles/r05*): the counting machine spends ≈ 1 000 cycles on every job that reaches it — also on the 40 % of them the plan
// already predicted dead —, and the two workers are busy 65 – 90 % of the time, i.e. either chain bounds the kernel at about the same length.  This kernel shortens both:
//
//   * The sets: level g belongs to wavefront g, alone.  Its words, its two summaries and its first node are that wavefront's uniform state (the word that holds the first node is
//     cached in registers: removing the level's first node — what every command does — reads nothing from LDS while that word lasts).  Every worker reads every command and acts on
//     the ones that name its level: as the SOURCE it removes the level's first k nodes, writes the tasks' nodes and hands (word, mask) to the target level's worker through the
//     ring of that (source, target) pair; as the TARGET it takes the nodes in when it reaches the command.  Every level sees its removals and insertions in command order, nodes only
//     move DOWN the levels, so the wait-for graph has no cycle (a producer waits only for a consumer that is behind it; the worker that is furthest behind waits for nobody).
//     Eight levels = eight short chains instead of two long ones.
//   * The counting machine: lane j of a stretch of 64 jobs holds job j.  A gang of one class that the plan predicted dead and that does not fit the capacities at the stretch's
//     start is dead for good (capacities only shrink during allocate): it takes no part in the walk.  All such a gang books is cap + 1 decisions with the capacity AT ITS TURN;
//     lane q − 1 keeps cap[q] = Σ_g (g / q)·cnt[g] up to date with every command (two table look-ups and a subtraction) and books the dead gangs of a segment between two walked
//     jobs with one population count.  Outcomes, Statement numbers and operation offsets of a stretch come out of one ballot and one prefix sum at its end instead of three
//     selects per job.  What is left per walked gang is: two v_readlane for its parameters, the level look-up on the scalar unit, one command, the counts' update.
//
// Results are identical to k_fill_counts / k_fill_buckets (and through them to the oracle): tests/test_batch_path.py and tests/test_gpu_parity.py run the fills against each other, the
// emulator runs this kernel's nine wavefronts as fibers that really interleave (KW_EMU_ORDER), tests/host_sim shadows every launch with the scalar C++ fill.  Clusters with more than
// eight levels (16-device nodes) or with static class bitmaps stay on k_fill_counts / k_fill_buckets.
#pragma once
#include "kai_fill_counts.hpp"

namespace kai {

constexpr int KFL_LMAX = 8;                                // levels = worker wavefronts
constexpr int KFL_RING = 2048;                             // commands the ring holds (a gang of several classes, <= KB_PLACED_MAX commands, is written in full before it is published)
constexpr int KFL_XR = 32;                                 // entries of a hand-over ring
constexpr int KFL_PAIRS = KFL_LMAX * (KFL_LMAX - 1) / 2;   // (source level g, target level g2 < g)
struct FlMove { uint64_t mask; int32_t w; int32_t seq; };  // the nodes `mask` of word w (bit 30 of w: the command's last entry); seq = the entry's number in its ring + 1, stored last (release)
struct FlLds {
    FcCmd ring[KFL_RING];
    FlMove x[KFL_PAIRS][KFL_XR];
    int32_t xtail[KFL_PAIRS];      // entries the pair's consumer has taken
    int32_t cnt0[KBK_GMAX];
    int32_t tail[KFL_LMAX];        // commands worker g has passed
    int32_t head, done;
    int64_t w_idle[KFL_LMAX], w_total[KFL_LMAX];  // (the clocks: profiling)
};
KW_BODY int kfl_pair(int g, int g2) { return (g - 1) * (g - 2) / 2 + (g2 - 1); }
// a / b for 0 <= a <= 1024, 1 <= b <= 8 on the scalar unit: a·ceil(2^15 / b) >> 15 (the error a·(ceil − exact) / 2^15 stays below 1/32, the fraction of a / b below 7/8)
KW_BODY int kfl_div(int a, int b) {
    const uint64_t inv = b <= 4 ? 0x2000'2AAB'4000'8000ull : 0x1000'124A'1556'199Aull;  // ceil(32768 / b), 16 bits each: b = 1..4 / 5..8
    return (int)(((uint32_t)a * (uint32_t)((inv >> (16 * ((b - 1) & 3))) & 0xffff)) >> 15);
}
// uniform accesses of a worker to its own level's LDS words: every lane reads the same address and takes lane 0's value (a scalar from then on); lane 0 writes
KW_BODY uint64_t kfl_read(KW_LDS_PTR(uint64_t) p) { return kw::bcast(*p, 0); }
KW_BODY void kfl_write(KW_LDS_PTR(uint64_t) p, uint64_t v) { if (kw::lane() == 0) *p = v; }

KW_BODY void kb_fill_levels(const KaiCtx& c, RoundParams rp, BucketParams bp) {
    KW_SHARED FlLds L;
    const BatchCtx& b = c.bt;
    const int tid = kw::tid(), T = kw::bdim(), lane = kw::lane(), C = c.C;
    BkView v; v.NW = bp.nw; v.NW1 = bp.nw1; v.LV = bp.levels;
    unsigned char* dyn = kw::dyn_lds();
    v.gw = (KW_LDS_PTR(uint64_t))dyn; v.s1 = v.gw + (size_t)v.LV * v.NW; v.ok = v.s1 + (size_t)v.LV * v.NW1 + KBK_GMAX;
    const int64_t tstart = kw::clock();
    if (tid < KBK_GMAX) L.cnt0[tid] = 0;
    if (tid < KFL_LMAX) { L.tail[tid] = 0; L.w_idle[tid] = 0; L.w_total[tid] = 0; }
    if (tid < KFL_PAIRS) L.xtail[tid] = 0;
    for (int i = tid; i < KFL_PAIRS * KFL_XR; i += T) L.x[i / KFL_XR][i % KFL_XR].seq = 0;
    if (tid == 0) { L.head = 0; L.done = 0; }
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
    if (tid < 64) {
        // ------------------------------------------------------------------ wavefront 0: the counting machine.
        // lane l: cnt = nodes of level l + 1.  lane q − 1: tab = the quotients g / q, g = 0 .. 8, four bits each; capq = Σ_g (g / q)·cnt[g], the tasks that ask for q devices the levels hold.
        // lane k: qk = devices class k asks for.
#if defined(KFL_EXP_PRIO) && defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_setprio(3);
#endif
        int cnt = lane < v.LV ? L.cnt0[lane] : 0;
        const bool act = lane < C;
        const int qk = act ? (int)c.cls[lane].req[KAI_RES_GPU] : 0x7fffffff;
        uint64_t tab = 0;
        for (int g = 1; g <= v.LV; g++) tab |= (uint64_t)(g / (lane + 1)) << (4 * g);
        int capq = 0;
        for (int g = 1; g <= v.LV; g++) capq += (int)((tab >> (4 * g)) & 15) * kw::bcast(cnt, g - 1);
        uint32_t nz = (uint32_t)kw::ballot(cnt > 0);  // bit l: level l + 1 holds a node
        int decisions = 0, attempted = 0, committed = 0, rollbacks = 0, ops = 0, n_done = rp.start, mismatch = 0;
        int dec_v = 0;                    // lane q − 1: decisions of the dead gangs that ask for q devices, without their "+ 1"s
        int wp = 0, tail_seen = 0, pub = 0;  // commands written / the slowest worker's progress as last read / commands published (per 64 commands and at the end of a stretch)
        int64_t a_wait = 0;               // cycles this wavefront waited for room in the ring
#ifdef KFL_PROF
        int64_t pc[5] = {0, 0, 0, 0, 0}; int64_t pt = kw::clock();
        #define KFL_T(i) do { const int64_t n_ = kw::clock(); pc[i] += n_ - pt; pt = n_; } while (0)
#else
        #define KFL_T(i) (void)0
#endif
        auto tails_min = [&]() { int t = lane < v.LV ? kw::lds_load_acq(&L.tail[lane]) : 0x7fffffff; for (int l = 1; l < v.LV; l++) { const int o = kw::bcast(t, l); t = o < t ? o : t; } return kw::bcast(t, 0); };
        // the lowest non-empty level >= qc, 0 = none
        #define KFL_LEVEL_FOR(qc) ((nz >> ((qc) - 1)) ? (qc) + __builtin_ctz(nz >> ((qc) - 1)) : 0)
        #define KFL_ROOM(n) do { if (wp - tail_seen > KFL_RING - (n)) { kw::lds_store_rel(&L.head, wp); pub = wp; const int64_t w0 = kw::clock(); while (wp - tail_seen > KFL_RING - (n)) { tail_seen = tails_min(); if (wp - tail_seen > KFL_RING - (n)) kw::relax(); } a_wait += kw::clock() - w0; } } while (0)
        #define KFL_EMIT(g_, g2_, k_, per_, tb_) do { if (lane == 0) { FcCmd cm; cm.lv = (g_) | ((g2_) << 8); cm.k = (k_); cm.per = (per_); cm.tbase = (tb_); L.ring[wp & (KFL_RING - 1)] = cm; } wp++; } while (0)
        // k nodes leave level g for level g2 (0: none): the counts, the non-empty mask, every request size's capacity
        #ifdef KFL_EXP_NOCAP
#define KFL_CAPUPD(g_, g2_, k_) (void)0
#else
#define KFL_CAPUPD(g_, g2_, k_) capq -= (k_) * ((int)((tab >> (4 * (g_))) & 15) - (int)((tab >> (4 * (g2_))) & 15))
#endif
        #define KFL_EVENT(g_, g2_, k_) do { KFL_CAPUPD(g_, g2_, k_); cnt -= lane == (g_) - 1 ? (k_) : 0; cnt += lane == (g2_) - 1 ? (k_) : 0; nz = (uint32_t)kw::ballot(cnt > 0); } while (0)
        // the 64 jobs of a stretch: one per lane; the NEXT stretch's loads are issued before this stretch is walked
        int nx_flag = 0, nx_first = 0, nx_nt = 0, nx_ucls = 0;
        if (V > rp.start) { const int gc = rp.start + lane < V ? rp.start + lane : V - 1; nx_flag = b.g_flag[gc]; nx_first = b.g_first[gc]; nx_nt = b.g_nt[gc]; nx_ucls = b.g_ucls[gc]; }
        for (int base = rp.start; base < V && !mismatch; base += 64) {
            const int gi = base + lane;
            const bool valid = gi < V;
            const int my_flag = valid ? nx_flag : BF_GATE, my_first = valid ? nx_first : 0, my_nt = valid ? nx_nt : 0, my_ucls = valid ? nx_ucls : 0;
            { const int gc = gi + 64 < V ? gi + 64 : V - 1; nx_flag = b.g_flag[gc]; nx_first = b.g_first[gc]; nx_nt = b.g_nt[gc]; nx_ucls = b.g_ucls[gc]; }
            const int jn = V - base < 64 ? V - base : 64;
            int my_q = kw::shfl(qk, my_ucls >= 0 ? my_ucls : 0);  // devices the job's one class asks for; 0 = a gang of several classes
            if (my_ucls < 0 || my_q > 31) my_q = my_ucls < 0 ? 0 : 31;  // (a request beyond every level: no level, no capacity)
            const int my_cap = kw::shfl(capq, my_q >= 1 ? my_q - 1 : 63);  // (lane 63 asks for 64 devices: capacity 0)
            const int my_pack = my_flag | (my_q << 2) | (my_nt << 7);  // flag: 2 bits, devices: 5 bits, tasks: up to KB_PLACED_MAX — one readlane per job
            // dead for good: a gang of one class the plan predicted dead that the levels cannot hold now (the capacities only shrink).  It books cap + 1 decisions at its turn.
            const bool is_def = valid && my_flag == BF_DEAD && my_q >= 1 && my_cap < my_nt;
            const uint64_t defm = kw::ballot(is_def);
            uint64_t todo = kw::ballot(valid && my_flag != BF_GATE && !is_def);
            uint64_t failm = 0;  // lane q − 1: the stretch's dead-for-good jobs that ask for q devices
            if (defm) for (int q = 1; q <= v.LV; q++) { const uint64_t m = kw::ballot(is_def && my_q == q); if (lane == q - 1) failm = m; }
            uint64_t okm = 0;    // jobs of this stretch that committed
            int seg_lo = 0, n_out = jn;
            attempted += jn; n_done = base + jn;
            // the dead-for-good jobs [lo, hi) of the stretch book their decisions with the capacities as they are now
            #define KFL_BOOK(lo, hi) do { const uint64_t sm_ = ((hi) >= 64 ? ~0ull : (1ull << (hi)) - 1) & ~((1ull << (lo)) - 1); if (defm & sm_) dec_v += capq * __builtin_popcountll(failm & sm_); } while (0)
            KFL_T(0);
            while (todo) {
                const int jj = __builtin_ctzll(todo); todo &= todo - 1;
                KFL_BOOK(seg_lo, jj); seg_lo = jj + 1;
                const int pack = kw::bcast(my_pack, jj), first = kw::bcast(my_first, jj);
                const int flag = pack & 3, qc = (pack >> 2) & 31, nt = pack >> 7;
                bool ok = true;
                KFL_T(1);
                if (qc) {
                    // a gang of ONE class.  The usual one fits on its class's best node (the lowest non-empty level g >= q holds nt·q devices): one command, no capacity needed
                    const int g0 = KFL_LEVEL_FOR(qc), need0 = nt * qc;
                    if (g0 && need0 <= g0) {
                        KFL_ROOM(1);
                        KFL_EMIT(g0, g0 - need0, 1, nt, first);
                        KFL_EVENT(g0, g0 - need0, 1);
                        decisions += nt;
                    } else {
                        // it fits iff the levels hold enough places for it (a node of level g holds g / q of its tasks, every placement takes exactly one place away) — else it places
                        // `cap` tasks, finds no node for the next one and is rolled back: cap + 1 decisions, the state it started from
                        const int cap = kw::bcast(capq, qc - 1);
                        if (cap < nt) { decisions += cap + 1; ok = false; }
                        else {
                            const uint64_t tq = kw::bcast(tab, qc - 1);
                            int done = 0;
                            while (done < nt) {
                                KFL_ROOM(1);
                                const int g = KFL_LEVEL_FOR(qc), r = (int)((tq >> (4 * g)) & 15), rem = nt - done, cg = kw::bcast(cnt, g - 1);
                                int k = 1, per = rem;
                                if (rem >= r) { per = r; k = r == 1 ? rem : kfl_div(rem, r); if (k > cg) k = cg; }
                                const int g2 = g - per * qc;
                                KFL_EMIT(g, g2, k, per, first + done);
                                KFL_EVENT(g, g2, k);
                                done += k * per;
                            }
                            decisions += nt;
                        }
                    }
                } else {
                    // a gang of several scan classes: task by task on a copy of the counts; its commands stay unpublished until the last task has found its level
                    KFL_ROOM(KB_PLACED_MAX);
                    const int cnt_s = cnt, capq_s = capq, wp_s = wp; const uint32_t nz_s = nz;
                    for (int tb = 0; tb < nt && ok; tb += 64) {
                        const int my_cls = tb + lane < nt ? b.t_cls[first + tb + lane] : 0;
                        const int tc = nt - tb < 64 ? nt - tb : 64;
                        for (int ti = 0; ti < tc; ti++) {
                            const int q1 = kw::bcast(qk, kw::bcast(my_cls, ti));
                            decisions++;
                            const int g = q1 <= 31 ? KFL_LEVEL_FOR(q1) : 0;
                            if (!g) { ok = false; break; }
                            KFL_EMIT(g, g - q1, 1, 1, first + tb + ti);
                            KFL_EVENT(g, g - q1, 1);
                        }
                    }
                    if (!ok) { cnt = cnt_s; capq = capq_s; nz = nz_s; wp = wp_s; }  // Statement.Rollback: nothing was published
                }
                KFL_T(2);
                if (ok) okm |= 1ull << jj; else rollbacks += 2;
                if (wp - pub >= 64) { kw::lds_store_rel(&L.head, wp); pub = wp; }
                if ((flag == BF_OK) != ok) { mismatch = 1; n_done = base + jj + 1; n_out = jj + 1; attempted -= jn - (jj + 1); break; }
                KFL_T(3);
            }
            KFL_T(3);
            KFL_BOOK(seg_lo, n_out);
            #undef KFL_BOOK
            {   // the stretch's outcomes: what every job ended with, its Statement number and the offset of its operations among the round's (a ballot and a prefix sum)
                const uint64_t outm = n_out >= 64 ? ~0ull : (1ull << n_out) - 1;
                const int ndef = __builtin_popcountll(defm & outm);
                decisions += ndef; rollbacks += 2 * ndef;
                const bool my_ok = (okm >> lane) & 1ull;
                const int myv = my_ok ? my_nt : 0, incl = kw::wave_scan_add(myv);
                const int my_stmt = committed + rp.stmt0 + __builtin_popcountll(okm & ((1ull << lane) - 1)), my_opoff = ops + rp.ops0 + incl - myv;
                if (lane < n_out) { b.g_out[base + lane] = (uint8_t)(my_ok ? BF_OK : BF_DEAD); b.g_opoff[base + lane] = my_opoff; b.g_stmt[base + lane] = my_stmt; }
                committed += __builtin_popcountll(okm); ops += kw::bcast(incl, 63);
            }
            if (wp != pub) { kw::lds_store_rel(&L.head, wp); pub = wp; }
            KFL_T(4);
        }
        #undef KFL_LEVEL_FOR
        #undef KFL_ROOM
        #undef KFL_EMIT
        #undef KFL_EVENT
        kw::lds_store_rel(&L.head, wp);
        kw::lds_store_rel(&L.done, 1);
        for (int l = 0; l < v.LV; l++) decisions += kw::bcast(dec_v, l);
        const uint64_t dead = kw::ballot(act && (qk > 32 || (nz >> (qk - 1)) == 0));
        if (lane == 0) {
            FillStatus s; s.n_done = n_done; s.mismatch = mismatch; s.all_dead = (C > 0 && dead == (C >= 64 ? ~0ull : ((1ull << C) - 1))) ? 1 : 0; s.planned = V; s.floor_stop = 0; s.pad = 0;
            s.decisions = decisions; s.attempted = attempted; s.committed = committed; s.rollbacks = rollbacks; s.ops = ops; s.dead_mask = dead;
            s.cycles_total = kw::clock() - tstart; s.cycles_load = a_wait; s.cycles_update = 0; s.cycles_rescan = 0; s.block_loads = 0;  // (cycles_update / cycles_rescan / block_loads / rescans1: the workers' clocks, added below)
            s.rescans1 = 0; s.rescans2 = wp; s.rescans3 = 0;  // rescans2: commands (a command moves the first k nodes of a level)
#ifdef KFL_PROF
            s.cycles_load = pc[0]; s.cycles_update = pc[1]; s.cycles_rescan = pc[2]; s.block_loads = pc[3]; s.rescans1 = pc[4];  // stretch prologue / decode + book / the gang / its tail / stretch epilogue
#endif
            b.fs[0] = s; b.dead_mask[0] = dead;
        }
    } else if ((tid >> 6) <= v.LV) {
        // ------------------------------------------------------------------ wavefront G = 1 .. LV: the worker of level G.  Everything here is uniform over the wavefront.
        const int G = tid >> 6, lw = (G - 1) * v.NW, l1 = (G - 1) * v.NW1;
        uint64_t s2 = kw::ballot(lane < v.NW1 && v.s1[l1 + (lane < v.NW1 ? lane : 0)] != 0);  // second summary: bit j = the 64 words of group j hold a node
        int firstn = KB_INF, cw = -1; uint64_t curw = 0;  // the level's first node (lowest name rank), the word that holds it (index and current value)
        auto first_of_group = [&](int w1, uint64_t m1) { cw = w1 * 64 + __builtin_ctzll(m1); curw = kfl_read(&v.gw[lw + cw]); firstn = (cw << 6) + __builtin_ctzll(curw); };
        auto refill = [&]() { if (!s2) { firstn = KB_INF; cw = -1; curw = 0; return; } const int w1 = __builtin_ctzll(s2); first_of_group(w1, kfl_read(&v.s1[l1 + w1])); };
        refill();
        // lane t: this worker's hand-over ring to level t + 1 — entries written, the consumer's progress as last read; lane s: its ring from level s + 1 — entries taken
        int xp = 0, xseen = 0, xc = 0;
        int tail = 0; int64_t w_idle = 0; const int64_t w_start = kw::clock();
        for (;;) {
            const int head = kw::lds_load_acq(&L.head);
            if (tail == head) {
                if (kw::lds_load_acq(&L.done) && tail == kw::lds_load_acq(&L.head)) break;
                const int64_t i0 = kw::clock(); while (kw::lds_load_acq(&L.head) == tail && !kw::lds_load_acq(&L.done)) kw::relax(); w_idle += kw::clock() - i0;
                continue;
            }
            while (tail < head) {
                // a batch of up to 64 commands, one per lane: the ones that name this level are walked, the others cost nothing
                const int nb = head - tail < 64 ? head - tail : 64;
                const FcCmd mc = L.ring[(tail + lane) & (KFL_RING - 1)];  // (lanes beyond the batch read a slot that is not used)
                uint64_t mine = kw::ballot(lane < nb && ((mc.lv & 0xff) == G || (mc.lv >> 8) == G));
                while (mine) {
                const int ci = __builtin_ctzll(mine); mine &= mine - 1;
                const int clv = kw::bcast(mc.lv, ci);
                const int g = clv & 0xff, g2 = clv >> 8;
                if (g == G) {
                    // SOURCE: the level's first k nodes leave it, `per` tasks on each
                    const int per = kw::bcast(mc.per, ci); int left = kw::bcast(mc.k, ci), tb = kw::bcast(mc.tbase, ci);
                    const int pr = g2 >= 1 ? kfl_pair(G, g2) : 0;
                    while (left > 0) {
                        const int w = cw; const uint64_t word = curw;
                        int m = 1; uint64_t mask = word & (0 - word);
                        if (left > 1) {  // several nodes: the set bits of the first node's word, from it upwards
                            m = __builtin_popcountll(word); mask = word;
                            if (m > left) { m = left; mask = 0; uint64_t x = word; for (int j = 0; j < m; j++) { mask |= x & (0 - x); x &= x - 1; } }
                            for (int t0 = 0; t0 < m * per; t0 += 64) {  // task t of this word's share sits on the (t / per)-th node of the mask
                                const int t = t0 + lane;
                                if (t < m * per) { uint64_t mm = mask; for (int j = bk_div_small(t, per); j > 0; j--) mm &= mm - 1; b.t_node[tb + t] = (w << 6) + __builtin_ctzll(mm); }
                            }
                        } else if (lane < per) b.t_node[tb + lane] = firstn;  // one node (the usual command): its tasks all sit on it (per <= 16)
                        const uint64_t neww = word ^ mask;
                        kfl_write(&v.gw[lw + w], neww);
                        if (neww) { curw = neww; firstn = (w << 6) + __builtin_ctzll(neww); }
                        else {  // the word is empty: its bit in the first summary goes, and the level's first node is the first node of the next word
                            const int w1 = w >> 6; const uint64_t m1 = kfl_read(&v.s1[l1 + w1]) & ~(1ull << (w & 63));
                            kfl_write(&v.s1[l1 + w1], m1);
                            if (m1) first_of_group(w1, m1); else { s2 &= ~(1ull << w1); refill(); }
                        }
                        left -= m; tb += m * per;
                        if (g2 >= 1) {  // hand the nodes over to the worker of the target level
                            const int n_w = kw::bcast(xp, g2 - 1);
                            if (n_w - kw::bcast(xseen, g2 - 1) >= KFL_XR) { int t; while (n_w - (t = kw::lds_load_acq(&L.xtail[pr])) >= KFL_XR) kw::relax(); if (lane == g2 - 1) xseen = t; }
                            const int sl = n_w & (KFL_XR - 1);
                            if (lane == 0) { L.x[pr][sl].mask = mask; L.x[pr][sl].w = w | (left == 0 ? 1 << 30 : 0); }
                            kw::lds_store_rel(&L.x[pr][sl].seq, n_w + 1);
                            if (lane == g2 - 1) xp++;
                        }
                    }
                } else if (g2 == G) {
                    // TARGET: take the command's nodes in as the source level's worker hands them over
                    const int pr = kfl_pair(g, G);
                    for (bool last = false; !last;) {
                        const int n_r = kw::bcast(xc, g - 1);
                        const int sl = n_r & (KFL_XR - 1);
                        if (kw::lds_load_acq(&L.x[pr][sl].seq) != n_r + 1) { const int64_t i0 = kw::clock(); while (kw::lds_load_acq(&L.x[pr][sl].seq) != n_r + 1) kw::relax(); w_idle += kw::clock() - i0; }
                        const int wf = kw::bcast(L.x[pr][sl].w, 0), w = wf & 0x3fffffff; const uint64_t mask = kw::bcast(L.x[pr][sl].mask, 0);
                        last = (wf >> 30) & 1;
                        kw::lds_store_rel(&L.xtail[pr], n_r + 1);
                        if (lane == g - 1) xc++;
                        if (w == cw) { curw |= mask; kfl_write(&v.gw[lw + w], curw); firstn = (w << 6) + __builtin_ctzll(curw); }
                        else {
                            const uint64_t old = kfl_read(&v.gw[lw + w]), nw = old | mask;
                            kfl_write(&v.gw[lw + w], nw);
                            if (!old) { const int w1 = w >> 6; const uint64_t o1 = kfl_read(&v.s1[l1 + w1]); kfl_write(&v.s1[l1 + w1], o1 | (1ull << (w & 63))); if (!o1) s2 |= 1ull << w1; }
                            const int n = (w << 6) + __builtin_ctzll(mask);
                            if (n < firstn) { firstn = n; cw = w; curw = nw; }  // (every node the word held before lies at or above the old first node)
                        }
                    }
                }
                }
                tail += nb;
            }
            kw::lds_store_rel(&L.tail[G - 1], tail);
        }
        if (lane == 0) { L.w_idle[G - 1] = w_idle; L.w_total[G - 1] = kw::clock() - w_start; }
    }
    kw::sync();
#ifndef KFL_PROF
    if (tid == 0) {  // the workers' clocks: idle and total summed, the busiest worker's busy cycles and its level
        int64_t idle = 0, total = 0, busy = 0; int lvl = 0;
        for (int l = 0; l < v.LV && l < KFL_LMAX; l++) { idle += L.w_idle[l]; total += L.w_total[l]; const int64_t x = L.w_total[l] - L.w_idle[l]; if (x > busy) { busy = x; lvl = l + 1; } }
        b.fs[0].cycles_update = idle; b.fs[0].cycles_rescan = total; b.fs[0].block_loads = busy; b.fs[0].rescans1 = lvl;
    }
#endif
    for (int i = tid; i < v.LV * v.NW; i += T) b.bk_words[i] = v.gw[i];
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(64 * (KFL_LMAX + 1)) k_fill_levels(KaiCtx c, RoundParams rp, BucketParams bp) { kb_fill_levels(c, rp, bp); }
#endif

}  // namespace kai
