// kai_fill_levels.hpp — the fill of kai_fill_counts.hpp taken apart further: a counting machine that only decides, ONE WAVEFRONT PER LEVEL for the sets, a bookkeeper for the dead gangs.
//
// kai_fill_counts.hpp split the bucket fill into a counting machine (the planned order over cnt[g] = nodes with g free devices) and two set workers that execute its commands on the
// LDS-resident sets.  Measured on BASELINE config 5 (profiles/r05*, r06*): a single wavefront issues an instruction every 8 – 10 cycles whatever it is, the counting machine spent ≈ 135
// of them on every job that reached it — also on the 40 % of them the plan already predicted dead —, and the two workers were busy 65 – 90 % of the time: either chain bounded the
// kernel at about the same length.  This kernel cuts the work of the wavefront every other one waits for to what DECIDING needs, and gives the rest to wavefronts beside it:
//
//   * Wavefront 0, the counting machine.  Lane j of a stretch of 64 jobs holds job j.  A gang of one class that the plan predicted dead and that does not fit the capacities at the
//     stretch's start is dead for good (capacities only shrink during allocate) and takes no part in the walk.  A walked gang is placed OPTIMISTICALLY on the counts — the lowest
//     non-empty level >= q, whole nodes per step — and its commands are published when its last task has found a level; a gang that runs out of levels is rolled back (counts, mask
//     and ring position: three registers) having booked the tasks it placed + 1 decisions, exactly what the capacity rule of kai_fill_counts.hpp books.  So this wavefront keeps no
//     capacities at all.  Outcomes, Statement numbers and operation offsets of a stretch come out of one ballot and one prefix sum at its end.  A command is 8 bytes.
//   * Wavefronts 1 .. L, the set workers: level g belongs to wavefront g, alone.  Its words, its two summaries and its first node are that wavefront's uniform state (the word that
//     holds the first node is cached in registers: removing the level's first node — what every command does — reads nothing from LDS while that word lasts).  A worker looks at 64
//     commands at a time, one per lane, and walks the ones that name its level: as the SOURCE it removes the level's first k nodes, writes the tasks' nodes and hands (word, mask) to
//     the target level's worker through the ring of that (source, target) pair; as the TARGET it takes the nodes in when it reaches the command.  Every level sees its removals and
//     insertions in command order; nodes only move DOWN the levels, so the wait-for graph has no cycle (a producer waits only for a consumer that is behind it; the worker that is
//     furthest behind waits for nobody).
//   * Wavefront L + 1, the bookkeeper: what a dead gang books is cap + 1 decisions with the capacity cap[q] = Σ_g (g / q)·cnt[g] AT ITS TURN.  The bookkeeper replays the command stream
//     (lane q − 1 keeps cap[q]: two table look-ups and a multiply-add per command), finds a command's job by its task range, and books the dead gangs between two walked jobs with one
//     population count — off the chain the other wavefronts wait for.
//
// Results are identical to k_fill_counts / k_fill_buckets (and through them to the oracle): tests/test_batch_path.py and tests/test_gpu_parity.py run the fills against each other, the
// emulator runs this kernel's wavefronts as fibers that really interleave (KW_EMU_ORDER), tests/host_sim shadows every launch with the scalar C++ fill.  Clusters with more than
// eight levels (16-device nodes) or with static class bitmaps stay on k_fill_counts / k_fill_buckets.
#pragma once
#include "kai_fill_counts.hpp"

namespace kai {

constexpr int KFL_LMAX = 8;                                // levels = worker wavefronts
constexpr int KFL_RING = 4096;                             // commands the ring holds (a gang, <= KB_PLACED_MAX commands, is written in full before it is published)
constexpr int KFL_XR = 32;                                 // entries of a hand-over ring
constexpr int KFL_SHORT = 16;                              // a gang of one class with at most this many tasks is "short": the stretch reserves its commands' room in the ring up front
constexpr int KFL_PAIRS = KFL_LMAX * (KFL_LMAX - 1) / 2;   // (source level g, target level g2 < g)
// a command, 8 bytes: bits 0-3 g, 4-7 g2, 8-11 per, 12-22 k | the upper word: tbase — the first k nodes of level g take `per` tasks each and move to level g2 (0: no level); their tasks are t_node[tbase ..)
KW_BODY uint64_t kfl_cmd(int g, int g2, int k, int per, int tbase) { return (uint64_t)((uint32_t)g | ((uint32_t)g2 << 4) | ((uint32_t)per << 8) | ((uint32_t)k << 12)) | ((uint64_t)(uint32_t)tbase << 32); }
struct FlMove { uint64_t mask; int32_t w; int32_t seq; };  // the nodes `mask` of word w (bit 30 of w: the command's last entry); seq = the entry's number in its ring + 1, stored last (release)
struct FlLds {
    uint64_t ring[KFL_RING];
    uint64_t dummy[64];            // where the lanes other than lane 0 put their copy of a command (a store without a branch)
    FlMove x[KFL_PAIRS][KFL_XR];
    int32_t xtail[KFL_PAIRS];      // entries the pair's consumer has taken
    int32_t cnt0[KBK_GMAX];
    int32_t tail[KFL_LMAX + 1];    // commands worker g (index g − 1) / the bookkeeper (index KFL_LMAX) has passed
    int32_t head, done, fin, b_dec;  // fin: jobs of the planned order the counting machine executed (for the bookkeeper's last stretch); b_dec: the bookkeeper's decisions
    int64_t w_idle[KFL_LMAX + 1], w_total[KFL_LMAX + 1];  // (the clocks: profiling)
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
// one stretch of 64 planned jobs as the counting machine and the bookkeeper both see it: lane j holds job j
struct FlStretch { int flag, first, nt, q; bool valid, is_def; uint64_t defm, todo; };
// lane q − 1: the quotients g / q for g = 1 .. 8, four bits each (bits 4(g − 1) ..)
KW_BODY uint32_t kfl_tab(int lane, int LV) { uint32_t t = 0; for (int g = 1; g <= LV; g++) t |= (uint32_t)(g / (lane + 1)) << (4 * (g - 1)); return t; }
KW_BODY int kfl_quot(uint32_t tab, int g) { return g >= 1 ? (int)((tab >> (4 * (g - 1))) & 15u) : 0; }
// classifies a stretch from its jobs' parameters and the capacities at its start (lane q − 1: capq).  qk: lane k = devices class k asks for.
KW_BODY void kfl_classify(FlStretch& s, int ucls, int qk, int capq) {
    int q = kw::shfl(qk, ucls >= 0 ? ucls : 0);  // devices the job's one class asks for; 0 = a gang of several classes
    if (ucls < 0 || q > 31) q = ucls < 0 ? 0 : 31;  // (a request beyond every level: no level, no capacity)
    const int cap = kw::shfl(capq, q >= 1 ? q - 1 : 63);  // (lane 63 asks for 64 devices: capacity 0)
    s.q = q;
    // dead for good: a gang of one class the plan predicted dead that the levels cannot hold now (the capacities only shrink).  It books cap + 1 decisions at its turn.
    s.is_def = s.valid && s.flag == BF_DEAD && q >= 1 && cap < s.nt;
    s.defm = kw::ballot(s.is_def);
    s.todo = kw::ballot(s.valid && s.flag != BF_GATE && !s.is_def);
}

KW_BODY void kb_fill_levels(const KaiCtx& c, RoundParams rp, BucketParams bp) {
    if (kb_round_off(c.bt)) return;
    KW_SHARED FlLds L;
    const BatchCtx& b = c.bt;
    const int tid = kw::tid(), T = kw::bdim(), lane = kw::lane(), C = c.C;
    BkView v; v.NW = bp.nw; v.NW1 = bp.nw1; v.LV = bp.levels;
    unsigned char* dyn = kw::dyn_lds();
    v.gw = (KW_LDS_PTR(uint64_t))dyn; v.s1 = v.gw + (size_t)v.LV * v.NW; v.ok = v.s1 + (size_t)v.LV * v.NW1 + KBK_GMAX;
    const int64_t tstart = kw::clock();
    if (tid < KBK_GMAX) L.cnt0[tid] = 0;
    if (tid <= KFL_LMAX) { L.tail[tid] = 0; L.w_idle[tid] = 0; L.w_total[tid] = 0; }
    if (tid < KFL_PAIRS) L.xtail[tid] = 0;
    for (int i = tid; i < KFL_PAIRS * KFL_XR; i += T) L.x[i / KFL_XR][i % KFL_XR].seq = 0;
    if (tid == 0) { L.head = 0; L.done = 0; L.fin = 0; L.b_dec = 0; }
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
    const int wave = tid >> 6;
    if (wave == 0) {
        // ------------------------------------------------------------------ wavefront 0: the counting machine.
        // lane l: cnt = nodes of level l + 1.  lane q − 1: tab = the quotients g / q.  lane k: qk = devices class k asks for.
        int cnt = lane < v.LV ? L.cnt0[lane] : 0;
        const bool act = lane < C;
        const int qk = act ? (int)c.cls[lane].req[KAI_RES_GPU] : 0x7fffffff;
        const uint32_t tab = kfl_tab(lane, v.LV);
        uint32_t nz = (uint32_t)kw::ballot(cnt > 0);  // bit l: level l + 1 holds a node
        int decisions = 0, attempted = 0, committed = 0, rollbacks = 0, ops = 0, n_done = rp.start, mismatch = 0;
        int wp = 0, tail_seen = 0, pub = 0;  // commands written / the slowest reader's progress as last read / commands published (per 64 commands and at the end of a stretch)
        int64_t a_wait = 0;               // cycles this wavefront waited for room in the ring
#ifdef KFL_PROF
        int64_t pc[5] = {0, 0, 0, 0, 0}; int64_t pt = kw::clock();
        #define KFL_T(i) do { const int64_t n_ = kw::clock(); pc[i] += n_ - pt; pt = n_; } while (0)
#else
        #define KFL_T(i) (void)0
#endif
        // the slowest reader's progress: lanes 0 .. LV − 1 the workers', lane LV the bookkeeper's
        auto tails_min = [&]() { int t = lane <= v.LV ? kw::lds_load_acq(&L.tail[lane < v.LV ? lane : KFL_LMAX]) : 0x7fffffff; for (int l = 1; l <= v.LV; l++) { const int o = kw::bcast(t, l); t = o < t ? o : t; } return kw::bcast(t, 0); };
        // the lowest non-empty level >= qc, 0 = none
        #define KFL_LEVEL_FOR(qc) ((nz >> ((qc) - 1)) ? (qc) + __builtin_ctz(nz >> ((qc) - 1)) : 0)
        // room for n more commands in the ring (a gang stays unpublished until its last task has found a level)
        #define KFL_ROOM(n) do { if (wp - tail_seen > KFL_RING - (n)) { kw::lds_store_rel(&L.head, wp); pub = wp; const int64_t w0 = kw::clock(); while (wp - tail_seen > KFL_RING - (n)) { tail_seen = tails_min(); if (wp - tail_seen > KFL_RING - (n)) kw::relax(); } a_wait += kw::clock() - w0; } } while (0)
        // lane 0 writes the command, the other lanes a copy into a slot of their own: one store, no branch
        #define KFL_EMIT(g_, g2_, k_, per_, tb_) do { KW_LDS_PTR(uint64_t) sl_ = lane == 0 ? (KW_LDS_PTR(uint64_t))&L.ring[wp & (KFL_RING - 1)] : (KW_LDS_PTR(uint64_t))&L.dummy[lane]; *sl_ = kfl_cmd(g_, g2_, k_, per_, tb_); wp++; } while (0)
        // k nodes leave level g for level g2 (0: none): the counts and the non-empty mask
        #define KFL_EVENT(g_, g2_, k_) do { cnt -= lane == (g_) - 1 ? (k_) : 0; cnt += lane == (g2_) - 1 ? (k_) : 0; nz = (uint32_t)kw::ballot(cnt > 0); } while (0)
        // the 64 jobs of a stretch: one per lane; the NEXT stretch's loads are issued before this stretch is walked
        int nx_flag = 0, nx_first = 0, nx_nt = 0, nx_ucls = 0;
        if (V > rp.start) { const int gc = rp.start + lane < V ? rp.start + lane : V - 1; nx_flag = b.g_flag[gc]; nx_first = b.g_first[gc]; nx_nt = b.g_nt[gc]; nx_ucls = b.g_ucls[gc]; }
        for (int base = rp.start; base < V && !mismatch; base += 64) {
            const int gi = base + lane;
            FlStretch s; s.valid = gi < V;
            s.flag = s.valid ? nx_flag : (int)BF_GATE; s.first = s.valid ? nx_first : 0; s.nt = s.valid ? nx_nt : 0;
            const int my_ucls = s.valid ? nx_ucls : 0;
            { const int gc = gi + 64 < V ? gi + 64 : V - 1; nx_flag = b.g_flag[gc]; nx_first = b.g_first[gc]; nx_nt = b.g_nt[gc]; nx_ucls = b.g_ucls[gc]; }
            const int jn = V - base < 64 ? V - base : 64;
            int capq = 0;  // lane q − 1: the tasks that ask for q devices the levels hold at the stretch's start
            for (int g = 1; g <= v.LV; g++) capq += kfl_quot(tab, g) * kw::bcast(cnt, g - 1);
            kfl_classify(s, my_ucls, qk, capq);
            // a job's parameters in one word: flag (2 bits), devices (5 bits), tasks (up to KB_PLACED_MAX: 11 bits), bit 18: the long way (several classes, or more tasks than the
            // stretch's reservation in the ring covers)
            const int my_pack = s.flag | (s.q << 2) | (s.nt << 7) | ((s.q == 0 || s.nt > KFL_SHORT) ? 1 << 18 : 0);
            const uint32_t my_tq = kw::shfl(tab, s.q >= 1 ? s.q - 1 : 63);  // the quotients g / q of the job's request
            uint64_t todo = s.todo, okm = 0;  // okm: jobs of this stretch that committed
            int n_out = jn, last_jj = 0;
            attempted += jn; n_done = base + jn;
            KFL_ROOM(64 * KFL_SHORT);  // room for every short gang of the stretch: no check per gang
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(my_tq), "v"(my_pack) : "memory");  // the stretch's shuffles have landed: the walk below never waits on the LDS counter (its own command stores stay in flight)
#endif
            KFL_T(0);
            // (a conditional branch costs this wavefront ≈ 20 cycles whether it is taken or not, a scalar instruction ≈ 5 — tools/micro/issue_rate.hip: the walk below selects instead of
            // branching wherever both sides are a few instructions)
            while (todo) {
                const int jj = __builtin_ctzll(todo); todo &= todo - 1;
                const int pack = kw::bcast(my_pack, jj), first = kw::bcast(s.first, jj);
                const uint32_t tq = kw::bcast(my_tq, jj);
                const int flag = pack & 3, qc = (pack >> 2) & 31, nt = (pack >> 7) & 0x7ff;
                KFL_T(1);
                // the gang on the counts; its commands stay unpublished until its last task has found a level
                const int cnt_s = cnt, wp_s = wp; const uint32_t nz_s = nz;
                int placed = 0; bool fail = false;
                if (pack >> 18) {
                    KFL_ROOM(nt + 64 * KFL_SHORT);
                    if (!qc) {
                        // a gang of several scan classes: task by task
                        for (int tb = 0; tb < nt && !fail; tb += 64) {
                            const int my_cls = tb + lane < nt ? b.t_cls[first + tb + lane] : 0;
                            const int tc = nt - tb < 64 ? nt - tb : 64;
                            for (int ti = 0; ti < tc; ti++) {
                                const int q1 = kw::bcast(qk, kw::bcast(my_cls, ti));
                                const int g = q1 <= 31 ? KFL_LEVEL_FOR(q1) : 0;
                                if (!g) { fail = true; break; }
                                KFL_EMIT(g, g - q1, 1, 1, first + placed);
                                KFL_EVENT(g, g - q1, 1);
                                placed++;
                            }
                        }
                    }
                }
                // a gang of ONE class: whole nodes per step.  The lowest non-empty level g >= q holds r = g / q of its tasks per node; the first k nodes of it take r tasks each and move to
                // level g mod q, a remainder of fewer than r tasks goes to one node, which then stays at level g − rem·q.  No level: the step moves nothing and the gang has failed.
                while ((placed < nt) & !fail & (qc != 0)) {
                    const int g = KFL_LEVEL_FOR(qc);
                    fail = g == 0;
                    int r = kfl_quot(tq, g); r = r > 1 ? r : 1;
                    const int rem = nt - placed;
                    int kq = rem; if (r != 1) kq = kfl_div(rem, r);               // whole nodes the rest of the gang fills (0: it is a remainder of fewer than r tasks, on one node)
                    const int cg = kw::bcast(cnt, (g - 1) & 63);
                    int k = kq < cg ? kq : cg; k = k > 1 ? k : 1; k = g ? k : 0;  // min(kq, nodes of the level), at least the one node; no level: nothing moves
                    const int per = r < rem ? r : rem, g2 = g - per * qc;
                    KFL_EMIT(g, g2, k, per, first + placed);  // (a failed step writes a slot that stays unpublished: whatever its fields hold)
                    KFL_EVENT(g, g2, k);
                    placed += k * per;
                }
                KFL_T(2);
                const bool ok = !fail;
                okm |= (uint64_t)ok << jj;
                decisions += placed + (int)fail;  // every task placed is a decision (a gang that fits places them all); a gang that found no node for its next task booked that one too
                rollbacks += 2 * (int)fail;
                cnt = ok ? cnt : cnt_s; nz = ok ? nz : nz_s; wp = ok ? wp : wp_s;  // Statement.Rollback: nothing was published
                if (wp - pub >= 64) { kw::lds_store_rel(&L.head, wp); pub = wp; }
                const bool mism = (flag == BF_OK) != ok;  // the job ended differently from its prediction: it is the round's last
                mismatch |= (int)mism; last_jj = jj; todo = mism ? 0 : todo;
                KFL_T(3);
            }
            if (mismatch) { n_done = base + last_jj + 1; n_out = last_jj + 1; attempted -= jn - n_out; }
            KFL_T(3);
            {   // the stretch's outcomes: what every job ended with, its Statement number and the offset of its operations among the round's (a ballot and a prefix sum)
                const uint64_t outm = n_out >= 64 ? ~0ull : (1ull << n_out) - 1;
                const int ndef = __builtin_popcountll(s.defm & outm);  // (the capacities at their turns: the bookkeeper's part)
                decisions += ndef; rollbacks += 2 * ndef;
                const bool my_ok = (okm >> lane) & 1ull;
                const int myv = my_ok ? s.nt : 0, incl = kw::wave_scan_add(myv);
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
        if (lane == 0) L.fin = n_done;
        kw::lds_store_rel(&L.head, wp);
        kw::lds_store_rel(&L.done, 1);
        const uint64_t dead = kw::ballot(act && (qk > 32 || (nz >> (qk - 1)) == 0));
        if (lane == 0) {
            FillStatus s; s.n_done = n_done; s.mismatch = mismatch; s.all_dead = (C > 0 && dead == (C >= 64 ? ~0ull : ((1ull << C) - 1))) ? 1 : 0; s.planned = V; s.floor_stop = 0; s.pad = 0;
            s.decisions = decisions; s.attempted = attempted; s.committed = committed; s.rollbacks = rollbacks; s.ops = ops; s.dead_mask = dead;  // (decisions: the bookkeeper's part is added below)
            s.cycles_total = kw::clock() - tstart; s.cycles_load = a_wait; s.cycles_update = 0; s.cycles_rescan = 0; s.block_loads = 0;  // (cycles_update / cycles_rescan / block_loads / rescans1: the workers' clocks, added below)
            s.rescans1 = 0; s.rescans2 = wp; s.rescans3 = 0;  // rescans2: commands (a command moves the first k nodes of a level)
#ifdef KFL_PROF
            s.cycles_load = pc[0]; s.cycles_update = pc[1]; s.cycles_rescan = pc[2]; s.block_loads = pc[3]; s.rescans1 = pc[4];  // stretch prologue / decode / the gang / its tail / stretch epilogue
#endif
            b.fs[0] = s; b.dead_mask[0] = dead;
        }
    } else if (wave <= v.LV) {
        // ------------------------------------------------------------------ wavefront G = 1 .. LV: the worker of level G.  Everything here is uniform over the wavefront.
        const int G = wave, lw = (G - 1) * v.NW, l1 = (G - 1) * v.NW1;
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
                const uint64_t mc = L.ring[(tail + lane) & (KFL_RING - 1)];  // (lanes beyond the batch read a slot that is not used)
                uint64_t mine = kw::ballot(lane < nb && ((int)(mc & 15) == G || (int)((mc >> 4) & 15) == G));
                while (mine) {
                    const int ci = __builtin_ctzll(mine); mine &= mine - 1;
                    const int ca = kw::bcast((int)(uint32_t)mc, ci);
                    const int g = ca & 15, g2 = (ca >> 4) & 15;
                    if (g == G) {
                        // SOURCE: the level's first k nodes leave it, `per` tasks on each
                        const int per = (ca >> 8) & 15; int left = (ca >> 12) & 0x7ff, tb = kw::bcast((int)(uint32_t)(mc >> 32), ci);
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
                            } else if (lane < per) b.t_node[tb + lane] = firstn;  // one node (the usual command): its tasks all sit on it (per <= 8)
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
                                kw::lds_store_ordered(&L.x[pr][sl].seq, n_w + 1);
                                if (lane == g2 - 1) xp++;
                            }
                        }
                    } else {
                        // TARGET: take the command's nodes in as the source level's worker hands them over
                        const int pr = kfl_pair(g, G);
                        for (bool last = false; !last;) {
                            const int n_r = kw::bcast(xc, g - 1);
                            const int sl = n_r & (KFL_XR - 1);
                            if (kw::lds_load_acq(&L.x[pr][sl].seq) != n_r + 1) { const int64_t i0 = kw::clock(); while (kw::lds_load_acq(&L.x[pr][sl].seq) != n_r + 1) kw::relax(); w_idle += kw::clock() - i0; }
                            const int wf = kw::bcast(L.x[pr][sl].w, 0), w = wf & 0x3fffffff; const uint64_t mask = kw::bcast(L.x[pr][sl].mask, 0);
                            last = (wf >> 30) & 1;
                            kw::lds_store_ordered(&L.xtail[pr], n_r + 1);
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
    } else if (wave == v.LV + 1) {
        // ------------------------------------------------------------------ wavefront LV + 1: the bookkeeper.  It replays the commands on the capacities (lane q − 1: capq) and books,
        // for every dead-for-good gang, the capacity of its request size at its turn.  A command's job is the walked job whose task range holds the command's tbase.
        const int qk = lane < C ? (int)c.cls[lane].req[KAI_RES_GPU] : 0x7fffffff;
        const uint32_t tab = kfl_tab(lane, v.LV);
        int capq = 0;
        { const int cnt = lane < v.LV ? L.cnt0[lane] : 0; for (int g = 1; g <= v.LV; g++) capq += kfl_quot(tab, g) * kw::bcast(cnt, g - 1); }
        int dec_v = 0;                     // lane q − 1: Σ of the capacities the dead gangs that ask for q devices met at their turns
        int base = rp.start, seg_lo = 0;   // the stretch the bookkeeper is in; its jobs below seg_lo are booked
        FlStretch s; s.valid = false; s.flag = BF_GATE; s.first = 0; s.nt = 0; s.q = 0; s.is_def = false; s.defm = 0; s.todo = 0;
        uint64_t failm = 0;                // lane q − 1: the stretch's dead-for-good jobs that ask for q devices
        auto enter = [&]() {
            const int gi = base + lane; s.valid = gi < V; const int gc = s.valid ? gi : (V > 0 ? V - 1 : 0);
            s.flag = s.valid ? (int)b.g_flag[gc] : (int)BF_GATE; s.first = s.valid ? b.g_first[gc] : 0; s.nt = s.valid ? b.g_nt[gc] : 0;
            kfl_classify(s, s.valid ? b.g_ucls[gc] : 0, qk, capq);
            failm = 0; if (s.defm) for (int q = 1; q <= v.LV; q++) { const uint64_t m = kw::ballot(s.is_def && s.q == q); if (lane == q - 1) failm = m; }
            seg_lo = 0;
        };
        // the dead-for-good jobs [seg_lo, hi) of the stretch book their decisions with the capacities as they are now
        auto book = [&](int hi) { if (hi > seg_lo) { const uint64_t sm = (hi >= 64 ? ~0ull : (1ull << hi) - 1) & ~((1ull << seg_lo) - 1); if (s.defm & sm) dec_v += capq * __builtin_popcountll(failm & sm); seg_lo = hi; } };
        if (V > rp.start) enter();
        int tail = 0; int64_t w_idle = 0; const int64_t w_start = kw::clock();
        for (;;) {
            const int head = kw::lds_load_acq(&L.head);
            if (tail == head) {
                if (kw::lds_load_acq(&L.done) && tail == kw::lds_load_acq(&L.head)) break;
                const int64_t i0 = kw::clock(); while (kw::lds_load_acq(&L.head) == tail && !kw::lds_load_acq(&L.done)) kw::relax(); w_idle += kw::clock() - i0;
                continue;
            }
            while (tail < head) {
                const int nb = head - tail < 64 ? head - tail : 64;
                const uint64_t mc = L.ring[(tail + lane) & (KFL_RING - 1)];
                for (int ci = 0; ci < nb; ci++) {
                    const int ca = kw::bcast((int)(uint32_t)mc, ci), tb = kw::bcast((int)(uint32_t)(mc >> 32), ci);
                    const int g = ca & 15, g2 = (ca >> 4) & 15, k = (ca >> 12) & 0x7ff;
                    uint64_t hit;
                    while (!(hit = kw::ballot(((s.todo >> lane) & 1ull) && tb >= s.first && tb < s.first + s.nt)) && base + 64 < V) { book(64); base += 64; enter(); }  // the command's job lies in a later stretch
                    {   // the dead gangs in front of the command's job met the capacities before it (no branch: an empty segment books nothing)
                        const int hi = hit ? __builtin_ctzll(hit) : seg_lo;
                        const uint64_t sm = ((1ull << hi) - 1) & ~((1ull << seg_lo) - 1);
                        dec_v += capq * __builtin_popcountll(failm & sm);
                        seg_lo = hi > seg_lo ? hi : seg_lo;
                    }
                    capq -= k * (kfl_quot(tab, g) - kfl_quot(tab, g2));
                }
                tail += nb;
            }
            kw::lds_store_rel(&L.tail[KFL_LMAX], tail);
        }
        // the stretches behind the last command, up to the last job the counting machine executed
        const int fin = kw::lds_load_acq(&L.fin);
        while (base < fin) { const int hi = fin - base < 64 ? fin - base : 64; book(hi); if (base + 64 >= fin) break; base += 64; enter(); }
        int total = 0; for (int l = 0; l < v.LV; l++) total += kw::bcast(dec_v, l);
        if (lane == 0) { L.b_dec = total; L.w_idle[KFL_LMAX] = w_idle; L.w_total[KFL_LMAX] = kw::clock() - w_start; }
    }
    kw::sync();
    if (tid == 0) {
        b.fs[0].decisions += L.b_dec;
#ifndef KFL_PROF
        // the readers' clocks: idle and total summed, the busiest one's busy cycles and its level (9 = the bookkeeper)
        int64_t idle = 0, total = 0, busy = 0; int lvl = 0;
        for (int l = 0; l <= KFL_LMAX; l++) { if (l >= v.LV && l < KFL_LMAX) continue; idle += L.w_idle[l]; total += L.w_total[l]; const int64_t x = L.w_total[l] - L.w_idle[l]; if (x > busy) { busy = x; lvl = l + 1; } }
        b.fs[0].cycles_update = idle; b.fs[0].cycles_rescan = total; b.fs[0].block_loads = busy; b.fs[0].rescans1 = lvl;
#endif
    }
    for (int i = tid; i < v.LV * v.NW; i += T) b.bk_words[i] = v.gw[i];
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(64 * (KFL_LMAX + 2)) k_fill_levels(KaiCtx c, RoundParams rp, BucketParams bp) { kb_fill_levels(c, rp, bp); }
#endif

}  // namespace kai
