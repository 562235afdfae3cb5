// native_bucket_fill.hpp — TEST INFRASTRUCTURE (tests/host_sim only; the product library never sees it).
//
// The bucket fill of the batch path (kai-scheduler_amd/csrc/kai_fill_buckets.hpp: sets of nodes by free devices with two summary levels, the planned order walked gang by gang,
// whole nodes of a one-class gang per step, dead gangs decided from the levels' populations) written as what it is on one CPU core: plain scalar C++ over the same arrays —
// no lanes, no fibers, no ballots.  It exists to answer one question honestly ("cpu_same_algorithm"): how fast is THIS algorithm on one host core, next to the
// one wavefront that runs it on the MI355X?  tests/host_sim runs it as a shadow of every emulated k_fill_buckets launch (KAI_HOSTSIM_NATIVE_FILL=1): same inputs, its own copy
// of the sets, every output compared with the emulated kernel's (outcomes, operation offsets, the tasks' nodes, the counters, the sets afterwards), its time summed up.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

namespace kai_native {

using namespace kai;

struct NativeFillOut {
    std::vector<uint64_t> words;                 // the sets after the launch [levels][nw]
    std::vector<uint8_t> g_out; std::vector<int32_t> g_opoff, g_stmt;  // per planned job from rp.start on
    std::vector<int32_t> t_node;                 // [P] nodes of the tasks of committed jobs (others untouched: -1)
    FillStatus fs{};
    double ms = 0;
};

struct Sets {
    int LV, NW, NW1; std::vector<uint64_t> gw, s1, s2; std::vector<int32_t> cnt; const uint64_t* ok_home; int ok_stride;
    uint64_t ok_word(int cls_slot_class, int w) const { return cls_slot_class < 0 ? ~0ull : ok_home[(size_t)cls_slot_class * ok_stride + w]; }
    // first node (lowest name rank) of the lowest level >= from_g that class okc (-1: no static bitmap) may use
    void find(int okc, int from_g, int& og, int& on) const {
        for (int g = from_g < 1 ? 1 : from_g; g <= LV; g++) {
            uint64_t m2 = s2[g - 1];
            while (m2) {
                const int w1 = __builtin_ctzll(m2); m2 &= m2 - 1;
                uint64_t m1 = s1[(size_t)(g - 1) * NW1 + w1];
                while (m1) {
                    const int w = w1 * 64 + __builtin_ctzll(m1); m1 &= m1 - 1;
                    uint64_t word = gw[(size_t)(g - 1) * NW + w] & ok_word(okc, w);
                    if (word) { og = g; on = w * 64 + __builtin_ctzll(word); return; }
                }
            }
        }
        og = 0; on = -1;
    }
    void toggle(int g, int w, uint64_t bits, bool leaving) {  // level g (1-based; 0 = no level)
        if (g < 1) return;
        uint64_t& word = gw[(size_t)(g - 1) * NW + w]; const uint64_t old = word; word ^= bits;
        cnt[g - 1] += leaving ? -__builtin_popcountll(bits) : __builtin_popcountll(bits);
        if (leaving ? word == 0 : old == 0) {
            uint64_t& a = s1[(size_t)(g - 1) * NW1 + (w >> 6)]; const uint64_t o1 = a; a ^= 1ull << (w & 63);
            if (leaving ? a == 0 : o1 == 0) s2[g - 1] ^= 1ull << (w >> 6);
        }
    }
    uint64_t move_mask(int w, uint64_t bits, int from, int to) { toggle(from, w, bits, true); toggle(to, w, bits, false); return gw[(size_t)(from - 1) * NW + w]; }
};

inline uint32_t key_of(int g, int n) { return n < 0 ? 0xffffffffu : ((uint32_t)g << 20) | (uint32_t)n; }

inline void native_fill_buckets(const KaiCtx& c, RoundParams rp, const BucketParams& bp, NativeFillOut& out) {
    const BatchCtx& b = c.bt; const int C = c.C;
    Sets S; S.LV = bp.levels; S.NW = bp.nw; S.NW1 = bp.nw1; S.ok_home = b.bk_ok; S.ok_stride = bp.nw;
    S.gw.assign(b.bk_words, b.bk_words + (size_t)S.LV * S.NW);
    out.t_node.assign((size_t)c.P, -1);
    const auto t0 = std::chrono::steady_clock::now();
    S.s1.assign((size_t)S.LV * S.NW1, 0); S.s2.assign(S.LV, 0); S.cnt.assign(S.LV, 0);
    for (int l = 0; l < S.LV; l++) for (int w = 0; w < S.NW; w++) { const uint64_t x = S.gw[(size_t)l * S.NW + w]; if (x) { S.s1[(size_t)l * S.NW1 + (w >> 6)] |= 1ull << (w & 63); S.s2[l] |= 1ull << (w >> 6); S.cnt[l] += __builtin_popcountll(x); } }
    int q[64], okc[64]; uint32_t top[64]; bool plain = true;
    for (int k = 0; k < C; k++) { q[k] = (int)c.cls[k].req[KAI_RES_GPU]; okc[k] = bp.okslot[k] >= 0 ? k : -1; if (okc[k] >= 0) plain = false; int g, n; S.find(okc[k], q[k], g, n); top[k] = key_of(g, n); }
    const bool batched = rp.pad2 == 0;
    const int V = rp.mode != 1 ? b.q_valid[c.Q] : 0;
    int64_t decisions = 0, attempted = 0, committed = 0, rollbacks = 0, ops = 0; int n_done = rp.start, mismatch = 0;
    const int n_jobs = V > rp.start ? V - rp.start : 0;
    out.g_out.assign(n_jobs, 0); out.g_opoff.assign(n_jobs, 0); out.g_stmt.assign(n_jobs, 0);
    std::vector<int32_t> placed_node(KB_PLACED_MAX), placed_info(KB_PLACED_MAX);
    // after a node (or the nodes of `mask`) moved from level g to g2: every class patches its best in O(1); a class whose best stopped fitting looks the next one up
    auto patch_tops = [&](int n, int g, int g2, uint64_t rest, bool check_ok) {
        const uint32_t tk_from = key_of(g, n), cand = ((uint32_t)g2 << 20) | (uint32_t)n; uint64_t need = 0;
        for (int k = 0; k < C; k++) {
            const bool mine = top[k] == tk_from, fits2 = g2 >= q[k];
            bool okn = true; if (check_ok && okc[k] >= 0) okn = (S.ok_word(okc[k], n >> 6) >> (n & 63)) & 1ull;
            if (mine && !fits2) need |= 1ull << k;
            if (fits2 && (mine || (okn && cand < top[k]))) top[k] = cand;
        }
        if (!need) return;
        if (plain) {
            uint32_t fk;
            if (rest) fk = ((uint32_t)g << 20) | (uint32_t)((n & ~63) + __builtin_ctzll(rest));
            else { int fg, fn; S.find(-1, g, fg, fn); fk = key_of(fg, fn); }
            for (int k = 0; k < C; k++) if ((need >> k) & 1) top[k] = fk;
        } else for (int k = 0; k < C; k++) if ((need >> k) & 1) { int fg, fn; S.find(okc[k], g, fg, fn); top[k] = key_of(fg, fn); }
    };
    for (int gi = rp.start; gi < V && !mismatch; gi++) {
        const int flag = b.g_flag[gi], first = b.g_first[gi], nt = b.g_nt[gi], ucls = b.g_ucls[gi];
        const int opoff = (int)ops + rp.ops0, stmtoff = (int)committed + rp.stmt0;
        bool ok = flag != BF_GATE; int placed = 0;
        if (flag != BF_GATE) {
            if (ucls >= 0 && plain && batched) {
                const int qc = q[ucls];
                if (nt > 1 && flag == BF_DEAD) {  // does the gang fit at all?  Σ_levels (g / q) · nodes(g)
                    int64_t cap = 0; for (int g = qc; g <= S.LV; g++) cap += (int64_t)(g / qc) * S.cnt[g - 1];
                    if (cap < nt) { decisions += cap + 1; ok = false; }
                }
                int done = 0;
                while (ok && done < nt) {
                    const uint32_t tk = top[ucls];
                    if (tk == 0xffffffffu) { decisions++; ok = false; break; }
                    const int n = (int)(tk & 0xfffffu), g = (int)(tk >> 20), r = g / qc, w = n >> 6;
                    // (the kernel walks a gang in stretches of 64 tasks: a step never hands out tasks beyond the stretch it is in)
                    const int stretch_left = 64 - (done & 63), rem = (nt - done) < stretch_left ? (nt - done) : stretch_left;
                    int k = 1; const int per = rem < r ? rem : r; uint64_t mask = 1ull << (n & 63);
                    if (rem >= 2 * r) { uint64_t word = S.gw[(size_t)(g - 1) * S.NW + w]; const int want = rem / r; mask = 0; for (k = 0; k < want && word; k++) { mask |= word & (0 - word); word &= word - 1; } }
                    const int g2 = g - per * qc;
                    { uint64_t m = mask; int t = done; for (int j = 0; j < k; j++) { const int nj = (w << 6) + __builtin_ctzll(m); m &= m - 1; for (int x = 0; x < per; x++, t++) { placed_node[t] = nj; placed_info[t] = ucls | ((g - x * qc) << 8); } } }
                    decisions += (int64_t)k * per; done += k * per;
                    const uint64_t rest = S.move_mask(w, mask, g, g2);
                    patch_tops(n, g, g2, rest, false);
                }
                placed = done;
            } else {
                for (int ti = 0; ti < nt; ti++) {
                    const int kcls = ucls >= 0 ? ucls : b.t_cls[first + ti];
                    decisions++;
                    const uint32_t tk = top[kcls];
                    if (tk == 0xffffffffu) { ok = false; break; }
                    const int n = (int)(tk & 0xfffffu), g = (int)(tk >> 20), g2 = g - q[kcls];
                    placed_node[placed] = n; placed_info[placed] = kcls | (g << 8); placed++;
                    const uint64_t rest = S.move_mask(n >> 6, 1ull << (n & 63), g, g2);
                    patch_tops(n, g, g2, rest, !plain);
                }
            }
            if (!ok) {  // Statement.Rollback: the undone operations in reverse order, then every class's best from the restored sets
                for (int i = placed - 1; i >= 0; i--) { const int n = placed_node[i], info = placed_info[i], gb = info >> 8; S.move_mask(n >> 6, 1ull << (n & 63), gb - q[info & 0xff], gb); }
                if (placed) for (int k = 0; k < C; k++) { int g, n; S.find(okc[k], q[k], g, n); top[k] = key_of(g, n); }
                rollbacks += 2;
            } else { committed++; ops += nt; for (int i = 0; i < nt; i++) out.t_node[(size_t)first + i] = placed_node[i]; }
        }
        attempted++; n_done = gi + 1;
        out.g_out[gi - rp.start] = ok ? BF_OK : BF_DEAD; out.g_opoff[gi - rp.start] = opoff; out.g_stmt[gi - rp.start] = stmtoff;
        if ((flag == BF_OK) != ok) mismatch = 1;
    }
    uint64_t dead = 0; for (int k = 0; k < C; k++) if (top[k] == 0xffffffffu) dead |= 1ull << k;
    out.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    FillStatus s{}; s.n_done = n_done; s.mismatch = mismatch; s.all_dead = (C > 0 && dead == (C >= 64 ? ~0ull : ((1ull << C) - 1))) ? 1 : 0; s.planned = V;
    s.decisions = decisions; s.attempted = attempted; s.committed = committed; s.rollbacks = rollbacks; s.ops = ops; s.dead_mask = dead;
    out.fs = s; out.words = S.gw;
}

}  // namespace kai_native
