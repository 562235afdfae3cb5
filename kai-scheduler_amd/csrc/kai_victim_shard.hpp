// kai_victim_shard.hpp — the victim search's simulation waves dealt out over the GPUs of a node-sharded group ("reclaim scenario search shards the same
// way").
//
// One GPU already runs the simulations of a partial job on several engines (kai_engine_solver.inc solve_partial_multi: every engine the same control flow on its own
// replica, a wave's simulations handed out in the reference's order, outcomes in MultiCtx, every engine counts the wave the same way and runs the winning
// simulation itself).  A group of XR GPUs runs the SAME protocol with XR times the engines: every rank holds the whole session (as for every action the group does not
// shard by nodes), simulation i of a wave belongs to rank i mod XR, and at the wave's end — between the two grid barriers of solve_partial_multi — the ranks
// all-gather what their engines found: per simulation the status word and the eight counter deltas (68 bytes), per rank its lowest simulation that did not simply
// fail and how many it handed out.  After the merge MultiCtx::res / cnt / hit / xrun of the wave's buffer are identical on every rank, so every engine of every rank
// counts the wave to the same simulation, leaves the same nodes feasible and replays the same winner: no rank ever reads another rank's session state.
//
// The exchange is the HOST's: on the device engine 0 rings a mailbox in pinned host memory (XMail) and waits; the host thread inside kai_action_execute copies the
// wave's buffer out of HBM, runs the group's all-gather (the caller's collective on host memory — kai_shard_attach_host — or the library's own RCCL communicator on a
// second stream), merges, copies the merged buffer back and answers.  On the emulator (tests/host_sim) engine 0 calls the same pack / merge code directly.
//
// Faults: a rank whose engines left the protocol says so in its next message (`done`); the others then give up as well, and nobody issues a collective after a message
// with `done` was seen — every rank of the group ends a victim action with exactly one such message, so the number of collectives is the same everywhere.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "kai_engine.hpp"

namespace kai {

constexpr int32_t XW_MAGIC = 0x4b584d57;
struct XWaveHdr { int32_t magic, seq, done, fault, hit, next, n_own, pad; };  // hit: lowest own simulation that did not simply fail (INT_MAX: none); next: own simulations handed out

inline int xw_cap(int xcap) { return (xcap > 0 && xcap < KAI_MW_WAVE) ? xcap : KAI_MW_WAVE; }
inline int xw_own_max(int cap, int R) { return (cap + R - 1) / R; }
inline size_t xw_msg_bytes(int cap, int R) { return sizeof(XWaveHdr) + (size_t)xw_own_max(cap, R) * (sizeof(int64_t) * KAI_MW_CNT + sizeof(int64_t)); }  // (status words padded to 8 bytes each: one layout)
// default length of a wave on a group: two simulations per engine of the group — a rank learns of another rank's hit only at the wave's end, so what a wave runs beyond
// its first hit is bounded by its length, and a wave without a hit simply continues in the next one
inline int xw_default_cap(int R, int G) { return std::max(R, std::min((int)KAI_MW_WAVE, 2 * R * std::max(G, 1))); }

// this rank's part of wave buffer b: res_b[i], cnt_b[i] of the simulations i = l * R + r it handed out
inline void xw_pack(unsigned char* msg, int seq, int done, int fault, int hit, int next, const int32_t* res_b, const int64_t* cnt_b /* [KAI_MW_WAVE][KAI_MW_CNT] */, int cap, int R, int r) {
    XWaveHdr h{}; h.magic = XW_MAGIC; h.seq = seq; h.done = done; h.fault = fault; h.hit = hit; h.next = next;
    const int own_max = xw_own_max(cap, R);
    int n = next < own_max ? next : own_max; while (n > 0 && (n - 1) * R + r >= cap) n--;
    h.n_own = done ? 0 : n;
    std::memcpy(msg, &h, sizeof h);
    int64_t* c = reinterpret_cast<int64_t*>(msg + sizeof h); int64_t* s = c + (size_t)own_max * KAI_MW_CNT;
    for (int l = 0; l < h.n_own; l++) { const int i = l * R + r; std::memcpy(c + (size_t)l * KAI_MW_CNT, cnt_b + (size_t)i * KAI_MW_CNT, sizeof(int64_t) * KAI_MW_CNT); s[l] = res_b[i]; }
}
struct XWaveMerged { int32_t hit, xrun, fault, done; };
// the group's wave from the R gathered messages (rank-major): res_b / cnt_b get every rank's simulations; false = a message that is not of this exchange
inline bool xw_merge(const unsigned char* msgs, int seq, int cap, int R, int32_t* res_b, int64_t* cnt_b, XWaveMerged& out) {
    const size_t mb = xw_msg_bytes(cap, R); const int own_max = xw_own_max(cap, R);
    out.hit = 0x7fffffff; out.xrun = cap; out.fault = 0; out.done = 0;
    for (int rr = 0; rr < R; rr++) {
        const unsigned char* m = msgs + (size_t)rr * mb; XWaveHdr h; std::memcpy(&h, m, sizeof h);
        if (h.magic != XW_MAGIC || h.seq != seq || h.n_own < 0 || h.n_own > own_max) return false;
        if (h.done) out.done = 1;
        if (h.fault) out.fault = 1;
        if (h.done) continue;
        if (h.hit < out.hit) out.hit = h.hit;
        const int64_t first_not_run = (int64_t)h.next * R + rr;  // (a rank runs every simulation it hands out below the cap)
        if (first_not_run < out.xrun) out.xrun = (int32_t)first_not_run;
        const int64_t* c = reinterpret_cast<const int64_t*>(m + sizeof h); const int64_t* s = c + (size_t)own_max * KAI_MW_CNT;
        for (int l = 0; l < h.n_own; l++) { const int i = l * R + rr; if (i >= cap) break; std::memcpy(cnt_b + (size_t)i * KAI_MW_CNT, c + (size_t)l * KAI_MW_CNT, sizeof(int64_t) * KAI_MW_CNT); res_b[i] = (int32_t)s[l]; }
    }
    return true;
}

// host side of one exchange, written against what the caller can do with the device: Io::pull(hdr[8], res_b, cnt_b, b, cap) reads MultiCtx (world, fault, bar_count,
// bar_gen, next[2], hit[2] and buffer b's outcomes), Io::push(b, res_b, cnt_b, cap, hit, xrun, fault) writes the merged wave back, Io::allgather(send, recv, bytes).
struct XShardHost {
    int R = 1, r = 0, cap = KAI_MW_WAVE, seq = 0; bool seen_done = false; int64_t exchanges = 0;
    std::vector<int32_t> res; std::vector<int64_t> cnt; std::vector<unsigned char> send, recv;
    void begin(int world, int rank, int xcap) {
        R = world; r = rank; cap = xw_cap(xcap); seq = 0; seen_done = false; exchanges = 0;
        res.assign(KAI_MW_WAVE, 0); cnt.assign((size_t)KAI_MW_WAVE * KAI_MW_CNT, 0); send.assign(xw_msg_bytes(cap, R), 0); recv.assign(xw_msg_bytes(cap, R) * (size_t)R, 0);
    }
    template <class Io> int wave(Io& io, int b) {  // 0 = merged and written back; else a kai_status (the caller raises the fault flag of its engines)
        if (seen_done) return KAI_ERR_COMM;
        int32_t hdr[8];
        if (int rc = io.pull(hdr, res.data(), cnt.data(), b, cap)) return rc;
        xw_pack(send.data(), ++seq, 0, hdr[1], hdr[6 + b], hdr[4 + b], res.data(), cnt.data(), cap, R, r);
        if (int rc = io.allgather(send.data(), recv.data(), (int64_t)send.size())) return rc;
        exchanges++;
        XWaveMerged m;
        if (!xw_merge(recv.data(), seq, cap, R, res.data(), cnt.data(), m)) { seen_done = true; return KAI_ERR_COMM; }
        if (m.done) seen_done = true;
        return io.push(b, res.data(), cnt.data(), cap, m.hit, m.xrun, (m.fault || m.done) ? 1 : 0);  // (a rank that is gone cannot take its simulations: the wave cannot be counted)
    }
    template <class Io> int finish(Io& io, int fault) {  // the action's last message: "this rank is done" (with its fault flag); skipped once somebody else's was seen
        if (seen_done) return 0;
        xw_pack(send.data(), ++seq, 1, fault, 0x7fffffff, 0, res.data(), cnt.data(), cap, R, r);
        if (int rc = io.allgather(send.data(), recv.data(), (int64_t)send.size())) return rc;
        exchanges++; seen_done = true;
        XWaveMerged m;
        if (!xw_merge(recv.data(), seq, cap, R, res.data(), cnt.data(), m)) return KAI_ERR_COMM;
        return (m.fault && !fault) ? KAI_ERR_COMM : 0;  // another rank ended this action with a fault, this one did not: the group's results are not to be trusted
    }
};

}  // namespace kai
