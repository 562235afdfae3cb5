// host_sim.cpp — TEST INFRASTRUCTURE: compiles the engine's control flow (kai_engine.hpp) for the host.
//
// There is no GPU in the development container, so the device engine's sequential control flow
// (kai::Engine<>, the same header the gfx950 kernel instantiates) is additionally compiled here with
// plain g++ and a serial node scanner, to debug it against the oracle in the `-m "not gpu"` suite.
// This is NOT a CPU fallback: libkai_core never contains it, the package never loads it, and every
// parity claim is made by the `-m gpu` tests through the C ABI on a real MI355X.
#define KAI_SHARED_GPUS 1  // the host twin carries the shared-GPU engine code (ABI v4); the device library is built without it until it is verified on the MI355X
#include <cstdlib>
static int g_dom_lanes_min = std::getenv("KAI_HOSTSIM_DOM_MIN") ? std::atoi(std::getenv("KAI_HOSTSIM_DOM_MIN")) : 16;  // kai_hostsim_set_dom_lanes_min: tests lower it so that small topologies take the scan-lane forms of the domain loops
#define KAI_DOM_LANES_MIN g_dom_lanes_min
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../kai-scheduler_amd/csrc/kai_host_prep.hpp"
#include "../../kai-scheduler_amd/csrc/kai_batch_kernels.hpp"
#include "../../kai-scheduler_amd/csrc/kai_batch_driver.hpp"
#include "../../kai-scheduler_amd/csrc/kai_victim_shard.hpp"
#include "native_bucket_fill.hpp"

using namespace kai;

// the victim search's waves over the ranks of a node-sharded group (kai_victim_shard.hpp): the test's all-gather on host memory, the exchange state of the running action
static int (*g_x_fn)(void*, const void*, void*, int64_t) = nullptr; static void* g_x_user = nullptr; static int g_x_on = 0, g_x_cap = 0; static int64_t g_x_exchanges = 0;
static kai::XShardHost g_xs;
struct HostXIo {  // MultiCtx lives in this process: the "copies" of the device path are plain reads and writes between the wave's two barriers
    kai::MultiCtx* M;
    int pull(int32_t* hdr, int32_t* res, int64_t* cnt, int b, int cap) {
        hdr[0] = M->world; hdr[1] = __atomic_load_n(&M->fault, __ATOMIC_ACQUIRE); hdr[2] = 0; hdr[3] = 0; for (int k = 0; k < 2; k++) { hdr[4 + k] = __atomic_load_n(&M->next[k], __ATOMIC_ACQUIRE); hdr[6 + k] = __atomic_load_n(&M->hit[k], __ATOMIC_ACQUIRE); }
        std::memcpy(res, M->res[b], sizeof(int32_t) * (size_t)cap); std::memcpy(cnt, M->cnt[b], sizeof(int64_t) * kai::KAI_MW_CNT * (size_t)cap); return 0;
    }
    int push(int b, const int32_t* res, const int64_t* cnt, int cap, int hit, int xrun, int fault) {
        std::memcpy(M->res[b], res, sizeof(int32_t) * (size_t)cap); std::memcpy(M->cnt[b], cnt, sizeof(int64_t) * kai::KAI_MW_CNT * (size_t)cap);
        __atomic_store_n(&M->hit[b], hit, __ATOMIC_RELEASE); __atomic_store_n(&M->xrun[b], xrun, __ATOMIC_RELEASE); if (fault) __atomic_store_n(&M->fault, 1, __ATOMIC_RELEASE); return 0;
    }
    int allgather(const void* s, void* r, int64_t n) { return g_x_fn ? g_x_fn(g_x_user, s, r, n) : (int)KAI_ERR_COMM; }
};
namespace {

struct HostBackend {  // serial twin of DevBackend / service_loop (kai_kernels.hpp)
    static constexpr bool kVictim = true;
    template <class T> static void assume_tree(T*) {}
    bool sim_tree(QNode*&, int32_t*&, int32_t*&) { return false; }
    const KaiCtx* cref = nullptr; EngineLocal loc;
    void bind(const KaiCtx& c) { cref = &c; }
    const KaiCtx& ctx() const { return *cref; }
    EngineLocal& local() { return loc; }
    static constexpr bool kBig = true; EngineBig bigv; EngineBig& big() { return bigv; }
    std::vector<uint64_t> s2_key, top_key; std::vector<int32_t> s2_node, top_node;
    static void add_f64(double* p, double v) { *p += v; }
    static void add_i32(int32_t* p, int32_t v) { *p += v; }
    static double coh_f64(const double* p) { return *p; }
    static int32_t coh_i32(const int32_t* p) { return *p; }
    bool topo_scan(const KaiCtx& c, TopoScan& t);  // (below: needs Engine<HostBackend>)
    bool topo_scan_nodes(const KaiCtx& c, TopoScan& t) {  // the node loops of subset_nodes stay serial here; op 4 (build_node_set) word by word as the scan lanes do it
        if (t.op == 15) {  // the survey in a plain loop: what the scan lanes gather (level minima / maxima over `parent`, Idle + Releasing and pod counts per leaf domain)
            const int DT = c.D + c.T;
            t.any = 0; for (int l = 0; l < KAI_TOPO_SCAN_LEVELS; l++) { t.lvl_min[l] = 0x7fffffff; t.lvl_max[l] = -0x7fffffff - 1; }
            for (int n = 0; n < c.N; n++) {
                const int top = c.node_domain[(size_t)t.row0 * c.N + n], leaf = c.node_domain[(size_t)(t.row0 + t.L - 1) * c.N + n];
                if (top < 0 || leaf < 0) continue;
                if (!t.parent || ((t.parent[n >> 5] >> (n & 31)) & 1u)) {
                    t.any = 1;
                    for (int l = 0; l < t.L; l++) { const int dd = c.node_domain[(size_t)(t.row0 + l) * c.N + n]; if (dd < t.lvl_min[l]) t.lvl_min[l] = dd; if (dd > t.lvl_max[l]) t.lvl_max[l] = dd; }
                }
                double av[KAI_MAX_RES];
                for (int r = 0; r < KAI_MAX_RES; r++) av[r] = r < t.R ? c.n_idle[(size_t)r * c.N + n] + c.n_rel[(size_t)r * c.N + n] : 0.0;
                for (int r = 0; r < t.R; r++) c.dom_free[(size_t)leaf * KAI_MAX_RES + r] += av[r];
                if (t.what & 2) c.dom_tmp[2 * DT + leaf] += topo_node_count(t, av);
            }
            return true;
        }
        if (t.op != 4) return false;
        for (int w = 0; w < c.W; w++) {
            uint32_t word = 0; const uint32_t pw = t.parent ? t.parent[w] : 0xffffffffu;
            for (int b = 0; b < 32; b++) {
                const int n = w * 32 + b; if (n >= c.N) break;
                bool in = (pw >> b) & 1u;
                if (in && t.domain >= 0) in = t.dl < 0 ? c.node_domain[(size_t)t.row0 * c.N + n] >= 0 : c.node_domain[(size_t)(t.row0 + t.dl) * c.N + n] == t.domain;
                if (in) word |= 1u << b;
            }
            t.out[w] = word;
        }
        return true;
    }
    bool pfor(const KaiCtx&, const PforReq&) { return false; }
    void or32(uint32_t* w, uint32_t bits) { *w |= bits; }
    void minmax(const KaiCtx& c, int r, double& mn, double& mx) {
        double lo = 1.7976931348623157e308, hi = 0;
        for (int n = 0; n < c.N; n++) {
            if (loc.scope_bits && !((loc.scope_bits[n >> 5] >> (n & 31)) & 1)) continue;
            if (c.n_alloc[(size_t)r * c.N + n] == 0) continue;
            double cur = c.n_idle[(size_t)r * c.N + n] + c.n_rel[(size_t)r * c.N + n];
            if (cur < lo) lo = cur;
            if (cur > hi) hi = cur;
        }
        mn = lo; mx = hi;
    }
    bool eval_nodes(const KaiCtx& c, const ScanReq& q, const int32_t* nodes, int m, double* sc, uint8_t* ok) {  // the single-node re-evaluations of Engine::best_node_kept
        for (int i = 0; i < m; i++) { double s = 0; ok[i] = scan_node_score(c, q, nodes[i], s) ? 1 : 0; sc[i] = s; }
        return true;
    }
    int best_node(const KaiCtx& c, const ScanReq& q, double* score_out = nullptr) {
        int best = -1; double bs = 0;
        for (int n = 0; n < c.N; n++) {
            if (loc.scope_bits && !((loc.scope_bits[n >> 5] >> (n & 31)) & 1)) continue;
            double sc = 0;
            if (!scan_node_score(c, q, n, sc)) continue;
            if (loc.scope_row >= 0) { int dd = c.node_domain[(size_t)loc.scope_row * c.N + n]; double ts = dd >= 0 ? loc.scope_score[dd] : -1.0; if (ts < 0) continue; sc += ts; }
            if (best < 0 || sc > bs) { best = n; bs = sc; }
        }
        if (score_out) *score_out = bs;
        return best;
    }
    void l2(const KaiCtx& c, int k, int sb) {
        uint64_t bk = 0; int bn = 0x7fffffff;
        for (int e = sb * 64; e < std::min(c.NB, sb * 64 + 64); e++) { uint64_t key = c.sum1_key[(size_t)k * c.NB + e]; int n = c.sum1_node[(size_t)k * c.NB + e]; if (key_better(key, n, bk, bn)) { bk = key; bn = n; } }
        s2_key[k * c.NSB + sb] = bk; s2_node[k * c.NSB + sb] = bn;
    }
    void top(const KaiCtx& c, int k) {
        uint64_t bk = 0; int bn = 0x7fffffff;
        for (int sb = 0; sb < c.NSB; sb++) { uint64_t key = s2_key[k * c.NSB + sb]; int n = s2_node[k * c.NSB + sb]; if (key_better(key, n, bk, bn)) { bk = key; bn = n; } }
        top_key[k] = bk; top_node[k] = bn;
    }
    void begin(const KaiCtx& c) {
        if (!c.use_index) return;
        s2_key.assign((size_t)KAI_CMAX * KAI_NSB_MAX, 0); s2_node.assign((size_t)KAI_CMAX * KAI_NSB_MAX, 0); top_key.assign(KAI_CMAX, 0); top_node.assign(KAI_CMAX, 0);
        for (int k = 0; k < c.C; k++) { for (int sb = 0; sb < c.NSB; sb++) l2(c, k, sb); top(c, k); }
    }
    static void build_block(const KaiCtx& c, int k, int b) {
        uint64_t bk = 0; int bn = b * KAI_BLOCK;
        for (int n = b * KAI_BLOCK; n < std::min(c.N, (b + 1) * KAI_BLOCK); n++) { uint64_t key = class_key(c, c.cls[k], n); if (key_better(key, n, bk, bn)) { bk = key; bn = n; } }
        c.sum1_key[(size_t)k * c.NB + b] = bk; c.sum1_node[(size_t)k * c.NB + b] = bn;
    }
    int32_t blocks[KAI_MAXD]; int n = 0;
    bool dirty_add(int b) { for (int i = 0; i < n; i++) if (blocks[i] == b) return true; if (n == KAI_MAXD) return false; blocks[n++] = b; return true; }
    int dirty_count() { return n; }
    void refresh(const KaiCtx& c) {
        for (int i = 0; i < n; i++) for (int k = 0; k < c.C; k++) build_block(c, k, blocks[i]);
        for (int k = 0; k < c.C; k++) { for (int i = 0; i < n; i++) l2(c, k, blocks[i] / 64); top(c, k); }
        n = 0;
    }
    void class_top(const KaiCtx&, int k, uint64_t& key, int& node) { key = top_key[k]; node = top_node[k]; }
    bool all_dead(const KaiCtx& c) { for (int k = 0; k < c.C; k++) if (top_key[k]) return false; return true; }
    // job staging (kai_engine.hpp JobPf): the device hands it to a service wave and overlaps it with the rest of the pop; here it runs in place,
    // with several "lanes" so that the strided pod assignment is exercised
    void stage_async(const KaiCtx& c, int j) {
        if (n) { KAI_JOBPF.job = -1; KAI_JOBPF.ok = 0; return; }
        int bad = 0; for (int lane = 3; lane >= 0; lane--) bad |= stage_job_lane(c, j, KAI_FRAME, KAI_JOBPF, lane, 4);
        KAI_JOBPF.ok = KAI_JOBPF.shape && !bad;
    }
    bool staged(int j) { bool r = KAI_JOBPF.job == j && KAI_JOBPF.ok; KAI_JOBPF.job = -1; return r; }
    void hot(const KaiCtx& c, QNode*& qn, int32_t*& qheap, int32_t*& root_heap) { qn = c.qn; qheap = c.qheap; root_heap = c.root_heap; }
    // several engines of one victim action (MultiCtx): here they are threads of this process
    static void mw_store32(int32_t* p, int32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
    static int32_t mw_load32(const int32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
    static void mw_store64(int64_t* p, int64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
    static int64_t mw_load64(const int64_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
    static int32_t mw_fetch_add32(int32_t* p, int32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
    static void mw_fetch_min32(int32_t* p, int32_t v) { int32_t o = __atomic_load_n(p, __ATOMIC_ACQUIRE); while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {} }
    static void mw_exchange(const KaiCtx&, MultiCtx* m, int b) { HostXIo io{m}; if (g_xs.wave(io, b)) __atomic_store_n(&m->fault, 1, __ATOMIC_RELEASE); }  // engine 0, between the wave's two barriers
    static void grid_sync(MultiCtx* m, int clear) {
        const int gen = __atomic_load_n(&m->bar_gen, __ATOMIC_ACQUIRE);
        if (__atomic_add_fetch(&m->bar_count, 1, __ATOMIC_ACQ_REL) == m->world) { __atomic_store_n(&m->next[clear], 0, __ATOMIC_RELAXED); __atomic_store_n(&m->hit[clear], 0x7fffffff, __ATOMIC_RELAXED); __atomic_store_n(&m->bar_count, 0, __ATOMIC_RELAXED); __atomic_add_fetch(&m->bar_gen, 1, __ATOMIC_RELEASE); }
        else { long spins = 0; while (__atomic_load_n(&m->bar_gen, __ATOMIC_ACQUIRE) == gen) { if (++spins > 64) std::this_thread::yield(); if (spins > 400000000L) { __atomic_store_n(&m->fault, 1, __ATOMIC_RELEASE); break; } } }
    }
    int64_t clock() {
#if defined(KAI_PROF_VICTIM) && defined(__x86_64__)
        return (int64_t)__builtin_ia32_rdtsc();  // phase clocks of the host-compiled engine (debug builds only)
#else
        return 0;
#endif
    }
};

// TopoScan ops over domains (5..9): the same per-domain bodies the scan lanes of the action kernel run (Engine::topo_dom_body), in a plain loop
bool HostBackend::topo_scan(const KaiCtx& c, TopoScan& t) {
    if (t.op < 5 || t.op == 15) return topo_scan_nodes(c, t);
    HostBackend tmp; Engine<HostBackend> e(c, tmp);
    int cnt = 0;
    for (int d = 0; d < c.D + c.T; d++) cnt += e.topo_dom_body(t, d);
    if (t.op == 9) t.any = cnt;
    return true;
}

static double g_native_fill_ms = 0; static int64_t g_native_fill_launches = 0, g_native_fill_decisions = 0, g_native_fill_diffs = 0;  // the native shadow of the bucket fill, summed since the last read
extern "C" int kai_hostsim_native_fill(double* ms, int64_t* launches, int64_t* decisions, int64_t* diffs) {
    *ms = g_native_fill_ms; *launches = g_native_fill_launches; *decisions = g_native_fill_decisions; *diffs = g_native_fill_diffs;
    g_native_fill_ms = 0; g_native_fill_launches = g_native_fill_decisions = g_native_fill_diffs = 0; return 0;
}
// the batch path's kernels on the lock-step emulator (kai_simt.hpp)
struct HostLauncher {
    void static_rank(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_static_rank(c); }); }
    void static_check(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_static_check(c); }); }
    void qualify(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_qualify(c); }); }
    void nrec(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_build_nrec(c); }); }
    void plan_setup(int g, int b, const KaiCtx& c, RoundParams rp) { kw::launch(g, b, 0, [&] { kb_plan_setup(c, rp); }); }
    void plan_leaf(int g, int b, const KaiCtx& c, RoundParams rp) { kw::launch(g, b, 0, [&] { kb_plan_leaf(c, rp); }); }
    void plan_rank(int g, int b, const KaiCtx& c, RoundParams rp) { kw::launch(g, b, 0, [&] { kb_plan_rank(c, rp); }); }
    void plan_gather(int g, int b, const KaiCtx& c, RoundParams rp) { kw::launch(g, b, 0, [&] { kb_plan_gather(c, rp); }); }
    void plan_scan(int g, int b, const KaiCtx& c, RoundParams rp) { kw::launch(g, b, 0, [&] { kb_plan_scan(c, rp); }); }
    void seg_sum(int g, int b, const KaiCtx& c, RoundParams rp, int segs) { kw::launch(g, b, 0, [&] { kb_seg_sum(c, rp, segs); }); }
    void seg_gate(int g, int b, const KaiCtx& c, RoundParams rp, int segs) { kw::launch(g, b, 0, [&] { kb_seg_gate(c, rp, segs); }); }
    void seg_keys(int g, int b, const KaiCtx& c, RoundParams rp, int segs) { kw::launch(g, b, 0, [&] { kb_seg_keys(c, rp, segs); }); }
    void seg_max(int g, int b, const KaiCtx& c, RoundParams rp, int segs) { kw::launch(g, b, 0, [&] { kb_seg_max(c, rp, segs); }); }
    void plan_emit(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_plan_emit(c); }); }
    void class_capacity(int g, int b, const KaiCtx& c, int buckets, int levels) { kw::launch(g, b, 0, [&] { kb_class_capacity(c, buckets, levels); }); }
    void fill(int g, int b, size_t dyn, const KaiCtx& c, RoundParams rp, int l1) { kw::launch(g, b, dyn, [&] { kb_fill(c, rp, l1); }); }
    void bucket_build(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_bucket_build(c); }); }
    // KAI_HOSTSIM_NATIVE_FILL=1: the same algorithm as plain scalar C++ (native_bucket_fill.hpp) on a copy of the sets first — timed, and every output checked against the emulated kernel's
    template <class F> void with_native_shadow(const KaiCtx& c, RoundParams rp, BucketParams bp, F&& run) {
        const bool shadow = std::getenv("KAI_HOSTSIM_NATIVE_FILL") != nullptr;  // (read per launch: a test process sets it for single tests)
        kai_native::NativeFillOut nat;
        if (shadow) kai_native::native_fill_buckets(c, rp, bp, nat);
        run();
        if (!shadow) return;
        const BatchCtx& bt = c.bt; const FillStatus& f = bt.fs[0]; bool same = true;
        same = same && f.n_done == nat.fs.n_done && f.mismatch == nat.fs.mismatch && f.all_dead == nat.fs.all_dead && f.planned == nat.fs.planned && f.decisions == nat.fs.decisions && f.attempted == nat.fs.attempted &&
               f.committed == nat.fs.committed && f.rollbacks == nat.fs.rollbacks && f.ops == nat.fs.ops && f.dead_mask == nat.fs.dead_mask;
        for (size_t i = 0; i < nat.words.size() && same; i++) same = bt.bk_words[i] == nat.words[i];
        if (!same && std::getenv("KAI_HOSTSIM_DEBUG")) std::fprintf(stderr, "native shadow: counters or sets differ\n");
        for (int gi = rp.start; gi < f.n_done && same; gi++) {
            const int x = gi - rp.start;
            same = bt.g_out[gi] == nat.g_out[x] && (nat.g_out[x] != BF_OK || (bt.g_opoff[gi] == nat.g_opoff[x] && bt.g_stmt[gi] == nat.g_stmt[x]));  // (offsets: what the apply kernels read, committed jobs only)
            if (!same && std::getenv("KAI_HOSTSIM_DEBUG")) std::fprintf(stderr, "native shadow: job %d (start %d) flag %d out %d/%d opoff %d/%d stmt %d/%d\n", gi, rp.start, (int)bt.g_flag[gi], (int)bt.g_out[gi], (int)nat.g_out[x], bt.g_opoff[gi], nat.g_opoff[x], bt.g_stmt[gi], nat.g_stmt[x]);
            if (same && nat.g_out[x] == BF_OK && bt.g_flag[gi] != BF_GATE) for (int t = 0; t < bt.g_nt[gi] && same; t++) same = bt.t_node[bt.g_first[gi] + t] == nat.t_node[(size_t)bt.g_first[gi] + t];
        }
        if (!same && std::getenv("KAI_HOSTSIM_DEBUG")) std::fprintf(stderr, "native shadow differs: n_done %d/%d mismatch %d/%d all_dead %d/%d planned %d/%d decisions %lld/%lld attempted %lld/%lld committed %lld/%lld rollbacks %lld/%lld ops %lld/%lld dead %llx/%llx\n",
            f.n_done, nat.fs.n_done, f.mismatch, nat.fs.mismatch, f.all_dead, nat.fs.all_dead, f.planned, nat.fs.planned, (long long)f.decisions, (long long)nat.fs.decisions, (long long)f.attempted, (long long)nat.fs.attempted,
            (long long)f.committed, (long long)nat.fs.committed, (long long)f.rollbacks, (long long)nat.fs.rollbacks, (long long)f.ops, (long long)nat.fs.ops, (unsigned long long)f.dead_mask, (unsigned long long)nat.fs.dead_mask);
        g_native_fill_ms += nat.ms; g_native_fill_launches++; g_native_fill_decisions += nat.fs.decisions; if (!same) g_native_fill_diffs++;
    }
    static bool round_off(const KaiCtx& c) { return c.bt.dev_loop && c.bt.ctl->done; }  // a round enqueued ahead of the loop's end: its kernels leave at once (no shadow, no dump)
    void fill_buckets(int g, int b, size_t dyn, const KaiCtx& c, RoundParams rp, BucketParams bp) { if (round_off(c)) return; with_native_shadow(c, rp, bp, [&] { kw::launch(g, b, dyn, [&] { kb_fill_buckets(c, rp, bp); }); }); }
    // KAI_HOSTSIM_FILL_DUMP=<prefix>: inputs and outputs of every fill launch over >= 1 000 planned jobs as <prefix>_<n>.bin (what tools/micro/fill_bench.hip replays on the MI355X)
    static void fill_dump(const KaiCtx& c, RoundParams rp, BucketParams bp, bool outputs) {
        const char* pre = std::getenv("KAI_HOSTSIM_FILL_DUMP"); if (!pre || rp.mode == 1) return;
        const BatchCtx& bt = c.bt; const int V = bt.q_valid[c.Q]; if (V - rp.start < 1000) return;
        static int seq = 0; if (!outputs) seq++;
        char path[512]; std::snprintf(path, sizeof path, "%s_%d.%s", pre, seq, outputs ? "out" : "in");
        FILE* f = std::fopen(path, "wb"); if (!f) return;
        auto w = [&](const void* p, size_t n) { std::fwrite(p, 1, n, f); };
        if (!outputs) {
            int32_t hdr[16] = {0x4b464c31, c.C, c.Q, c.P, V, bp.levels, bp.nw, bp.nw1, bp.n_ok, c.NB, 0, 0, 0, 0, 0, 0}; w(hdr, sizeof hdr); w(&rp, sizeof rp); w(&bp, sizeof bp);
            for (int k = 0; k < 64; k++) { const double q = k < c.C ? c.cls[k].req[KAI_RES_GPU] : 0.0; w(&q, 8); }
            w((const void*)bt.g_flag, (size_t)V); w((const void*)bt.g_first, (size_t)V * 4); w((const void*)bt.g_nt, (size_t)V * 4); w((const void*)bt.g_ucls, (size_t)V * 4);
            w((const void*)bt.t_cls, (size_t)c.P * 4); w((const void*)bt.bk_words, (size_t)bp.levels * bp.nw * 8);
        } else {
            w((const void*)bt.fs, sizeof(FillStatus)); w((const void*)bt.g_out, (size_t)V); w((const void*)bt.g_opoff, (size_t)V * 4); w((const void*)bt.g_stmt, (size_t)V * 4);
            w((const void*)bt.t_node, (size_t)c.P * 4); w((const void*)bt.bk_words, (size_t)bp.levels * bp.nw * 8);
        }
        std::fclose(f);
    }
    void fill_levels(int g, int b, size_t dyn, const KaiCtx& c, RoundParams rp, BucketParams bp) { if (round_off(c)) return; fill_dump(c, rp, bp, false); with_native_shadow(c, rp, bp, [&] { kw::launch(g, b, dyn, [&] { kb_fill_levels(c, rp, bp); }); }); fill_dump(c, rp, bp, true); }
    void fill_counts(int g, int b, size_t dyn, const KaiCtx& c, RoundParams rp, BucketParams bp) { if (round_off(c)) return; with_native_shadow(c, rp, bp, [&] { kw::launch(g, b, dyn, [&] { kb_fill_counts(c, rp, bp); }); }); }
    void apply_jobs(int g, int b, const KaiCtx& c, int64_t ops_base, int64_t stmt_base) { kw::launch(g, b, 0, [&] { kb_apply_jobs(c, ops_base, stmt_base); }); }
    void apply_nodes(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_apply_nodes(c); }); }
    void index_from_recs(int g, int b, const KaiCtx& c, const NodeRec* recs, int n_recs, uint64_t* l1k, int32_t* l1n, int nb, int blk0, int blk1) { kw::launch(g, b, 0, [&] { kb_index_from_recs(c, recs, n_recs, l1k, l1n, nb, blk0, blk1); }); }
    void shard_mask_nrec(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_shard_mask_nrec(c); }); }
    void shard_keys(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_shard_keys(c); }); }
    void shard_select(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_shard_select(c); }); }
    void shard_compact(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_shard_compact(c); }); }
    void shard_vbuild(int g, int b, const KaiCtx& c) { kw::launch(g, b, 0, [&] { kb_shard_vbuild(c); }); }
    void shard_scatter(int g, int b, const KaiCtx& c, int total) { kw::launch(g, b, 0, [&] { kb_shard_scatter(c, total); }); }
    int (*ag_fn)(void*, const void*, void*, int64_t) = nullptr; void* ag_user = nullptr;
    int allgather(const void* send, void* recv, int64_t bytes) { return ag_fn ? ag_fn(ag_user, send, recv, bytes) : (int)KAI_ERR_COMM; }  // the test's gloo all-gather
    // rounds without the host: launches complete where they are made, so the "pinned copies" are plain copies
    RoundCtl rslot[KB_ROUND_SLOTS];
    void round_init(const KaiCtx& c, int remaining, int H0, int policy, int64_t ops_base, int64_t stmt_base) { kw::launch(1, 64, 0, [&] { kb_round_init(c, remaining, H0, policy, ops_base, stmt_base); }); }
    void round_next(const KaiCtx& c) { kw::launch(1, 64, 0, [&] { kb_round_next(c); }); }
    void round_finish(const KaiCtx& c) { kw::launch(1, 64, 0, [&] { kb_round_finish(c); }); }
    void round_begin(int) {}
    int round_post(int slot, const void* src, size_t n) { std::memcpy(&rslot[slot], src, n); return 0; }
    int round_wait(int slot, void* dst, size_t n) { std::memcpy(dst, &rslot[slot], n); return 0; }
    int read(void* dst, const void* src, size_t n) { if (n) std::memcpy(dst, src, n); return 0; }
    int write(void* dst, const void* src, size_t n) { std::memcpy(dst, src, n); return 0; }
    int zero(void* dst, size_t n) { std::memset(dst, 0, n); return 0; }
};

template <class T> T* own(std::vector<std::vector<char>>& pool, size_t n) {
    pool.emplace_back(std::max<size_t>(n, 1) * sizeof(T), 0);
    return reinterpret_cast<T*>(pool.back().data());
}
template <class T> const T* copy(std::vector<std::vector<char>>& pool, const T* src, size_t n) {
    T* d = own<T>(pool, n);
    if (n && src) std::memcpy(d, src, n * sizeof(T));
    return d;
}

}  // namespace

// exhaustive checks of the small arithmetic helpers of kai_fill_levels.hpp (tests/test_batch_path.py): 0 = all hold, else the first failing case as a << 8 | b
extern "C" int kai_hostsim_fill_levels_selfcheck() {
    for (int b = 1; b <= 8; b++) for (int a = 0; a <= 1024; a++) if (kfl_div(a, b) != a / b) return (a << 8) | b;
    for (int lv = 1; lv <= KFL_LMAX; lv++) for (int lane = 0; lane < 64; lane++) { const uint32_t t = kfl_tab(lane, lv); for (int g = 0; g <= lv; g++) if (kfl_quot(t, g) != g / (lane + 1)) return (g << 8) | (lane + 1) | (1 << 30); }
    for (int g = 2; g <= KFL_LMAX; g++) for (int g2 = 1; g2 < g; g2++) { const int p = kfl_pair(g, g2); if (p < 0 || p >= KFL_PAIRS) return (g << 8) | g2 | (1 << 29); for (int h = 2; h <= KFL_LMAX; h++) for (int h2 = 1; h2 < h; h2++) if ((h != g || h2 != g2) && kfl_pair(h, h2) == p) return (g << 8) | g2 | (1 << 28); }
    for (int g = 0; g <= 8; g++) for (int g2 = 0; g2 <= 8; g2++) for (int per = 0; per <= 8; per++) for (int k : {0, 1, 7, 1024}) { const uint64_t c = kfl_cmd(g, g2, k, per, 0x7ffffff0); const int a = (int)(uint32_t)c;
        if ((a & 15) != g || ((a >> 4) & 15) != g2 || ((a >> 8) & 15) != per || ((a >> 12) & 0x7ff) != k || (int)(uint32_t)(c >> 32) != 0x7ffffff0) return (g << 8) | g2 | (1 << 27); }
    return 0;
}
// node-sharded run (tests/test_dist_gloo.py): rank / world / offers per class and the all-gather of the test's process group, for the next run
static int g_sh_rank = 0, g_sh_world = 1, g_sh_k = 0; static int (*g_sh_fn)(void*, const void*, void*, int64_t) = nullptr; static void* g_sh_user = nullptr; static int64_t g_sh_exchanges = 0;
extern "C" void kai_hostsim_set_shard(int rank, int world, int k, int (*fn)(void*, const void*, void*, int64_t), void* user) { g_sh_rank = rank; g_sh_world = world; g_sh_k = k; g_sh_fn = fn; g_sh_user = user; }
extern "C" int64_t kai_hostsim_last_exchanges() { return g_sh_exchanges; }
// victim actions of a node-sharded run: on = their simulation waves are dealt out over the ranks (kai_victim_shard.hpp) through fn, an all-gather on host memory;
// cap = simulations per wave (0 = the default: two per engine of the group).  Off = every rank runs the whole action (replicated).
extern "C" void kai_hostsim_set_victim_shard(int on, int cap, int (*fn)(void*, const void*, void*, int64_t), void* user) { g_x_on = on; g_x_cap = cap; g_x_fn = fn; g_x_user = user; g_x_exchanges = 0; }
extern "C" int64_t kai_hostsim_victim_exchanges() { return g_x_exchanges; }
static int g_mw_world = 1; static int64_t g_mw_waves = 0, g_mw_sims_run = 0, g_mw_sims_used = 0, g_mw_replays = 0;
extern "C" void kai_hostsim_set_dom_lanes_min(int n) { g_dom_lanes_min = n < 1 ? 1 : n; }
extern "C" void kai_hostsim_set_multi(int engines) { g_mw_world = engines < 1 ? 1 : engines > KAI_MW_MAX ? KAI_MW_MAX : engines; g_mw_waves = g_mw_sims_run = g_mw_sims_used = g_mw_replays = 0; }  // victim actions of the next runs on that many engines
extern "C" void kai_hostsim_multi_stats(int64_t* out) { out[0] = g_mw_waves; out[1] = g_mw_sims_run; out[2] = g_mw_sims_used; out[3] = g_mw_replays; }
static std::vector<int32_t> g_last_groups;  // PodInfo.GPUGroups[0] of the active fraction pods after the last run
static int64_t g_last_victim_stats[3] = {0, 0, 0};  // scenarios, simulations, filtered scenarios of the last run, summed over its actions (the oracle exports the same: kai_oracle_last_victim_stats)
extern "C" int kai_hostsim_last_victim_stats(int64_t* out3) { for (int i = 0; i < 3; i++) out3[i] = g_last_victim_stats[i]; return 0; }
extern "C" int kai_hostsim_last_gpu_groups(int32_t* out, int cap) { int n = (int)g_last_groups.size(); for (int i = 0; i < n && i < cap; i++) out[i] = g_last_groups[i]; return n; }
// host clocks of the per-cycle host preparation (HostPrep + SharedPods: what kai_session_open does before the first upload), phase by phase — timing aid
extern "C" int kai_hostsim_prep_ms(const kai_config* cfg, const kai_snapshot_soa* s, double* out, int cap) {
    auto t0 = std::chrono::steady_clock::now();
    // (kept between calls, as kai_core keeps them between sessions: after the first call the arrays are touched memory)
    static SharedPods sp; if (!sp.build(*cfg, s)) return KAI_ERR_UNSUPPORTED;
    auto t1 = std::chrono::steady_clock::now();
    static HostPrep prep; std::string err;
    if (prep.build(*cfg, s, err)) return KAI_ERR_INVALID_ARG;
    auto t2 = std::chrono::steady_clock::now();
    if (cap > 0) out[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (cap > 1) out[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    for (int i = 0; i < 8 && i + 2 < cap; i++) out[i + 2] = prep.phase_ms[i];
    return 0;
}
// HostPrep::build's verdict on a snapshot: its return code and, for a refused one, the message (what kai_session_open hands to kai_last_error)
extern "C" int kai_hostsim_prep_error(const kai_config* cfg, const kai_snapshot_soa* s, char* msg, int cap) {
    HostPrep prep; std::string err;
    const int rc = prep.build(*cfg, s, err);
    if (msg && cap > 0) { std::snprintf(msg, (size_t)cap, "%s", err.c_str()); }
    return rc;
}
extern "C" int kai_hostsim_run(const kai_config* cfg, const kai_snapshot_soa* s, const int* actions, int n_actions,
                               kai_op* ops_out, int64_t ops_cap, int64_t* n_ops, int32_t* pod_status_out, int32_t* pod_node_out,
                               kai_queue_share* shares_open, kai_queue_share* shares_final, kai_node_state* nodes_out,
                               kai_action_stats* stats, double* elapsed_ms_out) {
    if (!cfg || !s || s->abi_version != KAI_ABI_VERSION) return KAI_ERR_INVALID_ARG;
    if (const char* e = std::getenv("KAI_HOSTSIM_MW")) { const int v = std::atoi(e); if (v > 0) g_mw_world = v > KAI_MW_MAX ? KAI_MW_MAX : v; }
    SharedPods sp;  // shared GPUs in the engine: one GPU memory size for the whole cluster (the queue-capacity step stays node independent)
    if (!sp.build(*cfg, s)) return KAI_ERR_UNSUPPORTED;
    const bool shared = sp.any;
    const int N = s->n_nodes, P = s->n_pods, S = s->n_podsets, J = s->n_jobs, Q = s->n_queues, R = s->n_res;
    // lean_shared_pods_hold: kai_session_open (kai_core.hip) does not SEND the per-pod arrays of the shared-GPU model for a snapshot without shared-GPU requests and MIG rows — it
    // writes the constants they hold on the device.  That they hold exactly those constants is checked here, on every snapshot the CPU suite runs.
    if (!sp.on) for (int p = 0; p < P; p++) {
        const double g = s->pod_req[(size_t)KAI_RES_GPU * P + p];
        const bool same = std::memcmp(&sp.acc_gpu[p], &g, 8) == 0 && std::memcmp(&sp.pend_gpu[p], &g, 8) == 0 && std::memcmp(&sp.quota_gpu[p], &g, 8) == 0;
        const double z = 0.0;
        if (!same || sp.shared[p] != 0 || sp.mem[p] != 0 || sp.gmem[p] != 0 || std::memcmp(&sp.mig_q[p], &z, 8) != 0) { std::fprintf(stderr, "lean_shared_pods_hold: pod %d breaks the constants kai_session_open writes on the device\n", p); return KAI_ERR_STATE; }
    }
    std::vector<std::vector<char>> pool;
    HostPrep prep; std::string err;
    prep.shared_pods = sp.any ? &sp : nullptr;
    if (prep.build(*cfg, s, err)) return KAI_ERR_INVALID_ARG;
    {   // the other constants kai_session_open writes on the device instead of sending them (kai_core.hip: prep.groups_default, !prep.any_nominated) — checked on every snapshot the CPU suite runs
        bool ok = true;
        if (!prep.any_nominated) for (int p = 0; p < P; p++) ok = ok && prep.pod_nominated[p] == -1;
        if (prep.groups_default) {
            ok = ok && prep.G == J && (int)prep.g_job.size() == J && (int)prep.g_parent.size() == J && (int)prep.g_topo.size() == J && (int)prep.g_req.size() == J && (int)prep.g_pref.size() == J && (int)prep.g_name_rank.size() == J
                    && (int)prep.g_child_off.size() == J + 1 && (int)prep.j_has_topology.size() == std::max(J, 1) && (int)prep.s_group.size() == S && (int)prep.s_topo.size() == S && (int)prep.s_req.size() == S && (int)prep.s_pref.size() == S;
            for (int j = 0; j < J && ok; j++) ok = prep.g_parent[j] == -1 && prep.g_topo[j] == -1 && prep.g_req[j] == -1 && prep.g_pref[j] == -1 && prep.g_name_rank[j] == 0 && prep.g_child_off[j] == 0 && prep.j_has_topology[j] == 0;
            ok = ok && prep.g_child_off[J] == 0 && prep.j_has_topology[0] == 0;
            for (int k = 0; k < S && ok; k++) ok = prep.s_group[k] == s->podset_job[k] && prep.s_topo[k] == -1 && prep.s_req[k] == -1 && prep.s_pref[k] == -1;
        }
        if (!ok) { std::fprintf(stderr, "host_sim: HostPrep's default tables differ from the constants kai_session_open writes on the device\n"); return KAI_ERR_STATE; }
    }
    for (int p = 0; p < P; p++)  // NodeInfo.LegacyMIGTasks (node_info.go:407-409): a node that holds a legacy MIG task takes no MIG request
        if (s->pod_flags && (s->pod_flags[p] & KAI_POD_LEGACY_MIG) && prep.pod_node[p] >= 0 && (s->pod_status[p] & (KAI_POD_ALLOCATED | KAI_POD_PIPELINED | KAI_POD_BINDING | KAI_POD_BOUND | KAI_POD_RUNNING | KAI_POD_RELEASING))) prep.node_flags[prep.pod_node[p]] |= KAI_NODE_LEGACY_MIG_I;
    KaiCtx c{};
    // every array of the context (part 1: what the session-open math below works on); a lambda so that further engines of a victim action get replicas of identical layout
    auto alloc1 = [&](std::vector<std::vector<char>>& pool, KaiCtx& c) {
    c.N = N; c.P = P; c.S = S; c.J = J; c.Q = Q; c.R = R; c.n_pod_classes = std::max(1, s->n_pod_classes); c.n_node_classes = std::max(1, s->n_node_classes);
    c.plugins = cfg->plugins; c.gpu_strategy = cfg->gpu_strategy; c.cpu_strategy = cfg->cpu_strategy; c.restrict_nodes = cfg->restrict_node_scheduling;
    c.k_value = cfg->k_value <= 0.0 ? 0.0 : cfg->k_value;
    std::vector<int32_t> zerop(P, 0); std::vector<uint32_t> zeropu(P, 0); uint8_t one = 1;
    c.n_alloc = copy(pool, prep.node_alloc.data(), (size_t)R * N); c.n_flags = const_cast<uint32_t*>(copy(pool, prep.node_flags.data(), N));
    c.n_gpu_count = copy(pool, prep.node_gpu_count.data(), N); c.n_class = copy(pool, prep.node_class.data(), N);
    c.p_req = copy(pool, s->pod_req, (size_t)R * P); c.p_job = copy(pool, s->pod_job, P); c.p_podset = copy(pool, s->pod_podset, P);
    c.p_flags = copy(pool, s->pod_flags ? s->pod_flags : zeropu.data(), P); c.p_class = copy(pool, s->pod_class ? s->pod_class : zerop.data(), P);
    c.p_nominated = copy(pool, prep.pod_nominated.data(), P); c.p_scls = copy(pool, prep.pod_scls.data(), P);
    c.s_job = copy(pool, s->podset_job, S); c.s_min = copy(pool, s->podset_min_available, S); c.s_name_rank = copy(pool, s->podset_name_rank, S);
    c.j_queue = copy(pool, s->job_queue, J); c.j_prio = copy(pool, s->job_priority, J); c.j_preempt = copy(pool, s->job_preemptible, J);
    c.j_created = copy(pool, s->job_created_ns, J); c.j_uid_rank = copy(pool, s->job_uid_rank, J); c.j_first_pod = copy(pool, s->job_first_pod, J);
    c.j_n_pods = copy(pool, s->job_n_pods, J); c.j_first_ps = copy(pool, s->job_first_podset, J); c.j_n_ps = copy(pool, s->job_n_podsets, J);
    c.q_parent = copy(pool, s->queue_parent, Q); c.q_prio = copy(pool, s->queue_priority, Q); c.q_created = copy(pool, s->queue_created_ns, Q); c.q_uid_rank = copy(pool, s->queue_uid_rank, Q);
    if (s->class_fit && s->n_pod_classes > 0 && s->n_node_classes > 0) c.class_fit = copy(pool, s->class_fit, (size_t)s->n_pod_classes * s->n_node_classes);
    else c.class_fit = copy(pool, &one, 1);
    c.j_pods_sorted = copy(pool, prep.sorted.data(), P); c.q_child_off = copy(pool, prep.child_off.data(), Q + 2); c.q_children = copy(pool, prep.children.data(), std::max(Q, 1));
    c.q_job_off = copy(pool, prep.job_off.data(), Q + 1); c.jobs_static = copy(pool, prep.jobs_static.data(), std::max(J, 1)); c.q_depth_order = copy(pool, prep.depth_order.data(), Q);
    c.C = (int)prep.classes.size(); c.NB = (N + KAI_BLOCK - 1) / KAI_BLOCK; c.NSB = (c.NB + 63) / 64; c.use_index = c.C > 0; c.all_tracked = prep.all_tracked; c.fast_ok = prep.fast_ok; c.exact_sums = prep.exact_sums;
    if (shared || sp.mig) { c.all_tracked = 0; const char* e = std::getenv("KAI_SHARED_INDEX"); const int lvl = e ? std::atoi(e) : 2; if (sp.mig || lvl == 0) c.use_index = 0; if (sp.mig || lvl < 2) c.fast_ok = 0; }  // as kai_session_open: fractions in no class, the others indexed with the gpusharingorder bit
    { int d = cfg->queue_depth[KAI_ACTION_ALLOCATE]; c.queue_depth = d > 0 ? d : 0; }
    c.cls = copy(pool, prep.classes.data(), prep.classes.size());
    c.T = prep.T; c.TL = prep.TL; c.D = prep.D; c.G = prep.G; c.W = (N + 31) / 32;
    { const size_t DT = (size_t)prep.D + prep.T;
      c.topo_level_off = copy(pool, prep.topo_level_off.data(), prep.topo_level_off.size()); c.node_domain = copy(pool, prep.node_domain.data(), prep.node_domain.size());
      c.dom_level = copy(pool, prep.dom_level.data(), prep.dom_level.size()); c.dom_topo = copy(pool, prep.dom_topo.data(), prep.dom_topo.size());
      c.dom_parent = copy(pool, prep.dom_parent.data(), prep.dom_parent.size()); c.dom_id_rank = copy(pool, prep.dom_id_rank.data(), prep.dom_id_rank.size());
      c.dom_child_off = copy(pool, prep.dom_child_off.data(), prep.dom_child_off.size()); c.dom_children = const_cast<int32_t*>(copy(pool, prep.dom_children.data(), prep.dom_children.size()));
      c.dom_alloc_pods = own<int32_t>(pool, DT); c.dom_free = own<double>(pool, DT * KAI_MAX_RES); c.dom_tmp = own<int32_t>(pool, 3 * DT + 4); c.dom_ratio = own<double>(pool, DT); c.dom_key = own<int64_t>(pool, 2 * DT);
      c.ns_bits = own<uint32_t>(pool, (size_t)KAI_TDEPTH * std::max(c.W, 1)); c.ns_sets = own<int32_t>(pool, (size_t)KAI_TDEPTH * (DT + 1));
      c.sg_score = own<double>(pool, (size_t)KAI_TKEYS * std::max<size_t>(DT, 1)); c.sg_key = own<int32_t>(pool, KAI_TKEYS); c.sg_row = own<int32_t>(pool, KAI_TKEYS); }
    c.g_job = copy(pool, prep.g_job.data(), prep.g_job.size()); c.g_parent = copy(pool, prep.g_parent.data(), prep.g_parent.size()); c.g_name_rank = copy(pool, prep.g_name_rank.data(), prep.g_name_rank.size());
    c.g_topo = copy(pool, prep.g_topo.data(), prep.g_topo.size()); c.g_req = copy(pool, prep.g_req.data(), prep.g_req.size()); c.g_pref = copy(pool, prep.g_pref.data(), prep.g_pref.size());
    c.j_root_group = copy(pool, prep.j_root_group.data(), prep.j_root_group.size()); c.g_child_off = copy(pool, prep.g_child_off.data(), prep.g_child_off.size()); c.g_children = copy(pool, prep.g_children.data(), prep.g_children.size());
    c.s_group = copy(pool, prep.s_group.data(), prep.s_group.size()); c.s_topo = copy(pool, prep.s_topo.data(), prep.s_topo.size()); c.s_req = copy(pool, prep.s_req.data(), prep.s_req.size());
    c.s_pref = copy(pool, prep.s_pref.data(), prep.s_pref.size()); c.j_has_topology = copy(pool, prep.j_has_topology.data(), prep.j_has_topology.size());
    c.sum1_key = own<uint64_t>(pool, (size_t)std::max(c.C, 1) * std::max(c.NB, 1)); c.sum1_node = own<int32_t>(pool, (size_t)std::max(c.C, 1) * std::max(c.NB, 1));
    c.n_idle = const_cast<double*>(copy(pool, prep.node_alloc.data(), (size_t)R * N)); c.n_rel = own<double>(pool, (size_t)R * N); c.n_used = own<double>(pool, (size_t)R * N);
    c.p_status = const_cast<int32_t*>(copy(pool, s->pod_status, P)); c.p_node = const_cast<int32_t*>(copy(pool, prep.pod_node.data(), P));
    c.p_on_node = own<int32_t>(pool, P); c.p_on_node_status = own<int32_t>(pool, P); c.p_virtual = own<uint8_t>(pool, P); c.p_accepted = own<uint8_t>(pool, P);
    {   // shared-GPU tables (KaiCtx, KAI_SHARED_GPUS): nodes in name-rank order like every other node array
        c.shared_on = shared ? 1 : 0;
        double* por = own<double>(pool, P); int32_t* grp = own<int32_t>(pool, P); int32_t* ogrp = own<int32_t>(pool, P); int64_t* gm = own<int64_t>(pool, N);
        int32_t next_new = KAI_NEW_GROUP;
        for (int p = 0; p < P; p++) { por[p] = s->pod_gpu_portion ? s->pod_gpu_portion[p] : 0.0; grp[p] = (s->pod_gpu_group && sp.shared[p]) ? s->pod_gpu_group[p] : -1; ogrp[p] = -1; if (grp[p] >= next_new) next_new = grp[p] + 1; }
        c.p_shared = copy(pool, sp.shared.data(), P); c.p_mem = copy(pool, sp.mem.data(), P); c.p_gmem = copy(pool, sp.gmem.data(), P); c.p_acc_gpu = copy(pool, sp.acc_gpu.data(), P); c.p_pend_gpu = copy(pool, sp.pend_gpu.data(), P);
        c.p_quota_gpu = copy(pool, sp.quota_gpu.data(), P); c.p_mig_q = copy(pool, sp.mig_q.data(), P); c.p_kind = copy(pool, sp.kind.data(), P); c.quota_on = sp.on ? 1 : 0; c.mig_on = sp.mig ? 1 : 0;
        for (int r = 0; r < KAI_MAX_RES; r++) { c.res_mig_g[r] = sp.mig_g[r]; c.res_mig_m[r] = sp.mig_m[r]; }
        for (int n = 0; n < N; n++) gm[n] = s->node_gpu_memory ? s->node_gpu_memory[prep.perm[n]] : 100;
        c.p_portion = por; c.p_group = grp; c.p_on_group = ogrp; c.n_gpu_mem = gm;
        c.ng_id = own<int32_t>(pool, (size_t)N * KAI_GMAX); for (size_t i = 0; i < (size_t)N * KAI_GMAX; i++) c.ng_id[i] = -1;
        c.ng_used = own<int64_t>(pool, (size_t)N * KAI_GMAX); c.ng_rel = own<int64_t>(pool, (size_t)N * KAI_GMAX); c.ng_alloc = own<int64_t>(pool, (size_t)N * KAI_GMAX);
        c.ng_mark = own<uint32_t>(pool, N); c.ng_has_alloc = own<uint32_t>(pool, N);
        c.next_new_group = own<int32_t>(pool, 1); c.next_new_group[0] = next_new;
    }
    c.s_active_alloc = own<int32_t>(pool, S); c.s_active_used = own<int32_t>(pool, S); c.s_alive = own<int32_t>(pool, S); c.s_gated = own<int32_t>(pool, S); c.s_pipelined = own<int32_t>(pool, S);
    c.j_n_pending = own<int32_t>(pool, J); c.j_tta_valid = own<int32_t>(pool, J); c.j_tta_n = own<int32_t>(pool, J); c.tta = own<int32_t>(pool, P);
    c.j_tta_res = own<double>(pool, (size_t)4 * J); c.j_allocated = own<double>(pool, (size_t)4 * J);
    c.lq_sorted = own<int32_t>(pool, J); c.lq_side = own<int32_t>(pool, J); c.lq_cur = own<int32_t>(pool, Q); c.lq_end = own<int32_t>(pool, Q); c.lq_side_len = own<int32_t>(pool, Q); c.j_state = own<uint8_t>(pool, J);
    c.qheap = own<int32_t>(pool, Q + 1); c.root_heap = own<int32_t>(pool, Q + 1); c.qn = own<QNode>(pool, Q + 1);
    c.ops_cap = 4 * P + 64; c.ops = own<StmtOp>(pool, c.ops_cap); c.out_cap = ((int64_t)2 * P + 64) * (n_actions > 0 ? n_actions : 1); c.out_ops = own<kai_op>(pool, c.out_cap);  // the library gives every action its own 2P + 64; this harness keeps one list for the whole cycle
    c.scratch = own<int32_t>(pool, (size_t)P + 64); c.st = own<EngineState>(pool, 1);
    c.q_share = const_cast<QShare*>(copy(pool, prep.shares.data(), prep.shares.size()));
    };
    alloc1(pool, c);
    auto t0 = std::chrono::steady_clock::now();

    // ---- serial twins of the session-open kernels (kai_kernels.hpp)
    std::vector<int> acct(P); for (int p = 0; p < P; p++) acct[p] = p;
    if (shared) std::sort(acct.begin(), acct.end(), [&](int a, int b) { return s->pod_uid_rank[a] < s->pod_uid_rank[b]; });  // AddTasksToNode order of the fixtures (nodes_fake/nodes.go:289-302): the shared-GPU guards are order sensitive
    for (int p = 0; p < P; p++) c.p_on_node[p] = -1;
    for (int pi = 0; pi < P; pi++) {  // k_node_accounting
        const int p = acct[pi];
        int st = c.p_status[p], n = c.p_node[p];
        if (!st_active_used(st) || n < 0 || n >= N) continue;
        c.p_on_node[p] = n; c.p_on_node_status[p] = st; c.p_accepted[p] = 1;
        if (shared && c.p_shared[p] && c.p_group[p] >= 0) { c.p_on_group[p] = c.p_group[p]; }
        for (int r = 0; r < R; r++) {
            if (shared && r == KAI_RES_GPU && c.p_shared[p]) continue;
            double v = c.p_req[(size_t)r * P + p]; if (v == 0) continue; size_t i = (size_t)r * N + n;
            c.n_used[i] += v;
            if (st == KAI_POD_RELEASING) { c.n_rel[i] += v; c.n_idle[i] -= v; } else if (st == KAI_POD_PIPELINED) c.n_rel[i] -= v; else c.n_idle[i] -= v;
        }
        if (shared && c.p_shared[p] && c.p_on_group[p] >= 0) { SgNode g{c, n}; if (!g.add(st, c.p_mem[p], c.p_on_group[p])) return KAI_ERR_UNSUPPORTED; }
    }
    if (shared) for (int n = 0; n < N; n++) { SgNode g{c, n}; g.refit(); }  // as k_node_accounting_shared
    if (c.plugins & KAI_PLUGIN_PROPORTION) {  // k_total_nodes + k_total_foreign
        for (int n = 0; n < N; n++) {
            uint32_t f = c.n_flags[n]; if (f & KAI_NODE_NOT_READY) continue;
            bool ignore = c.restrict_nodes && !(f & KAI_NODE_GPU_WORKER);
            c.st->total[0] += c.n_alloc[(size_t)KAI_RES_CPU * N + n]; c.st->total[1] += c.n_alloc[(size_t)KAI_RES_MEM * N + n];
            if (!ignore) { c.st->total[2] += c.n_alloc[(size_t)KAI_RES_GPU * N + n]; if (c.mig_on) for (int r = KAI_RES_PODS + 1; r < R; r++) if (c.res_mig_g[r] > 0) c.st->total[2] += (double)c.res_mig_g[r] * c.n_alloc[(size_t)r * N + n]; }
        }
        for (int p = 0; p < P; p++) {
            if (!(c.p_flags[p] & KAI_POD_FOREIGN_SCHEDULER)) continue;
            int n = c.p_on_node[p]; if (n < 0 || !st_active_used(c.p_on_node_status[p]) || (c.n_flags[n] & KAI_NODE_NOT_READY)) continue;
            c.st->total[0] -= c.p_req[(size_t)KAI_RES_CPU * P + p]; c.st->total[1] -= c.p_req[(size_t)KAI_RES_MEM * P + p]; c.st->total[2] -= c.quota_on ? c.p_quota_gpu[p] : c.p_req[(size_t)KAI_RES_GPU * P + p];
        }
    }
    for (int j = 0; j < J; j++) {  // k_job_usage + k_leaf_usage
        double al[3] = {0, 0, 0}, rq[3] = {0, 0, 0}, ja[3] = {0, 0, 0}; int pending = 0;
        for (int i = 0; i < c.j_n_pods[j]; i++) {
            int p = c.j_first_pod[j] + i, st = c.p_status[p], ps = c.p_podset[p];
            if (st_active_allocated(st)) c.s_active_alloc[ps]++;
            if (st_active_used(st)) c.s_active_used[ps]++;
            if (st_alive(st)) c.s_alive[ps]++;
            if (st == KAI_POD_GATED) c.s_gated[ps]++;
            if (st == KAI_POD_PIPELINED) c.s_pipelined[ps]++;
            if (st == KAI_POD_PENDING) pending++;
            double q[3] = {c.p_req[(size_t)KAI_RES_CPU * P + p], c.p_req[(size_t)KAI_RES_MEM * P + p], c.p_req[(size_t)KAI_RES_GPU * P + p]};
            const double qa = c.quota_on ? c.p_acc_gpu[p] : q[2], qp = c.quota_on ? c.p_pend_gpu[p] : q[2], qq = c.quota_on ? c.p_quota_gpu[p] : q[2];  // accepted quota / pending weight / request quota (gpu-memory, MIG)
            if (st_allocated(st)) { for (int k = 0; k < 3; k++) ja[k] += k == 2 ? qq : q[k]; if (c.p_accepted[p]) for (int k = 0; k < 3; k++) { const double v = k == 2 ? qa : q[k]; al[k] += v; rq[k] += v; } }
            else if (st == KAI_POD_PENDING) for (int k = 0; k < 3; k++) rq[k] += k == 2 ? qp : q[k];
        }
        c.j_n_pending[j] = pending;
        for (int k = 0; k < 3; k++) c.j_allocated[(size_t)j * 4 + k] = ja[k];
        if ((c.plugins & KAI_PLUGIN_PROPORTION) && c.j_queue[j] >= 0) for (int k = 0; k < 3; k++) {
            QShare& x = c.q_share[(size_t)c.j_queue[j] * 3 + k];
            x.allocated += al[k]; x.request += rq[k]; if (!c.j_preempt[j]) x.allocated_np += al[k];
        }
    }
    // With fraction pods the queue sums are sums of non-integers and their last bit depends on the order of addition.  The device kernels roll pods
    // up job by job and queue by queue; the reference walks pod by pod up the ancestry in Go-map order (proportion.go:347-401), the oracle in (job,
    // status, pod) order.  KAI_HOSTSIM_POD_ORDER_SUMS=1 makes this harness add in the oracle's order, which takes the order of addition out of a
    // comparison with arbitrary portions (0.2, 0.3 ...): what then still differs is control flow.
    const bool pod_order_sums = shared && (c.plugins & KAI_PLUGIN_PROPORTION) && std::getenv("KAI_HOSTSIM_POD_ORDER_SUMS");
    if (pod_order_sums) {
        for (int q = 0; q < Q; q++) for (int k = 0; k < 3; k++) { QShare& x = c.q_share[(size_t)q * 3 + k]; x.allocated = 0; x.allocated_np = 0; x.request = 0; }
        std::vector<int> ord;
        for (int j = 0; j < J; j++) {
            if (c.j_queue[j] < 0) continue;
            ord.clear(); for (int i = 0; i < c.j_n_pods[j]; i++) ord.push_back(c.j_first_pod[j] + i);
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return c.p_status[a] < c.p_status[b]; });
            for (int p : ord) {
                const int st = c.p_status[p]; const bool al = st_allocated(st) && c.p_accepted[p], pe = st == KAI_POD_PENDING;
                if (!al && !pe) continue;
                for (int q = c.j_queue[j]; q >= 0; q = c.q_parent[q]) for (int k = 0; k < 3; k++) {
                    QShare& x = c.q_share[(size_t)q * 3 + k]; const double v = k == 2 ? (al ? c.p_acc_gpu[p] : c.p_pend_gpu[p]) : c.p_req[(size_t)(k == 0 ? KAI_RES_CPU : KAI_RES_MEM) * P + p];
                    x.request += v; if (al) { x.allocated += v; if (!c.j_preempt[j]) x.allocated_np += v; }
                }
            }
        }
    }
    if (c.plugins & KAI_PLUGIN_PROPORTION) {
        if (!pod_order_sums) for (int i = 0; i < Q; i++) {  // k_tree_usage
            int q = prep.depth_order[i], par = c.q_parent[q]; if (par < 0) continue;
            for (int k = 0; k < 3; k++) { QShare& x = c.q_share[(size_t)q * 3 + k]; QShare& d = c.q_share[(size_t)par * 3 + k]; d.allocated += x.allocated; d.allocated_np += x.allocated_np; d.request += x.request; }
        }
        std::vector<double> weight((size_t)3 * std::max(Q, 1)), rem_amt((size_t)3 * std::max(Q, 1)); std::vector<uint8_t> rem_has((size_t)3 * std::max(Q, 1));
        for (int l = 0; l < prep.n_levels; l++) for (int t = prep.lvl_off[l] * 3; t < prep.lvl_off[l + 1] * 3; t++) {  // k_fair_share
            int par = prep.lvl_parents[t / 3], k = t % 3;
            double total = par == Q ? c.st->total[k] : c.q_share[(size_t)par * 3 + k].fair;
            divide_sibling_set(c, c.q_children + c.q_child_off[par], c.q_child_off[par + 1] - c.q_child_off[par], k, total, c.k_value,
                               weight.data() + (size_t)k * Q, rem_amt.data() + (size_t)k * Q, rem_has.data() + (size_t)k * Q);
        }
    }
    auto fill = [&](kai_queue_share* out) {
        for (int q = 0; q < Q; q++) for (int k = 0; k < 3; k++) { const QShare& x = c.q_share[(size_t)q * 3 + k];
            out[q].fair_share[k] = x.fair; out[q].allocated[k] = x.allocated; out[q].allocated_non_preemptible[k] = x.allocated_np; out[q].request[k] = x.request; out[q].deserved[k] = x.deserved; out[q].max_allowed[k] = x.max_allowed; }
    };
    if (shares_open) fill(shares_open);
    if (c.use_index) for (int b = 0; b < c.NB; b++) for (int k = 0; k < c.C; k++) HostBackend::build_block(c, k, b);  // k_index_build
    auto alloc2 = [&](std::vector<std::vector<char>>& pool, KaiCtx& c) -> int {
    c.use_signatures = cfg->use_scheduling_signatures ? 1 : 0; c.j_signature = s->job_signature ? copy(pool, s->job_signature, J) : nullptr;
    c.j_last_start = s->job_last_start_ns ? copy(pool, s->job_last_start_ns, J) : nullptr; c.q_preempt_mr = s->queue_preempt_min_runtime_ns ? copy(pool, s->queue_preempt_min_runtime_ns, Q) : nullptr;
    c.q_reclaim_mr = s->queue_reclaim_min_runtime_ns ? copy(pool, s->queue_reclaim_min_runtime_ns, Q) : nullptr;
    c.now_ns = cfg->now_ns; c.def_preempt_mr = cfg->default_preempt_min_runtime_ns; c.def_reclaim_mr = cfg->default_reclaim_min_runtime_ns; c.reclaim_method = cfg->reclaim_resolve_method;
    c.max_consolidation_preemptees = cfg->max_consolidation_preemptees; c.allow_consolidating_reclaim = cfg->allow_consolidating_reclaim; c.saturation_multiplier = cfg->reclaimer_saturation_multiplier;
    { size_t bytes = solver_scratch_bytes(N, P, S, J, Q, c.W, c.D + c.T, c.TL, c.G); char* base = own<char>(pool, bytes); solver_scratch_bind(c.sv, base, N, P, S, J, Q, c.W, c.D + c.T, c.TL, c.G); for (int i = 0; i <= c.sv.xr_mask; i++) c.sv.xr_key[i] = -1; c.sv.xr_group = own<int32_t>(pool, (size_t)c.sv.xr_mask + 1); }
    if (int rc = batch_bind(c, prep, [&](size_t bytes) { return (void*)own<char>(pool, bytes); }, [&](void* d, const void* h, size_t n) { std::memcpy(d, h, n); return 0; }, g_sh_world, g_sh_rank, g_sh_k)) return rc;
    if (shared || std::getenv("KAI_HOSTSIM_NO_BATCH")) c.bt.enabled = 0;
        return 0;
    };
    if (int rc = alloc2(pool, c)) return rc;
    struct Rep { std::vector<std::vector<char>> pool; KaiCtx c{}; };
    std::vector<Rep> reps;
    HostBackend be; Engine<HostBackend> eng(c, be);
    int64_t batch_rounds = 0, batch_actions = 0, bucket_actions = 0, counts_actions = 0, levels_actions = 0;
    bool index_stale = false;
    for (int i = 0; i < n_actions; i++) {
        if (actions[i] < KAI_ACTION_ALLOCATE || actions[i] > KAI_ACTION_PREEMPT) return KAI_ERR_UNSUPPORTED;
        if (actions[i] != KAI_ACTION_ALLOCATE && cfg->use_scheduling_signatures && !s->job_signature && J > 0) return KAI_ERR_UNSUPPORTED;
        c.action = actions[i]; { int d = cfg->queue_depth[actions[i]]; c.queue_depth = d > 0 ? d : 0; }
        if (index_stale) { if (c.use_index) for (int b = 0; b < c.NB; b++) for (int k = 0; k < c.C; k++) HostBackend::build_block(c, k, b); index_stale = false; }  // as kai_action_execute: the bucket fill keeps the sets current, not the class index
        for (int j = 0; j < J; j++) { c.j_state[j] = job_init_state(c, j); if (c.j_state[j] != 3 && c.j_n_ps[j] <= 64) eng.ensure_tta(j, true); }  // k_job_init
        for (int q = 0; q < Q; q++) {                                      // k_leaf_init
            int b = c.q_job_off[q], e = c.q_job_off[q + 1], cnt = 0; c.lq_side_len[q] = 0;
            for (int x = b; x < e; x++) { int j = c.jobs_static[x]; int st = c.j_state[j]; if (st == 0) c.lq_sorted[b + cnt++] = j; else if (st != 3) eng.leaf_push(q, j); }
            c.lq_cur[q] = 0; c.lq_end[q] = cnt; qnode_init(c, q, cnt + c.lq_side_len[q]);
        }
        if (std::getenv("KAI_HOSTSIM_PS")) { int ps = std::atoi(std::getenv("KAI_HOSTSIM_PS")); std::fprintf(stderr, "host_sim: before action %d podset %d active_alloc %d used %d alive %d pipelined %d\n", actions[i], ps, c.s_active_alloc[ps], c.s_active_used[ps], c.s_alive[ps], c.s_pipelined[ps]); }
        if (c.st->non_allocate_commits) c.fast_ok = 0;  // as kai_action_execute does after every action
        if (actions[i] != KAI_ACTION_ALLOCATE) {
            if (g_sh_world > 1 && c.use_index) for (int b = 0; b < c.NB; b++) for (int k = 0; k < c.C; k++) HostBackend::build_block(c, k, b);  // node-sharded group: replicated action on the whole index
            // victim actions on several engines (kai_engine_solver.inc solve_partial_multi): every engine a thread on its own replica of the context; what must
            // hold afterwards — every replica committed the same operations and ended in the same state — is checked here on every run
            const int G = shared ? 1 : g_mw_world;
            const bool xsh = !shared && g_sh_world > 1 && g_x_on && g_x_fn;  // the waves of this action over the ranks of the group
            const int xcap = xsh ? (g_x_cap > 0 ? g_x_cap : xw_default_cap(g_sh_world, G)) : 0;
            c.mw_xworld = xsh ? g_sh_world : 0; c.mw_xrank = xsh ? g_sh_rank : 0; c.mw_xcap = xcap; c.mw_mail = nullptr;
            if (xsh) g_xs.begin(g_sh_world, g_sh_rank, xcap);
            if (G <= 1 && !xsh) { c.mw = nullptr; c.mw_rank = 0; c.mw_world = 1; if (std::getenv("KAI_HOSTSIM_FRESH")) { HostBackend bf; Engine<HostBackend> ef(c, bf); ef.execute_victim_action(); ef.flush_index(); } else eng.execute_victim_action(); continue; }
            if (reps.empty() && G > 1) { reps.resize(G - 1); for (auto& r : reps) { alloc1(r.pool, r.c); if (int rc = alloc2(r.pool, r.c)) return rc; if (r.pool.size() != pool.size()) return KAI_ERR_DEVICE_FAULT; } }
            static MultiCtx M; std::memset(&M, 0, sizeof M); M.world = G; M.hit[0] = M.hit[1] = 0x7fffffff;
            c.mw = &M; c.mw_rank = 0; c.mw_world = G;
            for (int w = 1; w < G; w++) {
                Rep& r = reps[w - 1];
                for (size_t k = 0; k < pool.size(); k++) { if (r.pool[k].size() != pool[k].size()) return KAI_ERR_DEVICE_FAULT; std::memcpy(r.pool[k].data(), pool[k].data(), pool[k].size()); }
                r.c.action = c.action; r.c.queue_depth = c.queue_depth; r.c.fast_ok = c.fast_ok; r.c.use_index = c.use_index; r.c.all_tracked = c.all_tracked; r.c.bt.enabled = 0;
                r.c.mw = &M; r.c.mw_rank = w; r.c.mw_world = G;
                r.c.mw_xworld = c.mw_xworld; r.c.mw_xrank = c.mw_xrank; r.c.mw_xcap = c.mw_xcap; r.c.mw_mail = nullptr;
            }
            std::vector<std::thread> th;
            for (int w = 1; w < G; w++) th.emplace_back([&, w] { HostBackend bw; Engine<HostBackend> ew(reps[w - 1].c, bw); ew.execute_victim_action(); ew.flush_index(); });  // (flush_index: what DevBackend::finish does when the kernel ends)
            { HostBackend b0; Engine<HostBackend> e0(c, b0); e0.execute_victim_action(); e0.flush_index(); }
            for (auto& t : th) t.join();
            if (xsh) { HostXIo io{&M}; const int rcx = g_xs.finish(io, (c.st->fault || M.fault) ? 1 : 0); g_x_exchanges += g_xs.exchanges; c.mw_xworld = 0; if (rcx && !c.st->fault) return rcx; }
            g_mw_waves += M.waves; g_mw_sims_run += M.sims_run; g_mw_sims_used += M.sims_used; g_mw_replays += M.replays;
            // (tasks-to-allocate caches: an engine may have filled the cache of a bystander job while another has not — the same content whenever it is computed,
            // job_info.go:253-256 invalidates it with every status change of the job's tasks — so: where both hold one, the same one)
            auto tta_same = [&](const KaiCtx& r) { for (int j = 0; j < J; j++) { if (!r.j_tta_valid[j] || !c.j_tta_valid[j]) continue; if (r.j_tta_n[j] != c.j_tta_n[j] || std::memcmp(r.tta + c.j_first_pod[j], c.tta + c.j_first_pod[j], (size_t)c.j_tta_n[j] * 4) != 0 || std::memcmp(r.j_tta_res + (size_t)j * 4, c.j_tta_res + (size_t)j * 4, 24) != 0) return false; } return true; };
            for (int w = 1; w < G; w++) {  // the engines must agree bit for bit
                const KaiCtx& r = reps[w - 1].c;
                bool same = r.st->out_len == c.st->out_len && r.st->fault == c.st->fault && std::memcmp(r.out_ops, c.out_ops, (size_t)c.st->out_len * sizeof(kai_op)) == 0 && std::memcmp(r.p_status, c.p_status, (size_t)P * 4) == 0 &&
                            std::memcmp(r.p_node, c.p_node, (size_t)P * 4) == 0 && std::memcmp(r.n_idle, c.n_idle, (size_t)R * N * 8) == 0 && std::memcmp(r.n_rel, c.n_rel, (size_t)R * N * 8) == 0 &&
                            std::memcmp(r.q_share, c.q_share, (size_t)Q * 3 * sizeof(QShare)) == 0 && tta_same(r) &&
                            r.st->decisions == c.st->decisions && r.st->simulations == c.st->simulations && r.st->scenarios == c.st->scenarios && r.st->scenarios_filtered == c.st->scenarios_filtered;
                if (!same && std::getenv("KAI_HOSTSIM_DEBUG")) {
                    std::fprintf(stderr, "host_sim: differs: ops %d status %d node %d idle %d rel %d shares %d tta_valid %d | decisions %lld/%lld sims %lld/%lld scen %lld/%lld filt %lld/%lld\n", std::memcmp(r.out_ops, c.out_ops, (size_t)c.st->out_len * sizeof(kai_op)) != 0, std::memcmp(r.p_status, c.p_status, (size_t)P * 4) != 0, std::memcmp(r.p_node, c.p_node, (size_t)P * 4) != 0,
                                 std::memcmp(r.n_idle, c.n_idle, (size_t)R * N * 8) != 0, std::memcmp(r.n_rel, c.n_rel, (size_t)R * N * 8) != 0, std::memcmp(r.q_share, c.q_share, (size_t)Q * 3 * sizeof(QShare)) != 0, std::memcmp(r.j_tta_valid, c.j_tta_valid, (size_t)J * 4) != 0,
                                 (long long)r.st->decisions, (long long)c.st->decisions, (long long)r.st->simulations, (long long)c.st->simulations, (long long)r.st->scenarios, (long long)c.st->scenarios, (long long)r.st->scenarios_filtered, (long long)c.st->scenarios_filtered);
                    for (int j = 0; j < J; j++) if (r.j_tta_valid[j] != c.j_tta_valid[j]) std::fprintf(stderr, "   job %d tta_valid %d vs %d (pending %d)\n", j, r.j_tta_valid[j], c.j_tta_valid[j], c.j_n_pending[j]);
                }
                if (!same) { if (std::getenv("KAI_HOSTSIM_DEBUG")) std::fprintf(stderr, "host_sim: engine %d of %d ended in another state than engine 0 (ops %lld vs %lld, fault %d vs %d)\n", w, G, (long long)r.st->out_len, (long long)c.st->out_len, r.st->fault, c.st->fault); return KAI_ERR_DEVICE_FAULT; }
            }
            c.mw = nullptr; c.mw_world = 1;
            continue;
        }
        {   // the batch path when the action qualifies (kai_batch.hpp), else the sequential engine — as kai_action_execute does
            HostLauncher hl; BatchStats bs; hl.ag_fn = g_sh_fn; hl.ag_user = g_sh_user;
            if (int rc = batch_allocate(hl, c, prep.shape, bs, c.st->out_len, c.st->stmts)) return rc;
            if (bs.ran) {
                if (!bs.st_on_device) { c.st->decisions += bs.decisions; c.st->jobs_attempted += bs.attempted; c.st->jobs_committed += bs.committed; c.st->rollbacks += bs.rollbacks; c.st->out_len += bs.ops; c.st->stmts += bs.committed;
                                        c.st->drain_pending = bs.drain; }
                batch_rounds += bs.rounds; batch_actions++; bucket_actions += bs.buckets ? 1 : 0; counts_actions += bs.buckets >= 2 ? 1 : 0; levels_actions += bs.buckets == 3 ? 1 : 0;
                if (bs.buckets) index_stale = true;
                g_sh_exchanges = bs.exchanges;
            } else {
                // a node-sharded group shards the batch path's fill; every other action runs replicated on every rank (kai_core.hip kai_action_execute)
                if (g_sh_world > 1 && c.use_index) for (int b = 0; b < c.NB; b++) for (int k = 0; k < c.C; k++) HostBackend::build_block(c, k, b);  // the sharded fill left an index of the own slice only
                eng.execute_allocate();
            }
        }
        if (c.st->drain_pending) {                                         // k_drain
            for (int x = 0; x < J; x++) {
                int q = prep.slot_queue[x]; if (q < 0) continue; int pos = x - c.q_job_off[q];
                int64_t att = 0, dec = 0, rb = 0;
                if (pos >= c.lq_cur[q] && pos < c.lq_end[q]) eng.drain_job(c.lq_sorted[x], att, dec, rb);
                if (pos < c.lq_side_len[q]) eng.drain_job(c.lq_side[x], att, dec, rb);
                c.st->jobs_attempted += att; c.st->decisions += dec; c.st->rollbacks += rb; c.st->drained_jobs += att; c.st->drained_decisions += dec;
            }
            c.st->drain_pending = 0;
        }
    }
    if (std::getenv("KAI_HOSTSIM_DEBUG")) std::fprintf(stderr, "host_sim: scenarios %lld simulations %lld filtered %lld attempted %lld committed %lld decisions %lld sim-queue pops %lld | N %d C %d NB %d NSB %d R %d plugins 0x%x\n", (long long)c.st->scenarios, (long long)c.st->simulations, (long long)c.st->scenarios_filtered, (long long)c.st->jobs_attempted, (long long)c.st->jobs_committed, (long long)c.st->decisions, (long long)c.st->prof[4], c.N, c.C, c.NB, c.NSB, c.R, (unsigned)c.plugins);
#ifdef KAI_PROF_VICTIM
    if (std::getenv("KAI_HOSTSIM_DEBUG")) { std::fprintf(stderr, "host_sim prof:"); for (int i = 0; i < KAI_NPROF; i++) std::fprintf(stderr, " %lld", (long long)c.st->prof[i]); std::fprintf(stderr, "\n"); }
#endif
    auto t1 = std::chrono::steady_clock::now();
    if (elapsed_ms_out) *elapsed_ms_out = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (c.st->fault) { if (std::getenv("KAI_HOSTSIM_DEBUG")) std::fprintf(stderr, "host_sim: engine fault %d at line %d\n", c.st->fault, c.st->fault_line); return KAI_ERR_DEVICE_FAULT; }
    if (n_ops) *n_ops = c.st->out_len;
    if (ops_out) { if (c.st->out_len > ops_cap) return KAI_ERR_CAPACITY; std::memcpy(ops_out, c.out_ops, (size_t)c.st->out_len * sizeof(kai_op)); for (int64_t i = 0; i < c.st->out_len; i++) if (ops_out[i].node >= 0) ops_out[i].node = prep.perm[ops_out[i].node]; }
    g_last_victim_stats[0] = c.st->scenarios; g_last_victim_stats[1] = c.st->simulations; g_last_victim_stats[2] = c.st->scenarios_filtered;
    g_last_groups.assign(P, -1);
    for (int p = 0; p < P; p++) if (shared && c.p_shared[p] && st_active_used(c.p_status[p])) g_last_groups[p] = c.p_group[p];
    if (pod_status_out) std::memcpy(pod_status_out, c.p_status, (size_t)P * 4);
    if (pod_node_out) for (int p = 0; p < P; p++) pod_node_out[p] = c.p_node[p] >= 0 ? prep.perm[c.p_node[p]] : -1;
    if (shares_final) fill(shares_final);
    if (nodes_out) for (int n = 0; n < N; n++) { kai_node_state& o = nodes_out[prep.perm[n]]; std::memset(&o, 0, sizeof(kai_node_state)); for (int r = 0; r < R; r++) { o.idle[r] = c.n_idle[(size_t)r * N + n]; o.releasing[r] = c.n_rel[(size_t)r * N + n]; o.used[r] = c.n_used[(size_t)r * N + n]; } }
    if (stats) { std::memset(stats, 0, sizeof(*stats)); stats->decisions = c.st->decisions; stats->node_scans = c.st->node_scans; stats->nodes_scanned = c.st->nodes_scanned; stats->jobs_attempted = c.st->jobs_attempted; stats->jobs_committed = c.st->jobs_committed; stats->rollbacks = c.st->rollbacks; stats->reserved[0] = c.st->index_queries; stats->reserved[1] = c.st->index_refreshes; stats->reserved[2] = c.st->drained_jobs; stats->reserved[3] = c.st->drained_decisions; stats->reserved[4] = batch_actions; stats->reserved[5] = batch_rounds; stats->reserved[6] = bucket_actions; stats->reserved[7] = counts_actions | (levels_actions << 32); }
    return KAI_OK;
}

// ---- kai_victim_shard.hpp on its own (tests/test_dist_gloo.py::test_wave_exchange_protocol): R ranks in one process, the "all-gather" a memcpy between their send buffers.
// mode 0: random waves — every rank must end with the identical merged wave, equal to what one rank running every simulation would hold (returns 0, or the failing check).
// mode 1: rank `arg` leaves the protocol (its closing message with the fault flag) while the others are in a wave — they must see the fault, skip their own closing
//         message, and every rank must have issued the same number of collectives.
extern "C" int kai_hostsim_xw_selftest(int R, int cap_in, int mode, int arg, uint64_t seed) {
    using namespace kai;
    if (R < 1 || R > 16) return -1;
    const int cap = xw_cap(cap_in);
    auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(seed >> 33); };
    struct Rank { XShardHost xs; std::vector<int32_t> res; std::vector<int64_t> cnt; int32_t hdr[8]; int hit = 0, xrun = 0, fault = 0, rc = 0; const unsigned char* gathered = nullptr; };
    std::vector<Rank> rk((size_t)R);
    for (int r = 0; r < R; r++) { rk[r].xs.begin(R, r, cap_in); rk[r].res.assign(KAI_MW_WAVE, 0); rk[r].cnt.assign((size_t)KAI_MW_WAVE * KAI_MW_CNT, 0); }
    // the exchange step for all ranks at once: phase 1 every rank packs (wave or closing message), phase 2 every rank merges the same R messages
    std::vector<unsigned char> all(xw_msg_bytes(cap, R) * (size_t)R);
    struct Io { Rank* me; std::vector<unsigned char>* all; int R; int phase;  // phase 0: publish the send buffer, report "not yet"; the driver below re-runs with phase 1
        int pull(int32_t* hdr, int32_t* res, int64_t* cnt, int b, int cap) { std::memcpy(hdr, me->hdr, sizeof me->hdr); std::memcpy(res, me->res.data(), 4 * (size_t)cap); std::memcpy(cnt, me->cnt.data(), 8 * KAI_MW_CNT * (size_t)cap); (void)b; return 0; }
        int push(int b, const int32_t* res, const int64_t* cnt, int cap, int hit, int xrun, int fault) { (void)b; std::memcpy(me->res.data(), res, 4 * (size_t)cap); std::memcpy(me->cnt.data(), cnt, 8 * KAI_MW_CNT * (size_t)cap); me->hit = hit; me->xrun = xrun; me->fault = fault; return 0; }
        int allgather(const void* s, void* r, int64_t n) { if (phase == 0) { std::memcpy(all->data() + (size_t)me->xs.r * (size_t)n, s, (size_t)n); return 1; } std::memcpy(r, all->data(), (size_t)n * R); return 0; } };
    auto exchange = [&](int b, const std::vector<int>& closing) {  // closing[r] = -1: rank r is in a wave; else its closing message with that fault flag; -2: takes no part (already gone)
        for (int phase = 0; phase < 2; phase++) for (int r = 0; r < R; r++) {
            if (closing[r] == -2) continue;
            Io io{&rk[r], &all, R, phase};
            XShardHost snap = rk[r].xs;  // phase 0 only publishes: run it on a copy so that the sequence number advances once
            XShardHost& x = phase == 0 ? snap : rk[r].xs;
            rk[r].rc = closing[r] == -1 ? x.wave(io, b) : x.finish(io, closing[r]);
        }
    };
    if (mode == 0) {
        for (int wave = 0; wave < 50; wave++) {
            const int b = wave & 1, n_run = (int)(rnd() % (uint32_t)(cap + 1));  // simulations 0 .. n_run-1 of this wave were run by their owners
            const int hit = (rnd() % 3) ? (int)(rnd() % (uint32_t)(n_run + 1)) : 0x7fffffff; const int true_hit = (hit < n_run) ? hit : 0x7fffffff;
            std::vector<int32_t> ref_res(KAI_MW_WAVE, 0); std::vector<int64_t> ref_cnt((size_t)KAI_MW_WAVE * KAI_MW_CNT, 0);
            for (int i = 0; i < n_run; i++) { ref_res[i] = (int32_t)(rnd() & 0x1ff); for (int k = 0; k < KAI_MW_CNT; k++) ref_cnt[(size_t)i * KAI_MW_CNT + k] = (int64_t)rnd() - 1000; }
            for (int r = 0; r < R; r++) {
                Rank& me = rk[r]; std::fill(me.res.begin(), me.res.end(), -7); std::fill(me.cnt.begin(), me.cnt.end(), -7);
                int next = 0, own_hit = 0x7fffffff;
                for (int i = r; i < n_run; i += R) { me.res[i] = ref_res[i]; for (int k = 0; k < KAI_MW_CNT; k++) me.cnt[(size_t)i * KAI_MW_CNT + k] = ref_cnt[(size_t)i * KAI_MW_CNT + k]; next++; if (i == true_hit) own_hit = i; }
                next += (int)(rnd() % 3);  // engines whose last fetch found nothing to run
                me.hdr[0] = 1; me.hdr[1] = 0; me.hdr[2] = me.hdr[3] = 0; me.hdr[4] = me.hdr[5] = next; me.hdr[6] = me.hdr[7] = own_hit;
            }
            exchange(b, std::vector<int>((size_t)R, -1));
            for (int r = 0; r < R; r++) {
                const Rank& me = rk[r];
                if (me.rc) return 10 + r;
                if (me.hit != true_hit || me.fault) return 100 + r;
                if (true_hit == 0x7fffffff && me.xrun < std::min(n_run, cap)) return 200 + r;  // everything that was run counts when nothing hit
                const int upto = true_hit != 0x7fffffff ? true_hit + 1 : std::min(std::min(me.xrun, n_run), cap);
                for (int i = 0; i < upto; i++) { if (me.res[i] != ref_res[i]) return 300 + r; for (int k = 0; k < KAI_MW_CNT; k++) if (me.cnt[(size_t)i * KAI_MW_CNT + k] != ref_cnt[(size_t)i * KAI_MW_CNT + k]) return 400 + r; }
            }
        }
        exchange(0, std::vector<int>((size_t)R, 0));
        for (int r = 0; r < R; r++) { if (rk[r].rc) return 500 + r; if (rk[r].xs.exchanges != 51) return 600 + r; }
        return 0;
    }
    // mode 1: two good waves, then rank `arg` is gone
    for (int r = 0; r < R; r++) { Rank& me = rk[r]; me.hdr[0] = 1; me.hdr[1] = 0; me.hdr[4] = me.hdr[5] = 1; me.hdr[6] = me.hdr[7] = 0x7fffffff; }
    exchange(0, std::vector<int>((size_t)R, -1)); exchange(1, std::vector<int>((size_t)R, -1));
    std::vector<int> closing((size_t)R, -1); closing[arg % R] = 1;
    exchange(0, closing);
    for (int r = 0; r < R; r++) {
        if (r == arg % R) { if (rk[r].rc != 0 && R > 1) return 700 + r; continue; }   // the rank that left: its closing message went out
        if (rk[r].rc != 0 || !rk[r].fault) return 800 + r;                            // the others: the wave came back with the fault flag (their engines give up)
        Io io{&rk[r], &all, R, 1};
        if (rk[r].xs.finish(io, 1) != 0) return 900 + r;                              // ... and their own closing message is NOT sent any more
    }
    for (int r = 0; r < R; r++) if (rk[r].xs.exchanges != 3) return 1000 + r;          // the same number of collectives everywhere
    return 0;
}
