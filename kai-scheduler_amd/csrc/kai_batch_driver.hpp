// kai_batch_driver.hpp — host side of the batch path: buffers, qualification and the round loop (plan → fill → apply), written once against
// a Launcher (kai_core.hip: HIP launches on the session's stream; tests/host_sim: the lock-step emulator of kai_simt.hpp).
//
// Launcher interface:  void <kernel>(grid, block, args...)  for every kernel of kai_batch_kernels.hpp,
//                      int read(void* host_dst, const void* dev_src, size_t bytes)   (synchronises),
//                      int write(void* dev_dst, const void* host_src, size_t bytes).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kai_host_prep.hpp"

namespace kai {

struct BatchStats {
    int64_t rounds = 0, mismatches = 0, planned = 0;
    int64_t decisions = 0, attempted = 0, committed = 0, rollbacks = 0, ops = 0;
    int64_t fill_cycles = 0, fill_load = 0, fill_update = 0, fill_rescan = 0, block_loads = 0, rescans1 = 0, rescans2 = 0, rescans3 = 0;
    int32_t drain = 0, ran = 0, max_h = 0, buckets = 0;  // buckets: the fill ran on kai_fill_buckets.hpp
    int32_t st_on_device = 0, dev_loop = 0;  // the sums are in the engine's state already (k_round_finish): the caller adds nothing; dev_loop: the round loop's state lived on the device
    int64_t exchanges = 0;  // node-sharded group: all-gathers of the action
};

// pools: every queued job can be an element of its leaf and of every ancestor (incl. the virtual root)
inline size_t batch_pool_e(const HostPrep& prep, int J, int Q) { return (size_t)J * (size_t)prep.n_heights + 2 * (size_t)Q + 64; }
inline size_t batch_pool_k(const HostPrep& prep, int J, int Q) { return (size_t)J * (size_t)prep.n_heights + 2 * (size_t)Q + 64; }

// zalloc(bytes) -> zero-filled memory the kernels can address; upload(dst, src, bytes)
template <class ZAlloc, class Upload>
int batch_bind(KaiCtx& c, const HostPrep& prep, ZAlloc&& zalloc, Upload&& upload, int world = 1, int rank = 0, int shard_k = 0) {
    BatchCtx& b = c.bt;
    b = BatchCtx{};
    b.enabled = prep.batch_ok && c.use_index && c.all_tracked && c.fast_ok && c.R <= 4 && (c.plugins & KAI_PLUGIN_PROPORTION) ? 1 : 0;
    if (!b.enabled) return 0;
    const int Q = c.Q, J = c.J, P = c.P;
    b.n_h = prep.n_heights; b.pool_e = (int32_t)batch_pool_e(prep, J, Q); b.pool_k = (int32_t)batch_pool_k(prep, J, Q);
#define KB_Z2(field, n) do { void* p_ = zalloc(std::max<size_t>((size_t)(n), 1) * sizeof(*b.field)); if (!p_) return KAI_ERR_HIP; b.field = (decltype(b.field))p_; } while (0)
#define KB_Z(field, n) do { void* p_ = zalloc(std::max<size_t>((size_t)(n), 1) * sizeof(*b.field)); if (!p_) return KAI_ERR_HIP; b.field = (decltype(b.field))p_; } while (0)
    KB_Z(q_height, Q + 1); KB_Z(h_off, b.n_h + 1); KB_Z(h_nodes, Q + 1); KB_Z(q_srank, Q + 1); KB_Z(q_islot, Q + 1);
    KB_Z(j_clsmask, J); KB_Z(j_ucls, J); KB_Z(cur_sp, Q + 1); KB_Z(qual, 4);
    KB_Z(q_cnt, Q + 1); KB_Z(q_ebase, Q + 1); KB_Z(q_kbase, Q + 1); KB_Z(q_valid, Q + 1); KB_Z(q_nk, Q + 1); KB_Z(q_sent, Q + 1); KB_Z(q_taken, Q + 1); KB_Z(q_complete, Q + 1); KB_Z(plan_tot, 2);
    KB_Z(pk, b.pool_k); KB_Z(sp, b.pool_k); KB_Z(k_owner, b.pool_k);
    KB_Z(el_leaf, b.pool_e); KB_Z(el_ck, b.pool_e); KB_Z(el_next, b.pool_e); KB_Z(e_job, b.pool_e); KB_Z(e_grank, b.pool_e); KB_Z(e_flag, b.pool_e);
    KB_Z(d_res, (size_t)b.pool_e * 3); KB_Z(d_meta, b.pool_e); KB_Z(d_spres, (size_t)b.pool_k * 3); KB_Z(d_spj, b.pool_k);
    { const size_t n_sg = (size_t)b.pool_e / KPS_SEG + (size_t)Q + 2; KB_Z(sg_tot, n_sg * 6); KB_Z(sg_key, n_sg); KB_Z(sg_kvalid, n_sg); KB_Z(sg_fb, Q + 1); KB_Z(d_ab, (size_t)b.pool_k * 3); }
    KB_Z(g_job, J + 1); KB_Z(g_opoff, J + 1); KB_Z(g_stmt, J + 1); KB_Z(g_first, J + 1); KB_Z(g_nt, J + 1); KB_Z(g_ucls, J + 1); KB_Z(g_flag, J + 1); KB_Z(g_out, J + 1);
    KB_Z(t_cls, P); KB_Z(t_node, P);
    KB_Z(nrec, (size_t)c.NB * KAI_BLOCK); KB_Z(fs, 1); KB_Z(dead_mask, 1); KB_Z(cls_cap, 64); KB_Z(ctl, 1);
    KB_Z(bk_words, (size_t)KBK_GMAX * c.NB); KB_Z(bk_ok, (size_t)std::max(c.C, 1) * c.NB); KB_Z(bk_meta, sizeof(BucketMeta) / 4);
    if (world > 1) {  // node-axis sharding: contiguous 64-node-block ranges in name-rank order, offers of K nodes per class and rank
        b.world = world; b.rank = rank;
        const int per = (c.NB + world - 1) / world;  // blocks per rank
        b.n_lo = std::min(c.N, rank * per * KAI_BLOCK); b.n_hi = std::min(c.N, (rank + 1) * per * KAI_BLOCK);
        b.shard_k = shard_k > 0 ? shard_k : 128;
        b.shard_mmax = std::max(1, std::min(std::max(c.C, 1) * b.shard_k, std::max(per * KAI_BLOCK, 1)));  // the same on every rank: the messages of an all-gather have one size
        b.msg_bytes = (int64_t)((sizeof(ShardHdr) + (size_t)b.shard_mmax * 4 + (size_t)b.shard_mmax * sizeof(NodeRec) + 63) & ~(size_t)63);
        b.vcap = ((b.shard_mmax * world + KAI_BLOCK - 1) / KAI_BLOCK + 1) * KAI_BLOCK;
        KB_Z2(sh_keys, (size_t)std::max(c.C, 1) * std::max(c.N, 1)); KB_Z2(cand_bits, (size_t)(c.N + 31) / 32 + 1);
        KB_Z2(send, (size_t)b.msg_bytes); KB_Z2(recv, (size_t)b.msg_bytes * world);
        KB_Z2(vrec, (size_t)b.vcap); KB_Z2(vmap, (size_t)b.vcap); KB_Z2(v1k, (size_t)std::max(c.C, 1) * (b.vcap / KAI_BLOCK)); KB_Z2(v1n, (size_t)std::max(c.C, 1) * (b.vcap / KAI_BLOCK));
        KB_Z2(floors, 64); KB_Z2(vstate, 4);
        b.nrec_home = b.nrec;
    } else { b.world = 1; b.rank = 0; b.n_lo = 0; b.n_hi = c.N; }
    if (int rc = upload((void*)b.q_height, prep.q_height.data(), prep.q_height.size() * 4)) return rc;
    if (int rc = upload((void*)b.h_off, prep.h_off.data(), prep.h_off.size() * 4)) return rc;
    if (int rc = upload((void*)b.h_nodes, prep.h_nodes.data(), prep.h_nodes.size() * 4)) return rc;
    b.n_inner = prep.n_inner;
    if (!prep.q_islot.empty()) if (int rc = upload((void*)b.q_islot, prep.q_islot.data(), prep.q_islot.size() * 4)) return rc;
#undef KB_Z
#undef KB_Z2
    return 0;
}

// LDS of the fill kernel: super-block level always, block level when it fits beside it
inline size_t batch_fill_lds(const KaiCtx& c, int& l1_in_lds) {
    const size_t l2 = (size_t)c.C * c.NSB * sizeof(IdxE), l1 = (size_t)c.C * c.NB * sizeof(IdxE);
    const size_t budget = 160 * 1024 - 16 * 1024;  // static LDS of the kernel (rollback list: 8 KB) + margin
    l1_in_lds = (l2 + l1 <= budget && !std::getenv("KAI_BATCH_L1_HBM")) ? 1 : 0;  // the variable forces the HBM variant (tests)
    return l2 + (l1_in_lds ? l1 : 0) + 16;
}

// The bucket fill (kai_fill_buckets.hpp) takes the action when k_bucket_build's proof holds for every node and the sets fit the LDS of one CU.
inline bool batch_bucket_params(const KaiCtx& c, const BucketMeta& m, BucketParams& bp, size_t& dyn) {
    bp = BucketParams{}; dyn = 0;
    if (m.bad || c.C < 1 || c.C > 64 || !(c.plugins & KAI_PLUGIN_NODEPLACEMENT)) return false;
    bp.levels = std::max(1, (int)m.max_free); bp.nw = c.NB; bp.nw1 = (c.NB + 63) / 64; bp.n_ok = 0;
    if (bp.levels > KBK_GMAX || bp.nw1 > 64) return false;
    for (int k = 0; k < 64; k++) bp.okslot[k] = -1;
    for (int k = 0; k < c.C; k++) if (m.ok_miss[k]) bp.okslot[k] = (int8_t)bp.n_ok++;
    dyn = ((size_t)bp.levels * bp.nw + (size_t)bp.levels * bp.nw1 + KBK_GMAX + (size_t)bp.n_ok * bp.nw) * 8 + 16;
    return dyn <= (size_t)(160 - 16) * 1024;  // beside the kernel's static LDS (rollback list: 8 KB) and a margin
}

// The fill of one planned round on a node-sharded group: offers -> all-gather -> the same virtual fill on every rank -> own records
// back, until the round's order is used up, a job ends differently from its prediction, or — the usual case — a class runs out of offered
// nodes that beat what the ranks hold back, which starts the next exchange.  fs ends up describing the whole round (like a single fill).
template <class L>
int batch_fill_sharded(L& l, KaiCtx& c, RoundParams rp, FillStatus& fs, int64_t& exchanges) {
    const int TB = 256; BatchCtx& b = c.bt;
    KaiCtx cv = c;  // the virtual cluster: same classes and plugins, its own records / block index; its size lives on the device (vstate)
    cv.bt.nrec = b.vrec; cv.sum1_key = b.v1k; cv.sum1_node = b.v1n; cv.N = 0x7fffffff; cv.NB = b.vcap / KAI_BLOCK; cv.NSB = (cv.NB + 63) / 64;
    int l1v = 0; size_t dynv = batch_fill_lds(cv, l1v); l1v = 0; dynv = (size_t)cv.C * cv.NSB * sizeof(IdxE) + 16;  // block level of the virtual index stays in HBM
    const int nloc = b.n_hi - b.n_lo, blk0 = b.n_lo / KAI_BLOCK, blk1 = (b.n_hi + KAI_BLOCK - 1) / KAI_BLOCK;
    FillStatus acc{}; acc.n_done = rp.start; int start = rp.start, stalled = 0;
    for (;;) {
        l.shard_keys(std::max(1, (int)(((int64_t)std::max(nloc, 1) * std::max(c.C, 1) + TB - 1) / TB)), TB, c);
        if (c.C) l.shard_select(c.C, TB, c);
        l.shard_compact(1, TB, c);
        if (int rc = l.allgather((const void*)b.send, (void*)b.recv, b.msg_bytes)) return rc;
        exchanges++;
        l.shard_vbuild(b.world + 1, TB, c);
        l.index_from_recs(std::max(1, cv.NB), 64, cv, (const NodeRec*)b.vrec, -1, (uint64_t*)b.v1k, (int32_t*)b.v1n, cv.NB, 0, cv.NB);
        RoundParams r2 = rp; r2.start = start; r2.ops0 = (int32_t)acc.ops; r2.stmt0 = (int32_t)acc.committed; if (rp.mode == 0) r2.mode = 2;
        l.fill(1, 64, dynv, cv, r2, l1v);
        FillStatus f1{};
        if (int rc = l.read(&f1, (const void*)b.fs, sizeof f1)) return rc;
        if (std::getenv("KAI_SHARD_DEBUG")) { int32_t vs[4] = {0, 0, 0, 0}; IdxE fl[4]; (void)l.read(vs, (const void*)b.vstate, sizeof vs); (void)l.read(fl, (const void*)b.floors, sizeof fl);
            std::fprintf(stderr, "[shard r%d] exchange %lld mode %d start %d -> n_done %d planned %d floor_stop %d mismatch %d decisions %lld | virtual nodes %d floor0 %llx/%d\n", b.rank, (long long)exchanges, r2.mode, start, f1.n_done, f1.planned, f1.floor_stop, f1.mismatch, (long long)f1.decisions, vs[0], (unsigned long long)fl[0].key, fl[0].node); }
        l.shard_scatter(std::max(1, (b.vcap + TB - 1) / TB), TB, c, b.vcap);
        if (blk1 > blk0) l.index_from_recs(blk1 - blk0, 64, c, (const NodeRec*)b.nrec, c.N, (uint64_t*)c.sum1_key, (int32_t*)c.sum1_node, c.NB, blk0, blk1);
        acc.decisions += f1.decisions; acc.attempted += f1.attempted; acc.committed += f1.committed; acc.rollbacks += f1.rollbacks; acc.ops += f1.ops;
        acc.cycles_total += f1.cycles_total; acc.block_loads += f1.block_loads; acc.rescans1 += f1.rescans1; acc.rescans2 += f1.rescans2; acc.rescans3 += f1.rescans3;
        acc.mismatch = f1.mismatch; acc.all_dead = f1.all_dead; acc.dead_mask = f1.dead_mask; acc.planned = f1.planned; acc.floor_stop = f1.floor_stop;
        if (rp.mode == 1) break;
        stalled = f1.n_done == start ? stalled + 1 : 0;
        start = f1.n_done; acc.n_done = start;
        if (f1.mismatch || !f1.floor_stop || start >= f1.planned) break;
        if (stalled >= 2) return KAI_ERR_COMM;  // a gang larger than the ranks' offers can hold: does not qualify for the sharded path
    }
    fs = acc;
    return l.write((void*)b.fs, &acc, sizeof acc);  // the apply kernels read the round's totals
}

// Runs the allocate action on the batch path.  ran = false: the action does not qualify, nothing was touched (run the sequential engine).
// On return with ran: out_len / counters are in bs; drain = the remaining queue is to be resolved by k_drain (no class fits anywhere).
inline int batch_policy() { const char* e = std::getenv("KAI_BATCH_POLICY"); const int v = e ? std::atoi(e) : 4; return v >= 0 && v <= 4 ? v : 4; }
template <class L>
int batch_allocate(L& l, KaiCtx& c, const HostPrep::BatchShape& shape, BatchStats& bs, int64_t ops_base0 = 0, int64_t stmt_base0 = 0) {
    bs = BatchStats{};
    c.bt.dev_loop = 0;
    if (!c.bt.enabled || c.action != KAI_ACTION_ALLOCATE || c.queue_depth > 0 || !c.fast_ok) return 0;
    const int TB = 256, Q = c.Q, J = c.J;
    int32_t qual[4] = {0, 0, 0, 0};
    if (int rc = l.zero((void*)c.bt.qual, sizeof qual)) return rc;
    if (Q) { l.static_rank((Q + TB - 1) / TB, TB, c); l.static_check((Q + TB - 1) / TB, TB, c); }
    if (J) l.qualify((J + TB - 1) / TB, TB, c);
    // (behind the qualification, before its verdict is read: the node records and the sets' build write arrays of this path only, and one drain of the stream then answers both)
    const bool sharded = c.bt.world > 1;
    const bool try_sets = !sharded && c.C >= 1 && !std::getenv("KAI_FILL_GENERAL");
    l.nrec(std::max(1, (c.NB * KAI_BLOCK + TB - 1) / TB), TB, c);
    if (try_sets) {
        if (int rc = l.zero((void*)c.bt.bk_meta, sizeof(BucketMeta))) return rc;
        l.bucket_build(std::max(1, (c.NB * KAI_BLOCK + TB - 1) / TB), TB, c);
    }
    if (int rc = l.read(qual, (const void*)c.bt.qual, sizeof qual)) return rc;
    if (qual[0] || qual[1]) return 0;
    if (c.bt.world > 1 && qual[3] > c.bt.shard_k) return 0;  // a gang that needs more nodes of one rank than a rank offers per exchange would stall mid-action: this group does not take the action
    bs.ran = 1;
    int remaining = qual[2];
    int l1_in_lds = 0; const size_t dyn = batch_fill_lds(c, l1_in_lds);
    RoundParams rp{}; rp.mode = 1;
    FillStatus fs{};
    // bin-packed GPU classes on nodes where only the devices can bind: the fill over sets of nodes by free devices, all in LDS (kai_fill_buckets.hpp);
    // KAI_FILL_GENERAL=1 keeps the general kernel (A/B runs, tests)
    bool buckets = false; BucketParams bp{}; size_t dyn_bk = 0;
    if (try_sets) {
        BucketMeta m{};
        if (int rc = l.read(&m, (const void*)c.bt.bk_meta, sizeof m)) return rc;
        buckets = batch_bucket_params(c, m, bp, dyn_bk);
    }
    const bool dev_loop = !sharded && !std::getenv("KAI_BATCH_HOST_LOOP");  // rounds without the host (below)
    bs.dev_loop = dev_loop ? 1 : 0;
    // ... and, when no class carries a static bitmap of its own, as two wavefronts side by side: the planned order over the levels' populations, the sets behind a command ring
    // (kai_fill_counts.hpp); KAI_FILL_ONE_WAVE=1 keeps the one-wave kernel (A/B runs, tests)
    const bool counts = buckets && bp.n_ok == 0 && !std::getenv("KAI_FILL_UNBATCHED") && !std::getenv("KAI_FILL_ONE_WAVE") && dyn_bk + sizeof(FcLds) <= (size_t)(160 - 16) * 1024;
    // ... and, up to eight levels, with a wavefront per level behind the counting machine (kai_fill_levels.hpp); KAI_FILL_TWO_WORKERS=1 keeps the kernel of kai_fill_counts.hpp (A/B runs, tests)
    const bool levels = counts && bp.levels <= KFL_LMAX && !std::getenv("KAI_FILL_TWO_WORKERS") && dyn_bk + sizeof(FlLds) <= (size_t)(160 - 16) * 1024;
    const int fill_tb = levels ? 64 * (bp.levels + 2) : 256;  // the counting machine, a worker per level, the bookkeeper
    bs.buckets = buckets ? (levels ? 3 : counts ? 2 : 1) : 0;
    if (sharded) { l.shard_mask_nrec(std::max(1, (c.NB * KAI_BLOCK + TB - 1) / TB), TB, c);
                   const int b0 = c.bt.n_lo / KAI_BLOCK, b1 = (c.bt.n_hi + KAI_BLOCK - 1) / KAI_BLOCK; (void)b0; (void)b1;
                   l.index_from_recs(std::max(1, c.NB), 64, c, (const NodeRec*)c.bt.nrec, c.N, (uint64_t*)c.sum1_key, (int32_t*)c.sum1_node, c.NB, 0, c.NB);
                   if (int rc = batch_fill_sharded(l, c, rp, fs, bs.exchanges)) return rc; }
    else { if (levels) l.fill_levels(1, fill_tb, dyn_bk, c, rp, bp); else if (counts) l.fill_counts(1, 256, dyn_bk, c, rp, bp); else if (buckets) l.fill_buckets(1, 256, dyn_bk, c, rp, bp); else l.fill(1, 64, dyn, c, rp, l1_in_lds); if (!dev_loop) if (int rc = l.read(&fs, (const void*)c.bt.fs, sizeof fs)) return rc; }  // (k_round_init reads the verdict on the device)
    int H = 256;  // jobs a leaf offers per round: everything a usual leaf holds (a plan is cheap next to the rounds a short one costs); halved while most of a plan is thrown away
    if (const char* e = std::getenv("KAI_BATCH_H0")) { const int v = std::atoi(e); if (v >= 8) H = v; }
    int64_t ops_base = ops_base0, stmt_base = stmt_base0;
    const bool capacity = !sharded && !std::getenv("KAI_BATCH_NO_CAPACITY");  // (a rank of a node-sharded group sees its own nodes only: no capacity prediction there)
    c.bt.cap_on = capacity ? 1 : 0;
    if (capacity) if (int rc = l.zero((void*)c.bt.cls_cap, 64 * sizeof(int32_t))) return rc;
    if (!capacity) { int32_t inf[64]; for (int k = 0; k < 64; k++) inf[k] = 0x7fffffff; if (int rc = l.write((void*)c.bt.cls_cap, inf, sizeof inf)) return rc; }
    // one round's plan / fill / apply kernels for a plan that looks H jobs into every leaf with at most `left` jobs queued (upper bounds when the loop's state lives on the device:
    // the per-slot kernels leave beyond what k_plan_setup laid out)
    // positions per node from which a height's streams are cut into segments (KAI_PLAN_SEG_MIN: A/B runs, tests; the emulator's default is small so that the tests' clusters take both forms)
#if defined(__HIPCC__)
    int64_t seg_min = 4096;
#else
    int64_t seg_min = 96;
#endif
    if (const char* e = std::getenv("KAI_PLAN_SEG_MIN")) { const long long v = std::atoll(e); if (v >= 1) seg_min = v; }
    auto enqueue_round = [&](int H, int left, int64_t ops_base, int64_t stmt_base) -> int {
        // tasks of every scan class the cluster still holds (for the plan's prediction of gangs that no longer fit; the fill verifies every prediction, so this only
        // saves rounds): summed here, read by k_plan_leaf, zeroed again by k_plan_emit
        if (capacity && c.C >= 1) l.class_capacity(std::max(1, c.NB), 64, c, buckets ? 1 : 0, bp.levels);
        rp = RoundParams{}; rp.h_leaf = H; rp.mode = 0; rp.pad2 = std::getenv("KAI_FILL_UNBATCHED") ? 1 : 0;
        const int64_t e_bound = std::min<int64_t>(left, (int64_t)shape.n_leaves * H);
        const int64_t slots = std::min<int64_t>(c.bt.pool_k, e_bound * shape.n_heights + Q + 1);
        rp.n_slots = (int32_t)slots;
        l.plan_setup(1, KB_PLAN_SETUP_THREADS, c, rp);
        l.plan_leaf(std::max(Q, 1), 64, c, rp);
        for (int h = 1; h < shape.n_heights; h++) {
            rp.height = h;
            l.plan_rank(std::max(1, (int)((slots + TB - 1) / TB)), TB, c, rp);
            if (h + 1 < shape.n_heights) l.plan_gather(std::max(1, (int)((slots + TB - 1) / TB)), TB, c, rp);  // (the virtual root's stream is not scanned)
            // one workgroup per queue node of this height, sized by what a node's stream can hold this round: a long stream is bound by the keys' f64 divisions (more wavefronts
            // hide more of them: r06g, 38 k positions per node: 1.03 ms at 512 threads x 1 position, 0.48 ms at 1 024 x 4), a short one by the barriers of its few steps
            const int64_t per_node = e_bound / std::max(shape.h_count[h], 1);
            // ... and a height of few nodes with long streams is cut into segments of KPS_SEG positions, a workgroup each, in four launches (kai_plan_segments.hpp)
            const int64_t segs = (e_bound + 1 + KPS_SEG - 1) / KPS_SEG;
            if (h + 1 < shape.n_heights && per_node >= seg_min && segs * shape.h_count[h] <= 32768) {
                const int g = (int)(segs * shape.h_count[h]);
                l.seg_sum(g, KPS_T, c, rp, (int)segs); l.seg_gate(g, KPS_T, c, rp, (int)segs); l.seg_keys(g, KPS_T, c, rp, (int)segs); l.seg_max(g, KPS_T, c, rp, (int)segs);
                continue;
            }
            const int scan_tb = std::min(KB_PLAN_SCAN_THREADS, per_node >= 8192 ? 1024 : per_node >= 1024 ? 512 : per_node >= 192 ? 128 : 64);
            l.plan_scan(std::max(shape.h_count[h], 1), scan_tb, c, rp);
        }
        l.plan_emit(std::max(1, (int)((e_bound + TB - 1) / TB)), TB, c);
        if (sharded) { rp.start = 0; if (int rc = batch_fill_sharded(l, c, rp, fs, bs.exchanges)) return rc; }
        else if (levels) l.fill_levels(1, fill_tb, dyn_bk, c, rp, bp);
        else if (counts) l.fill_counts(1, 256, dyn_bk, c, rp, bp);
        else if (buckets) l.fill_buckets(1, 256, dyn_bk, c, rp, bp);
        else l.fill(1, 64, dyn, c, rp, l1_in_lds);
        l.apply_jobs(std::max(1, (int)((e_bound + TB - 1) / TB)), TB, c, ops_base, stmt_base);
        if (Q) l.apply_nodes((Q + TB - 1) / TB, TB, c);
        return 0;
    };
    const int policy = batch_policy();
    const bool trace = std::getenv("KAI_BATCH_TRACE") != nullptr;
    if (dev_loop) {
        // Rounds without the host.  The loop's state (RoundCtl) lives on the device: k_round_next closes a round behind its apply kernels, the next round's kernels read how far to plan
        // and where their output starts from it, and leave at once when it says done.  The host only keeps the stream fed: it enqueues round r while round r-1 runs, sized by the state
        // after round r-2 — which it reads from a pinned copy, stream-ordered, without ever draining the stream (a drain per round was 0.1 ms of idle device: `profiles/r06l_*`,
        // 0.86 of config 5's 24.3 ms and a quarter of config 2's 1.5).  One round is enqueued in vain at the end (its kernels find `done` and leave).
        c.bt.dev_loop = 1;
        l.round_init(c, remaining, H, policy, ops_base, stmt_base);
        if (int rc = l.round_post(0, (const void*)c.bt.ctl, sizeof(RoundCtl))) return rc;
        RoundCtl seen{}; seen.H = H; seen.remaining = remaining;
        const int64_t max_rounds = (int64_t)remaining + 2;  // (every round executes at least one job)
        int64_t printed = 0;
        for (int64_t r = 1; remaining > 0; r++) {  // (nothing queued: no round at all)
            if (r >= 2) {
                if (int rc = l.round_wait((int)((r - 2) % KB_ROUND_SLOTS), &seen, sizeof seen)) return rc;
                if (seen.fault) return KAI_ERR_DEVICE_FAULT;
                if (trace && seen.rounds > printed) { printed = seen.rounds; std::fprintf(stderr, "kai batch round %lld: H %d planned %d executed %d mismatch %d decisions %lld steps %lld committed %lld remaining %d\n", (long long)seen.rounds, seen.last_h, seen.last_planned, seen.last_done, seen.last_mismatch, (long long)seen.last_decisions, (long long)seen.last_steps, (long long)seen.last_committed, seen.remaining); if (seen.last_mismatch) std::fprintf(stderr, "kai batch round %lld: mispredicted job: predicted %d actual %d tasks %d class %d\n", (long long)seen.rounds, seen.mm_flag, seen.mm_out, seen.mm_nt, seen.mm_cls); }
                if (seen.done) break;
                if (r > max_rounds) return KAI_ERR_DEVICE_FAULT;
            }
            const int h_ub = r >= 2 ? kb_round_policy(policy, seen.H, false, 0, 0) : H;  // round r-1 may have widened the plan
            l.round_begin((int)(r % KB_ROUND_SLOTS));
            if (int rc = enqueue_round(h_ub, seen.remaining, 0, 0)) return rc;
            l.round_next(c);
            if (int rc = l.round_post((int)(r % KB_ROUND_SLOTS), (const void*)c.bt.ctl, sizeof(RoundCtl))) return rc;
        }
        l.round_finish(c);
        c.bt.dev_loop = 0;
        bs.rounds = seen.rounds; bs.mismatches = seen.mismatches; bs.planned = seen.planned; bs.max_h = seen.max_h;
        bs.decisions = seen.decisions; bs.attempted = seen.attempted; bs.committed = seen.committed; bs.rollbacks = seen.rollbacks; bs.ops = seen.ops;
        bs.fill_cycles = seen.fill_cycles; bs.fill_load = seen.fill_load; bs.fill_update = seen.fill_update; bs.fill_rescan = seen.fill_rescan;
        bs.block_loads = seen.block_loads; bs.rescans1 = seen.rescans1; bs.rescans2 = seen.rescans2; bs.rescans3 = seen.rescans3;
        bs.drain = seen.drain; bs.st_on_device = 1;
        return 0;
    }
    // the loop on the host (a node-sharded group, whose fill is a host-driven exchange loop of its own; KAI_BATCH_HOST_LOOP=1: A/B runs, tests): one drain of the stream per round
    while (remaining > 0) {
        if (fs.all_dead) { bs.drain = 1; break; }
        if (int rc = enqueue_round(H, remaining, ops_base, stmt_base)) return rc;
        if (!sharded) { if (int rc = l.read(&fs, (const void*)c.bt.fs, sizeof fs)) return rc; } else if (int rc = l.read(nullptr, nullptr, 0)) return rc;
        if (fs.n_done <= 0) return KAI_ERR_DEVICE_FAULT;  // a round always executes at least one job
        bs.rounds++; bs.mismatches += fs.mismatch; bs.planned += fs.planned; bs.max_h = std::max(bs.max_h, H);
        bs.decisions += fs.decisions; bs.attempted += fs.attempted; bs.committed += fs.committed; bs.rollbacks += fs.rollbacks; bs.ops += fs.ops;
        bs.fill_cycles += fs.cycles_total; bs.fill_load += fs.cycles_load; bs.fill_update += fs.cycles_update; bs.fill_rescan += fs.cycles_rescan;
        bs.block_loads += fs.block_loads; bs.rescans1 += fs.rescans1; bs.rescans2 += fs.rescans2; bs.rescans3 += fs.rescans3;
        if (trace) std::fprintf(stderr, "kai batch round %lld: H %d planned %d executed %d mismatch %d decisions %lld steps %lld committed %lld remaining %d\n", (long long)bs.rounds, H, fs.planned, fs.n_done, fs.mismatch, (long long)fs.decisions, (long long)fs.rescans2, (long long)fs.committed, remaining - fs.n_done);
        ops_base += fs.ops; stmt_base += fs.committed; remaining -= fs.n_done;
        // How far the next plan looks.  A round without a surprise: back to the full depth at once (a leaf rarely holds more than 256 queued jobs, so that plan covers the whole
        // queue; climbing back by doubling cost config 5 two rounds of ~0.6 ms each); a plan mostly thrown away: a QUARTER as far (the next surprise is usually close: the short
        // plans in between are the cheaper the shorter they are).  Measured against the rule of rounds 1-4 (x2 up, /2 down): config 5 45.1 -> 42.9 ms, config 3 100.2 -> 99.0,
        // config 2 2.13 -> 1.89 (`profiles/r05s_*`).  KAI_BATCH_POLICY selects the other rules of that A/B run (0: x2 up, /2 down; 1: x4 up; 2: full depth at once, /2 down;
        // 3: x4 up, /4 down).  (kb_round_policy, kai_batch_kernels.hpp: the same rule for the loop on the device.)
        H = kb_round_policy(policy, H, fs.mismatch != 0, fs.n_done, fs.planned);
    }
    return 0;
}

}  // namespace kai
