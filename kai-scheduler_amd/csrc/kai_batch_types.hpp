// kai_batch_types.hpp — data of the batch path (kai_batch.hpp explains the algorithm); included by kai_engine.hpp because the context carries it.
#pragma once
#include <stdint.h>

namespace kai {

struct PlanKey { uint64_t w0, w1, w2, w3; };  // ascending = popped first
enum : uint8_t { BF_OK = 0, BF_GATE = 1, BF_DEAD = 2 };  // predicted outcome of an attempt: committed / fails the capacity gate / finds no node

struct FillStatus {  // written by the fill kernel, read by the host between rounds
    int32_t n_done;        // jobs of the planned order that were executed (the last one may be the mismatch)
    int32_t mismatch;      // 1 = job n_done-1 ended differently from its prediction
    int32_t all_dead;      // no class has a fitting node at the committed state
    int32_t planned;       // length of the planned order of this round
    int32_t floor_stop, pad;  // node-sharded fill: stopped because a class's best candidate no longer beats what the ranks hold back
    int64_t decisions, attempted, committed, rollbacks, ops;  // of this round (committed = Statements of the round: every committed job has operations)
    uint64_t dead_mask;    // classes without a fitting node at the committed state
    int64_t cycles_total, cycles_load, cycles_update, cycles_rescan;  // fill-wave clocks (profiling)
    int64_t block_loads, rescans1, rescans2, rescans3;
};

// The round loop's state on the device (rounds without the host): what batch_allocate's loop kept in host variables between rounds — how far the next plan looks, what is left of the
// queue, where the round's operations / Statements start in the action's output — and the action's sums of the rounds' FillStatus.  k_round_next closes a round on it; every plan /
// fill / apply kernel of a round the host enqueued ahead leaves at once when `done` is set.  The host reads it (pinned copy, stream-ordered) one round behind, to size the next grids.
struct RoundCtl {
    int32_t H, remaining, done, drain, fault, policy, rounds, max_h;
    int64_t ops_base, stmt_base;
    int64_t mismatches, planned, decisions, attempted, committed, rollbacks, ops;
    int64_t fill_cycles, fill_load, fill_update, fill_rescan, block_loads, rescans1, rescans2, rescans3;
    int32_t last_h, last_planned, last_done, last_mismatch;  // the round just closed (KAI_BATCH_TRACE)
    int64_t last_decisions, last_steps, last_committed;
    int32_t mm_flag, mm_out, mm_nt, mm_cls;  // the mispredicted job of that round: predicted / actual outcome, tasks, its one scan class (-1: several)
};

// k_plan_scan cut into segments (kai_plan_segments.hpp): a workgroup of KPS_T threads takes KPS_SEG consecutive positions of a node's stream, KPS_E per thread
#if defined(__HIPCC__)
enum { KPS_T = 256, KPS_E = 4 };
#else
enum { KPS_T = 64, KPS_E = 2 };  // the emulator: short segments, so that the tests' small clusters cut their streams into several
#endif
enum { KPS_SEG = KPS_T * KPS_E };
enum { KB_ROUND_SLOTS = 4 };  // pinned copies of RoundCtl the host reads behind the stream (round r in slot r % 4; the host is at most two rounds behind)
// How far the next plan looks.  A round without a surprise: back to the full depth at once (a leaf rarely holds more than 256 queued jobs, so that plan covers the whole queue);
// a plan mostly thrown away: a QUARTER as far (the next surprise is usually close: the short plans in between are the cheaper the shorter they are).  KAI_BATCH_POLICY selects the
// other rules of round 5's A/B run (0: x2 up, /2 down; 1: x4 up; 2: full depth at once, /2 down; 3: x4 up, /4 down; 4, the default: full depth at once, /4 down).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int kb_round_policy(int policy, int H, bool mismatch, int n_done, int planned) {
    if (!mismatch) { const int up = policy == 0 ? H * 2 : (policy == 1 || policy == 3) ? H * 4 : (H * 2 > 256 ? H * 2 : 256); return up < (1 << 20) ? up : (1 << 20); }
    if ((int64_t)n_done * 4 < planned) { const int dn = (policy == 3 || policy == 4) ? H / 4 : H / 2; return dn > 8 ? dn : 8; }  // most of the plan was thrown away
    return H;
}

// per-node record of the fill kernel: everything the class key reads of one node, 64 bytes, node-major (one wave loads a 64-node block as 4 KB).
// The static predicates (class_fit table, DRA / MIG rules, readiness, worker labels: plugins/predicates/predicates.go:173-262 minus the resource
// and pod-count checks) are folded into okmask when the records are built.
struct NodeRec {
    double idle[4];
    double cnt_gpu, cnt_cpu;  // divisors of the spread score (plugins/nodeplacement/spread.go:16-36)
    uint64_t okmask;          // bit k: every static predicate of scan class k passes on this node
    uint32_t cpu_node, pad;   // NodeInfo.IsCPUOnlyNode (node_info.go:697-702)
};
// one entry of the class index: arg-max key of a block / super-block / the cluster and the node that holds it (key 0 = no fitting node)
struct IdxE { uint64_t key; int32_t node, pad; };

struct BatchCtx {
    int32_t enabled, n_h, pool_e, pool_k;  // n_h: heights incl. the virtual root's
    // static per session
    KAI_GP(int32_t) q_height;    // [Q+1] leaf = 0; index Q = virtual root
    KAI_GP(int32_t) h_off;       // [n_h+1]
    KAI_GP(int32_t) h_nodes;     // [Q+1] queue nodes by ascending height
    KAI_GP(int32_t) q_srank;     // [Q] static rank among siblings: allocatable-share dominance, creation time (queue_order.go:214-240)
    KAI_GP(int32_t) q_islot;     // [Q+1] inner queue nodes numbered in h_nodes order behind the leaves (slot s = h_nodes[h_off[1] + s]), -1 = leaf / virtual root: k_apply_jobs sums their shares per workgroup in LDS
    // per action
    KAI_GP(uint64_t) j_clsmask;  // [J] scan classes of the job's chunk
    KAI_GP(int32_t) j_ucls;      // [J] the one scan class of the job's chunk, -1 = mixed
    KAI_GP(int32_t) cur_sp;      // [Q] stale-path job of an inner node (-1 = its true best job: nothing popped from it yet)
    KAI_GP(int32_t) qual;        // [4] 0: irregular jobs, 1: sibling sets without a strict static order, 2: queued jobs, 3: pad
    // per round
    KAI_GP(int32_t) q_cnt, q_ebase, q_kbase, q_valid, q_nk, q_sent, q_taken;  // [Q+1]
    KAI_GP(uint8_t) q_complete;  // [Q+1]
    KAI_GP(int32_t) plan_tot;    // [2] element / key slots the round's regions take (k_plan_setup): the per-slot kernels leave beyond them (the host sizes their grids by an upper bound)
    KAI_GP(PlanKey) pk;          // [pool_k] running maximum of the node's keys
    KAI_GP(int32_t) sp;          // [pool_k] stale-path job after t selections
    KAI_GP(int32_t) k_owner;     // [pool_k] queue node that wrote the key slot
    KAI_GP(int32_t) el_leaf, el_ck;  // [pool_e] leaf element of a merged element; key slot of its child stream AFTER the extraction
    KAI_GP(uint8_t) el_next;     // [pool_e] the child still has a key after this extraction
    KAI_GP(int32_t) e_job, e_grank;  // [pool_e] leaf regions: job, rank in the global order (INT_MAX = not in its valid prefix)
    KAI_GP(uint8_t) e_flag;      // [pool_e] leaf regions: BF_*
    // per position of an inner node's merged stream, gathered chip-wide before the node's scan (k_plan_gather): what the scan would otherwise fetch through three dependent loads per element
    KAI_GP(double) d_res;        // [pool_e][3] the element's job: resources of its tasks-to-allocate chunk
    KAI_GP(uint8_t) d_meta;      // [pool_e] its flag (BF_*) | non-preemptible << 2
    KAI_GP(double) d_spres;      // [pool_k][3] resources of the stale-path job the node's key is read through before pop t
    KAI_GP(int32_t) d_spj;       // [pool_k] that job
    // k_plan_scan cut into segments (kai_plan_segments.hpp): per segment slot the sums of its jobs' resources and its last running-maximum key; per node the first job its gate turns
    // away; per key position the shares before it
    KAI_GP(double) sg_tot;       // [n_sg][6]
    KAI_GP(PlanKey) sg_key;      // [n_sg]
    KAI_GP(int32_t) sg_kvalid;   // [n_sg]
    KAI_GP(int32_t) sg_fb;       // [Q+1]
    KAI_GP(double) d_ab;         // [pool_k][3]
    // global order + task stream
    KAI_GP(int32_t) g_stmt;      // [J+1] committed jobs before this one in the round (its Statement number minus the round's base)
    KAI_GP(int32_t) g_job, g_opoff;  // [J+1] planned global order: job, offset of its operations among the round's committed ones
    KAI_GP(int32_t) g_first, g_nt, g_ucls;   // [J+1] the job's pod range start, tasks in its chunk, its one scan class or -1 (what the fill kernel needs, coalesced)
    KAI_GP(uint8_t) g_flag, g_out;           // [J] predicted / actual outcome
    KAI_GP(int32_t) t_cls, t_node;           // [P] (a job's pod range) scan class of the i-th task of its chunk; node the fill kernel gave it
    KAI_GP(NodeRec) nrec;        // [NB*64]
    // bucket fill (kai_fill_buckets.hpp): HBM home of the sets "nodes with g free devices" (g = 1 .. 16), the classes' static-predicate bitmaps, the build's verdict
    KAI_GP(uint64_t) bk_words;   // [16][NB]
    KAI_GP(uint64_t) bk_ok;      // [C][NB]
    KAI_GP(int32_t) bk_meta;     // BucketMeta
    KAI_GP(FillStatus) fs;       // [1]
    KAI_GP(uint64_t) dead_mask;  // [1]
    KAI_GP(RoundCtl) ctl;        // [1] the round loop's state (dev_loop: the kernels take h_leaf / the output bases from it and leave when it says done)
    int32_t dev_loop, n_inner;   // n_inner: inner queue nodes (q_islot)
    KAI_GP(int32_t) cls_cap;     // [64] tasks of scan class k the cluster still holds at the round's start (k_class_capacity): a gang of one class that asks for more is predicted BF_DEAD
    // node-axis sharding over the GPUs of one node: this rank owns the nodes [n_lo, n_hi); everything else is replicated.
    // Per exchange every rank offers, per scan class, its K best nodes (records) and the key it holds back (its K+1st: the floor); the
    // all-gathered offers form a small VIRTUAL cluster on which every rank runs the same fill until a class's best candidate no longer
    // beats the best floor — only then can a node nobody offered matter, and the ranks exchange again.
    int32_t world, rank, n_lo, n_hi, shard_k, shard_mmax, vcap, cap_on;  // cap_on: cls_cap is summed every round (else it holds INT_MAX)
    int64_t msg_bytes;
    KAI_GP(uint64_t) sh_keys;    // [C][N] class keys of the own nodes (selection scratch)
    KAI_GP(uint32_t) cand_bits;  // [ceil(N/32)] union of the offered nodes
    KAI_GP(unsigned char) send, recv;  // exchange buffers: ShardHdr, node ids [mmax], records [mmax]; recv = world messages
    KAI_GP(NodeRec) vrec; KAI_GP(int32_t) vmap;  // [vcap] virtual cluster: records and global node ids, ascending
    KAI_GP(uint64_t) v1k; KAI_GP(int32_t) v1n;   // [C][vcap/64] block level of the virtual cluster's class index
    KAI_GP(IdxE) floors;         // [C] best held-back (key, node) over the ranks
    KAI_GP(int32_t) vstate;      // [4] nodes of the virtual cluster (written by k_shard_vbuild, read by the kernels that follow: no host round trip)
    KAI_GP(NodeRec) nrec_home;   // the rank's own records (nrec points at vrec while the virtual fill runs)
};
struct ShardHdr { int32_t count, pad; IdxE floor[64]; };

struct RoundParams { int32_t h_leaf, height, mode, n_slots, start, ops0, stmt0, pad2; };  // mode (fill): 0 = run the planned order from job `start`, 1 = index levels and
                                                                                        // dead-class mask only, 2 = like 0 on the virtual cluster of a node-sharded group (stops at a floor)

}  // namespace kai
