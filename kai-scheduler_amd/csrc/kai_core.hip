// kai_core.hip — C ABI of libkai_core (include/kai_core.h) on top of the gfx950 kernels.
//
// Host code here only moves the snapshot into HBM, derives index structures that are pure re-orderings / groupings
// of the input (kai_host_prep.hpp: nodes in name-rank order, CSR children, per-queue job lists, each job's pods in
// TaskOrderFn order, scan classes), launches the kernels and copies results back.  There is NO CPU implementation of
// the path in this library: without a HIP device every entry point fails with KAI_ERR_NO_DEVICE.
#define KAI_SHARED_GPUS 1  // fractions of one device (ABI v4): group tables per node, gpusharingorder score, FittingGPUs (kai_engine.hpp)
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kai_host_prep.hpp"
#include "kai_kernels.hpp"
#include "kai_batch_kernels.hpp"
#include "kai_batch_driver.hpp"
#include "kai_victim_shard.hpp"
#include <thread>

using namespace kai;

struct kai_core {
    kai_config cfg{};
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err = "ok";
    bool open = false;
    KaiCtx ctx{};
    KaiCtx* d_ctx = nullptr;  // HBM copy of ctx for the persistent kernel
    int fast_ok0 = 0;  // the snapshot's verdict on the staged job path (restored by kai_session_reset)
    bool solver_ready = false;  // scratch of the victim search allocated (first reclaim / preempt / consolidation of the session)
    std::vector<void*> bufs; std::vector<size_t> buf_bytes;  // session HBM: a few large slabs, sub-allocated (one contiguous range ⇒ few TLB entries for the latency-bound engine)
    std::vector<std::pair<void*, size_t>> spare;  // slabs of closed sessions, reused by the next open
    bool index_stale = false;  // the last action ran on the bucket fill, which keeps the sets current but not the class index (sum1_key / sum1_node): rebuilt before the next action reads it
    void* rep_buf = nullptr; size_t rep_buf_bytes = 0;  // the victim actions' replica memory: its own allocation (never part of the slab bookkeeping), kept between sessions while it is large enough
    void* pin_buf = nullptr; size_t pin_bytes = 0;  // pinned host staging for the operations handed to the caller (grows, lives with the handle)
    char* slab = nullptr; size_t slab_left = 0;
    size_t post_open_max = 0;  // largest slab a request made after kai_session_open needed in this session (trim rule of the next open)
    // device-only helpers
    double* d_jsum = nullptr; int32_t* d_slot_queue = nullptr;
    int32_t *d_lvl_off = nullptr, *d_lvl_parents = nullptr; int n_levels = 0; std::vector<int32_t> h_lvl_off;
    int32_t *d_h_off = nullptr, *d_h_nodes = nullptr; int n_heights = 0;  // queues by height (leaf = 0), for the usage roll-up
    double *d_weight = nullptr, *d_rem_amt = nullptr; uint8_t* d_rem_has = nullptr;
    int32_t* d_best_out = nullptr;
    int32_t *d_status0 = nullptr, *d_node0 = nullptr; QShare* d_shares0 = nullptr;  // HBM-resident initial state for kai_session_reset
    std::vector<int32_t> perm;  // engine node index (= name rank) → caller's node index
    kai_action_stats stats{};
    HostPrep::BatchShape shape;  // batch path of the allocate action (kai_batch.hpp)
    int world = 1, rank = 0, shard_k = 0; kai_allgather_fn ag_fn = nullptr; void* ag_user = nullptr;  // node-axis sharding over the GPUs of one node (kai_shard_attach)
    void* rccl_comm = nullptr;  // the library's own communicator (kai_shard_attach_rccl): the exchange is an ncclAllGather on `stream`, no host round trip
    bool shared = false; int32_t* d_group0 = nullptr; int32_t next_group0 = 0; int32_t *d_np_off = nullptr, *d_np_pods = nullptr;  // shared GPUs: initial groups, each node's active pods in UID order
    hipEvent_t bev[4] = {nullptr, nullptr, nullptr, nullptr};
    // rounds without the host (kai_batch_driver.hpp): per slot the round's phase events (plan start, fill start, fill end, apply end), the event behind its RoundCtl copy, the pinned copy
    hipEvent_t rev[KB_ROUND_SLOTS][5] = {}; unsigned char* rpin = nullptr; bool rev_ready = false;
    size_t fill_dyn_set[3] = {0, 0, 0};  // dynamic-LDS ceiling already set for k_fill_buckets / k_fill_counts / k_fill_levels (hipFuncSetAttribute is per process and device: once, not once per action)
    double batch_plan_ms = 0, batch_fill_ms = 0, batch_apply_ms = 0;
    // victim actions on several workgroups (kai_engine_solver.inc solve_partial_multi): every array a KaiCtx field points to, so that each workgroup gets a replica
    struct AllocRec { size_t field_off; char* base; size_t bytes; };
    std::vector<AllocRec> allocs; char* sv_base = nullptr; size_t sv_bytes = 0; char* xr_base = nullptr; size_t xr_bytes = 0;
    int mw_world = 0; char* rep_mem = nullptr; size_t rep_stride = 0; KaiCtx* d_ctxs = nullptr; MultiCtx* d_mw = nullptr; void* d_segs = nullptr; int n_segs = 0;
    ScanGrid* d_sg = nullptr;  // the scan grid's table (allocate action on the sequential engine, kai_kernels.hpp)
    // victim actions of a node-sharded group: the waves' outcomes exchanged over the ranks (kai_victim_shard.hpp) — the caller's all-gather on HOST memory (kai_shard_attach_host)
    // or the library's own RCCL communicator on xstream; the mailbox the running kernel rings, pinned staging for the copies in and out of MultiCtx
    kai_allgather_fn xag_fn = nullptr; void* xag_user = nullptr;
    XMail* mail = nullptr; XMail* d_mail = nullptr; hipStream_t xstream = nullptr;
    unsigned char* xpin = nullptr; size_t xpin_bytes = 0; unsigned char *xd_send = nullptr, *xd_recv = nullptr; size_t xd_bytes = 0;
    XShardHost xs;
    std::vector<MultiCtx> mw_host;  // staging of the victim actions' shared block (one per handle: handles of several ranks may run on threads of one process)
    // kai_session_open's host preparation, kept with the handle: a scheduler opens a session per cycle, and arrays that keep their memory are not mapped and page-faulted
    // in again every cycle (config 5: ~150 MB of host memory stays with the handle; build() rewrites every element it hands out)
    HostPrep prep; SharedPods sp;
    // pinned host staging for the snapshot's own arrays (kai_session_open, UploadStage): grows, lives with the handle
    void* up_pin = nullptr; size_t up_pin_bytes = 0;
};

#define HIP_TRY(core, expr)                                                                                        \
    do {                                                                                                           \
        hipError_t _e = (expr);                                                                                    \
        if (_e != hipSuccess) {                                                                                    \
            (core)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                                       \
            return KAI_ERR_HIP;                                                                                    \
        }                                                                                                          \
    } while (0)

namespace {

template <class T>
int dalloc(kai_core* core, T** out, size_t n) {
    size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    if (bytes > core->slab_left) {
        size_t want = std::max<size_t>(bytes, (size_t)256 << 20);  // 256 MiB slabs (2 MiB-aligned by the driver)
        if (core->open && want > core->post_open_max) core->post_open_max = want;  // (what the end-of-open trim of the NEXT session keeps a slab for)
        void* p = nullptr;
        // a slab of the previous session first: a scheduler opens a session per cycle (scheduler.go:112-138), and hipFree + hipMalloc of a few 256 MiB
        // slabs per cycle would be milliseconds of every one of them
        // (the smallest one that is large enough; what an open leaves unused is released at its end, trim_spare)
        { size_t best = core->spare.size();
          for (size_t i = 0; i < core->spare.size(); i++) if (core->spare[i].second >= want && (best == core->spare.size() || core->spare[i].second < core->spare[best].second)) best = i;
          if (best != core->spare.size()) { p = core->spare[best].first; want = core->spare[best].second; core->spare.erase(core->spare.begin() + (long)best); } }
        if (!p) HIP_TRY(core, hipMalloc(&p, want));
        core->bufs.push_back(p); core->buf_bytes.push_back(want);
        core->slab = static_cast<char*>(p); core->slab_left = want;
    }
    *out = reinterpret_cast<T*>(core->slab);
    core->slab += bytes; core->slab_left -= bytes;
    return KAI_OK;
}
template <class T>
int dupload(kai_core* core, const T** out, const T* host, size_t n) {
    T* p = nullptr;
    int rc = dalloc(core, &p, n);
    if (rc) return rc;
    if (n) HIP_TRY(core, hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, core->stream));
    else HIP_TRY(core, hipMemsetAsync(p, 0, sizeof(T), core->stream));  // (an empty array is one element nobody reads: zeros rather than what the slab held before)
    *out = p;
    return KAI_OK;
}
template <class T>
int dzero(kai_core* core, T** out, size_t n) {
    int rc = dalloc(core, out, n);
    if (rc) return rc;
    HIP_TRY(core, hipMemsetAsync(*out, 0, std::max<size_t>(n, 1) * sizeof(T), core->stream));
    return KAI_OK;
}
// KaiCtx fields are address-space-qualified pointers in the device pass (kai_engine.hpp KAI_GP): assign them through casts
#define KAI_VP(x) ((void*)(x))
template <class F>
int dalloc_f(kai_core* core, F& field, size_t n) {
    char* p = nullptr;
    int rc = dalloc(core, &p, std::max<size_t>(n, 1) * sizeof(*field));
    if (rc) return rc;
    field = (F)p;
    { const char* fp = reinterpret_cast<const char*>(&field); const char* c0 = reinterpret_cast<const char*>(&core->ctx);  // a field of the session context: its array is part of every replica
      if (fp >= c0 && fp + sizeof(void*) <= c0 + sizeof(KaiCtx)) core->allocs.push_back({(size_t)(fp - c0), p, std::max<size_t>(n, 1) * sizeof(*field)}); }
    return KAI_OK;
}
template <class F>
int dzero_f(kai_core* core, F& field, size_t n) {
    int rc = dalloc_f(core, field, n);
    if (rc) return rc;
    HIP_TRY(core, hipMemsetAsync(KAI_VP(field), 0, std::max<size_t>(n, 1) * sizeof(*field), core->stream));
    return KAI_OK;
}
template <class F, class T>
int dupload_f(kai_core* core, F& field, const T* host, size_t n) {
    static_assert(sizeof(*field) == sizeof(T), "element size");
    int rc = dalloc_f(core, field, n);
    if (rc) return rc;
    if (n) HIP_TRY(core, hipMemcpyAsync(KAI_VP(field), host, n * sizeof(T), hipMemcpyHostToDevice, core->stream));
    else HIP_TRY(core, hipMemsetAsync(KAI_VP(field), 0, sizeof(T), core->stream));  // (an empty array is one element nobody reads: zeros rather than what the slab held before)
    return KAI_OK;
}
// The snapshot's own arrays (nothing the host preparation derives) go up FIRST and from pinned memory: add() allocates the device array, flush() copies the host arrays into the
// handle's pinned staging buffer on the host's cores and enqueues one DMA per array — which then runs while kai_session_open's host preparation (milliseconds of loops over the same
// pods and jobs) keeps the cores busy.  hipMemcpyAsync from the caller's pageable arrays, as every other upload of the open is made, stages through the runtime's own bounce buffers on
// the calling thread and returns when the data has left the host: ~27 GB/s and nothing else happens meanwhile.  Below 4 MB (nearly every session of the test suites) and with
// KAI_OPEN_NO_STAGING the arrays are sent that way, in the same order to the same places.
struct UploadStage {
    kai_core* core; struct Seg { void* dst; const void* src; size_t bytes, off; }; std::vector<Seg> segs; size_t total = 0;
    template <class F, class T> int add(F& field, const T* host, size_t n) {
        static_assert(sizeof(*field) == sizeof(T), "element size");
        int rc = dalloc_f(core, field, n); if (rc) return rc;
        if (n) { segs.push_back({KAI_VP(field), host, n * sizeof(T), total}); total += (n * sizeof(T) + 255) & ~(size_t)255; }
        else HIP_TRY(core, hipMemsetAsync(KAI_VP(field), 0, sizeof(T), core->stream));
        return KAI_OK;
    }
    int flush() {
        bool pinned = total >= ((size_t)4 << 20) && !std::getenv("KAI_OPEN_NO_STAGING");
        if (pinned && total > core->up_pin_bytes) {
            if (core->up_pin) { (void)hipHostFree(core->up_pin); core->up_pin = nullptr; core->up_pin_bytes = 0; }
            const size_t want = total + total / 4;
            if (hipHostMalloc(&core->up_pin, want, hipHostMallocDefault) == hipSuccess) core->up_pin_bytes = want; else { core->up_pin = nullptr; pinned = false; (void)hipGetLastError(); }  // (no pinned memory to be had: the plain way)
        }
        if (pinned) {
            char* pin = static_cast<char*>(core->up_pin);
            parallel_chunks(total, [&](int, size_t a, size_t b) {  // byte range [a, b) of the staging buffer: the pieces of the arrays that fall into it
                for (const Seg& g : segs) { const size_t lo = std::max(a, g.off), hi = std::min(b, g.off + g.bytes); if (lo < hi) std::memcpy(pin + lo, static_cast<const char*>(g.src) + (lo - g.off), hi - lo); }
            }, (size_t)1 << 20);
            for (const Seg& g : segs) HIP_TRY(core, hipMemcpyAsync(g.dst, pin + g.off, g.bytes, hipMemcpyHostToDevice, core->stream));
        } else {
            for (const Seg& g : segs) HIP_TRY(core, hipMemcpyAsync(g.dst, g.src, g.bytes, hipMemcpyHostToDevice, core->stream));
        }
        segs.clear(); total = 0;
        return KAI_OK;
    }
};
// RCCL, resolved at run time (the library loads without it; only kai_shard_attach_rccl needs it).  Prototypes as in rccl/rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE,
// ncclUint8 = 1, ncclSuccess = 0.
struct RcclId { char internal[128]; };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi* rccl_api() {
    static RcclApi a; static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (a.lib) break; }
        if (a.lib) {
            a.GetUniqueId = reinterpret_cast<int (*)(RcclId*)>(dlsym(a.lib, "ncclGetUniqueId"));
            a.CommInitRank = reinterpret_cast<int (*)(void**, int, RcclId, int)>(dlsym(a.lib, "ncclCommInitRank"));
            a.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(a.lib, "ncclAllGather"));
            a.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(a.lib, "ncclCommDestroy"));
            a.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(a.lib, "ncclGetErrorString"));
        }
    }
    return (a.lib && a.GetUniqueId && a.CommInitRank && a.AllGather && a.CommDestroy) ? &a : nullptr;
}
int rccl_fail(kai_core* core, const char* what, int rc) {
    RcclApi* a = rccl_api();
    core->err = std::string(what) + ": " + ((a && a->GetErrorString) ? a->GetErrorString(rc) : "RCCL error");
    return KAI_ERR_COMM;
}

void free_session(kai_core* core, bool release = false) {
    // the slabs stay with the handle for the next session (dalloc takes them back); kai_core_destroy releases them
    if (core->bufs.size() != core->buf_bytes.size()) { for (void* p : core->bufs) (void)hipFree(p); core->bufs.clear(); core->buf_bytes.clear(); }  // (every slab carries its size: dalloc is the only writer of both; anything else is a bug — release rather than recycle under a wrong size)
    for (size_t i = 0; i < core->bufs.size(); i++) core->spare.push_back({core->bufs[i], core->buf_bytes[i]});
    core->bufs.clear(); core->buf_bytes.clear(); core->slab = nullptr; core->slab_left = 0;
    if (release) { for (auto& sp : core->spare) (void)hipFree(sp.first); core->spare.clear(); if (core->rep_buf) (void)hipFree(core->rep_buf); core->rep_buf = nullptr; core->rep_buf_bytes = 0; }
    core->allocs.clear(); core->sv_base = nullptr; core->xr_base = nullptr; core->mw_world = 0; core->rep_mem = nullptr; core->d_ctxs = nullptr; core->d_mw = nullptr; core->d_segs = nullptr; core->d_sg = nullptr;
    core->open = false;
}
int fail(kai_core* core, int code, const char* msg) { core->err = msg; return code; }

// session-open kernels over the HBM-resident snapshot (used by kai_session_open and kai_session_reset)
int launch_open_kernels(kai_core* core) {
    KaiCtx& c = core->ctx;
    const int N = c.N, P = c.P, J = c.J, Q = c.Q;
    const int TB = 256;
    if (core->shared) {  // shared GPUs: group tables reset, then every node adds its pods in UID order (one lane per node)
        HIP_TRY(core, hipMemsetAsync(KAI_VP(c.ng_id), 0xFF, sizeof(int32_t) * (size_t)std::max(N, 1) * KAI_GMAX, core->stream));
        if (P) hipLaunchKernelGGL(k_pod_accounting_reset, dim3((P + TB - 1) / TB), dim3(TB), 0, core->stream, c);
        if (N) hipLaunchKernelGGL(k_node_accounting_shared, dim3((N + TB - 1) / TB), dim3(TB), 0, core->stream, c, (const int32_t*)core->d_np_off, (const int32_t*)core->d_np_pods);
    } else if (P) hipLaunchKernelGGL(k_node_accounting, dim3((P + TB - 1) / TB), dim3(TB), 0, core->stream, c);
    if (core->cfg.plugins & KAI_PLUGIN_PROPORTION) {
        if (N) hipLaunchKernelGGL(k_total_nodes, dim3(std::min(1024, (N + TB - 1) / TB)), dim3(TB), 0, core->stream, c);
        if (P) hipLaunchKernelGGL(k_total_foreign, dim3((P + TB - 1) / TB), dim3(TB), 0, core->stream, c);
    }
    if (J) hipLaunchKernelGGL(k_job_usage, dim3((J + TB - 1) / TB), dim3(TB), 0, core->stream, c, core->d_jsum);
    if (core->cfg.plugins & KAI_PLUGIN_PROPORTION) {
        if (Q) hipLaunchKernelGGL(k_leaf_usage, dim3((Q + 3) / 4), dim3(TB), 0, core->stream, c, core->d_jsum);
        if (Q) hipLaunchKernelGGL(k_tree_usage, dim3(1), dim3(1024), 0, core->stream, c, (const int32_t*)core->d_h_off, (const int32_t*)core->d_h_nodes, core->n_heights);
        for (int l = 0; Q && l < core->n_levels; l++) {  // a level's totals are the fair shares of the level above: one launch per level
            const int first = core->h_lvl_off[l], count = core->h_lvl_off[l + 1] - first;
            if (count > 0) hipLaunchKernelGGL(k_fair_share_level, dim3((count * 3 + FS_WAVES - 1) / FS_WAVES), dim3(FS_WAVES * 64), 0, core->stream, c, (const int32_t*)core->d_lvl_parents, first, count,
                                              core->d_weight, core->d_rem_amt, core->d_rem_has);
        }
    }
    if (Q) hipLaunchKernelGGL(k_qnode_static, dim3((Q + TB - 1) / TB), dim3(TB), 0, core->stream, c);
    if (c.use_index && c.NB) hipLaunchKernelGGL(k_index_build, dim3((c.NB + 3) / 4), dim3(TB), 0, core->stream, c);
    HIP_TRY(core, hipGetLastError());
    return KAI_OK;
}
// the batch path's kernels on the session's stream (kai_batch_driver.hpp's Launcher)
struct DevLauncher {
    kai_core* core; int rc = 0; unsigned fill_attr_mask = 0; size_t fill_attr_dyn = 0;
    hipEvent_t* pev = nullptr;  // the phase events the next round records into: the session's four (loop on the host) or a slot's (rounds without the host)
    hipEvent_t ev(int i) { return pev ? pev[i] : core->bev[i]; }
    void static_rank(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_batch_static_rank, dim3(g), dim3(b), 0, core->stream, c); }
    void static_check(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_batch_static_check, dim3(g), dim3(b), 0, core->stream, c); }
    void qualify(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_batch_qualify, dim3(g), dim3(b), 0, core->stream, c); }
    void nrec(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_batch_nrec, dim3(g), dim3(b), 0, core->stream, c); }
    void plan_setup(int g, int b, const KaiCtx& c, RoundParams rp) { (void)hipEventRecord(ev(0), core->stream); hipLaunchKernelGGL(k_plan_setup, dim3(g), dim3(b), 0, core->stream, c, rp); }
    void plan_leaf(int g, int b, const KaiCtx& c, RoundParams rp) { hipLaunchKernelGGL(k_plan_leaf, dim3(g), dim3(b), 0, core->stream, c, rp); }
    void plan_rank(int g, int b, const KaiCtx& c, RoundParams rp) { hipLaunchKernelGGL(k_plan_rank, dim3(g), dim3(b), 0, core->stream, c, rp); }
    void plan_gather(int g, int b, const KaiCtx& c, RoundParams rp) { hipLaunchKernelGGL(k_plan_gather, dim3(g), dim3(b), 0, core->stream, c, rp); }
    void plan_scan(int g, int b, const KaiCtx& c, RoundParams rp) { hipLaunchKernelGGL(k_plan_scan, dim3(g), dim3(b), 0, core->stream, c, rp); }
    void seg_sum(int g, int b, const KaiCtx& c, RoundParams rp, int segs) { hipLaunchKernelGGL(k_seg_sum, dim3(g), dim3(b), 0, core->stream, c, rp, segs); }
    void seg_gate(int g, int b, const KaiCtx& c, RoundParams rp, int segs) { hipLaunchKernelGGL(k_seg_gate, dim3(g), dim3(b), 0, core->stream, c, rp, segs); }
    void seg_keys(int g, int b, const KaiCtx& c, RoundParams rp, int segs) { hipLaunchKernelGGL(k_seg_keys, dim3(g), dim3(b), 0, core->stream, c, rp, segs); }
    void seg_max(int g, int b, const KaiCtx& c, RoundParams rp, int segs) { hipLaunchKernelGGL(k_seg_max, dim3(g), dim3(b), 0, core->stream, c, rp, segs); }
    void plan_emit(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_plan_emit, dim3(g), dim3(b), 0, core->stream, c); }
    void class_capacity(int g, int b, const KaiCtx& c, int buckets, int levels) { hipLaunchKernelGGL(k_class_capacity, dim3(g), dim3(b), 0, core->stream, c, buckets, levels); }
    template <int MODE, bool SPEC, bool L1L> void fill_launch(int g, int b, size_t dyn, const KaiCtx& c, RoundParams rp) {
        const unsigned vbit = 1u << ((MODE == FM_SHARDED ? 4 : 0) + (SPEC ? 2 : 0) + (L1L ? 1 : 0));  // once per variant and action: the dynamic-LDS ceiling of the kernel
        if (!(fill_attr_mask & vbit) || dyn > fill_attr_dyn) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill<MODE, SPEC, L1L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) rc = KAI_ERR_HIP; fill_attr_mask |= vbit; fill_attr_dyn = std::max(fill_attr_dyn, dyn); }
        hipLaunchKernelGGL((k_fill<MODE, SPEC, L1L>), dim3(g), dim3(b), dyn, core->stream, c, rp);
    }
    void fill(int g, int b, size_t dyn, const KaiCtx& c, RoundParams rp, int l1) {
        if (rp.mode == 0 || (rp.mode == 2 && rp.start == 0)) (void)hipEventRecord(ev(1), core->stream);  // a sharded round: from its first virtual fill …
        const bool spec = (c.plugins & KB_KEY_PLUGINS) == KB_KEY_PLUGINS && c.R == 4, sh = rp.mode == 2;
        if (sh) { if (spec) { if (l1) fill_launch<FM_SHARDED, true, true>(g, b, dyn, c, rp); else fill_launch<FM_SHARDED, true, false>(g, b, dyn, c, rp); }
                  else { if (l1) fill_launch<FM_SHARDED, false, true>(g, b, dyn, c, rp); else fill_launch<FM_SHARDED, false, false>(g, b, dyn, c, rp); } }
        else { if (spec) { if (l1) fill_launch<FM_PLAIN, true, true>(g, b, dyn, c, rp); else fill_launch<FM_PLAIN, true, false>(g, b, dyn, c, rp); }
               else { if (l1) fill_launch<FM_PLAIN, false, true>(g, b, dyn, c, rp); else fill_launch<FM_PLAIN, false, false>(g, b, dyn, c, rp); } }
        if (rp.mode != 1) (void)hipEventRecord(ev(2), core->stream);                                       // … to its last (exchanges included)
    }
    void bucket_build(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_bucket_build, dim3(g), dim3(b), 0, core->stream, c); }
    void fill_counts(int g, int b, size_t dyn, const KaiCtx& c, RoundParams rp, BucketParams bp) {
        if (dyn > core->fill_dyn_set[1]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill_counts), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) rc = KAI_ERR_HIP; else core->fill_dyn_set[1] = dyn; }
        if (rp.mode == 0) (void)hipEventRecord(ev(1), core->stream);
        hipLaunchKernelGGL(k_fill_counts, dim3(g), dim3(b), dyn, core->stream, c, rp, bp);
        if (rp.mode != 1) (void)hipEventRecord(ev(2), core->stream);
    }
    void fill_levels(int g, int b, size_t dyn, const KaiCtx& c, RoundParams rp, BucketParams bp) {
        if (dyn > core->fill_dyn_set[2]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill_levels), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) rc = KAI_ERR_HIP; else core->fill_dyn_set[2] = dyn; }
        if (rp.mode == 0) (void)hipEventRecord(ev(1), core->stream);
        hipLaunchKernelGGL(k_fill_levels, dim3(g), dim3(b), dyn, core->stream, c, rp, bp);
        if (rp.mode != 1) (void)hipEventRecord(ev(2), core->stream);
    }
    void fill_buckets(int g, int b, size_t dyn, const KaiCtx& c, RoundParams rp, BucketParams bp) {
        if (dyn > core->fill_dyn_set[0]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill_buckets), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) rc = KAI_ERR_HIP; else core->fill_dyn_set[0] = dyn; }
        if (rp.mode == 0) (void)hipEventRecord(ev(1), core->stream);
        hipLaunchKernelGGL(k_fill_buckets, dim3(g), dim3(b), dyn, core->stream, c, rp, bp);
        if (rp.mode != 1) (void)hipEventRecord(ev(2), core->stream);
    }
    void apply_jobs(int g, int b, const KaiCtx& c, int64_t ops_base, int64_t stmt_base) { hipLaunchKernelGGL(k_apply_jobs, dim3(g), dim3(b), 0, core->stream, c, (long long)ops_base, (long long)stmt_base); }
    void apply_nodes(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_apply_nodes, dim3(g), dim3(b), 0, core->stream, c); (void)hipEventRecord(ev(3), core->stream); timed = true; }
    bool timed = false;
    void index_from_recs(int g, int b, const KaiCtx& c, const NodeRec* recs, int n_recs, uint64_t* l1k, int32_t* l1n, int nb, int blk0, int blk1) { hipLaunchKernelGGL(k_index_from_recs, dim3(g), dim3(b), 0, core->stream, c, recs, n_recs, l1k, l1n, nb, blk0, blk1); }
    void shard_mask_nrec(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_shard_mask_nrec, dim3(g), dim3(b), 0, core->stream, c); }
    void shard_keys(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_shard_keys, dim3(g), dim3(b), 0, core->stream, c); }
    void shard_select(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_shard_select, dim3(g), dim3(b), 0, core->stream, c); }
    void shard_compact(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_shard_compact, dim3(g), dim3(b), 0, core->stream, c); }
    void shard_vbuild(int g, int b, const KaiCtx& c) { hipLaunchKernelGGL(k_shard_vbuild, dim3(g), dim3(b), 0, core->stream, c); }
    void shard_scatter(int g, int b, const KaiCtx& c, int total) { hipLaunchKernelGGL(k_shard_scatter, dim3(g), dim3(b), 0, core->stream, c, total); }
    // the group's all-gather is the caller's (torch.distributed over RCCL / xGMI in the Python mirror): the library hands over its device buffers
    int allgather(const void* send, void* recv, int64_t bytes) {
        if (core->rccl_comm) {  // the library's own communicator: stream-ordered with the kernels on either side, nothing to wait for on the host
            const int rc = rccl_api()->AllGather(send, recv, (size_t)bytes, /*ncclUint8*/ 1, core->rccl_comm, core->stream);
            return rc ? rccl_fail(core, "ncclAllGather", rc) : KAI_OK;
        }
        if (!core->ag_fn) { core->err = "node-sharded group without kai_shard_attach"; return KAI_ERR_COMM; }
        if (hipStreamSynchronize(core->stream) != hipSuccess) return KAI_ERR_HIP;
        if (core->ag_fn(core->ag_user, send, recv, bytes) != 0) { core->err = "the caller's all-gather failed"; return KAI_ERR_COMM; }
        return KAI_OK;
    }
    int read(void* dst, const void* src, size_t n) {
        hipError_t e1 = n ? hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, core->stream) : hipSuccess, e2 = hipStreamSynchronize(core->stream), e3 = hipGetLastError();
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { core->err = std::string("batch path: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3); return KAI_ERR_HIP; }
        if (rc) return rc;
        if (timed) {  // the round's phases, from the events recorded around them
            float a = 0, f = 0, p = 0;
            if (hipEventElapsedTime(&p, core->bev[0], core->bev[1]) == hipSuccess && hipEventElapsedTime(&f, core->bev[1], core->bev[2]) == hipSuccess && hipEventElapsedTime(&a, core->bev[2], core->bev[3]) == hipSuccess) {
                core->batch_plan_ms += p; core->batch_fill_ms += f; core->batch_apply_ms += a;
            } else (void)hipGetLastError();  // an event that was not recorded this round: not an error of the path
            timed = false;
        }
        return KAI_OK;
    }
    // ---- rounds without the host: the loop's state on the device, its copies in pinned memory
    void round_init(const KaiCtx& c, int remaining, int H0, int policy, int64_t ops_base, int64_t stmt_base) { hipLaunchKernelGGL(k_round_init, dim3(1), dim3(64), 0, core->stream, c, remaining, H0, policy, (long long)ops_base, (long long)stmt_base); }
    void round_next(const KaiCtx& c) { hipLaunchKernelGGL(k_round_next, dim3(1), dim3(64), 0, core->stream, c); }
    void round_finish(const KaiCtx& c) { hipLaunchKernelGGL(k_round_finish, dim3(1), dim3(64), 0, core->stream, c); pev = nullptr; }
    bool round_ready() {
        if (core->rev_ready) return true;
        for (auto& slot : core->rev) for (hipEvent_t& e : slot) if (!e && hipEventCreate(&e) != hipSuccess) return false;
        if (!core->rpin) { void* p = nullptr; if (hipHostMalloc(&p, (size_t)KB_ROUND_SLOTS * 256, hipHostMallocDefault) != hipSuccess) return false; core->rpin = static_cast<unsigned char*>(p); }
        static_assert(sizeof(RoundCtl) <= 256, "a pinned slot holds one RoundCtl");
        return core->rev_ready = true;
    }
    void round_begin(int slot) { if (round_ready()) pev = core->rev[slot]; }
    unsigned timed_slots = 0;
    int round_post(int slot, const void* src, size_t n) {  // the state behind round `slot`, stream-ordered into the slot's pinned copy
        if (!round_ready()) { core->err = "batch path: no events / pinned memory for the round loop"; return KAI_ERR_HIP; }
        if (hipMemcpyAsync(core->rpin + (size_t)slot * 256, src, n, hipMemcpyDeviceToHost, core->stream) != hipSuccess || hipEventRecord(core->rev[slot][4], core->stream) != hipSuccess) { core->err = "batch path: the round's state copy failed"; return KAI_ERR_HIP; }
        if (timed) { timed_slots |= 1u << slot; timed = false; }
        return KAI_OK;
    }
    int round_wait(int slot, void* dst, size_t n) {  // waits for that copy — not for the stream: the next round is already behind it
        hipError_t e1 = hipEventSynchronize(core->rev[slot][4]), e2 = hipGetLastError();
        if (e1 != hipSuccess || e2 != hipSuccess) { core->err = std::string("batch path: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2); return KAI_ERR_HIP; }
        if (rc) return rc;
        std::memcpy(dst, core->rpin + (size_t)slot * 256, n);
        if (timed_slots & (1u << slot)) {
            float a = 0, f = 0, p = 0; hipEvent_t* e = core->rev[slot];
            if (hipEventElapsedTime(&p, e[0], e[1]) == hipSuccess && hipEventElapsedTime(&f, e[1], e[2]) == hipSuccess && hipEventElapsedTime(&a, e[2], e[3]) == hipSuccess) { core->batch_plan_ms += p; core->batch_fill_ms += f; core->batch_apply_ms += a; }
            else (void)hipGetLastError();
            timed_slots &= ~(1u << slot);
        }
        return KAI_OK;
    }
    int write(void* dst, const void* src, size_t n) { return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, core->stream) == hipSuccess ? KAI_OK : KAI_ERR_HIP; }
    int zero(void* dst, size_t n) { return hipMemsetAsync(dst, 0, n, core->stream) == hipSuccess ? KAI_OK : KAI_ERR_HIP; }  // (a copy from pageable host memory is staged and waited for; a memset is a launch)
};
// ---- victim actions of a node-sharded group: the host side of a wave's exchange (kai_victim_shard.hpp XShardHost) against the device
struct DevXIo {
    kai_core* core;
    // pinned layout: [0,32) MultiCtx header | res (cap x 4) | cnt (cap x 64) | scalars
    int pull(int32_t* hdr, int32_t* res, int64_t* cnt, int b, int cap) {
        MultiCtx* M = core->d_mw; unsigned char* p = core->xpin;
        if (hipMemcpyAsync(p, M, 32, hipMemcpyDeviceToHost, core->xstream) != hipSuccess) return KAI_ERR_HIP;
        if (hipMemcpyAsync(p + 64, &M->res[b][0], (size_t)cap * 4, hipMemcpyDeviceToHost, core->xstream) != hipSuccess) return KAI_ERR_HIP;
        if (hipMemcpyAsync(p + 64 + (size_t)KAI_MW_WAVE * 4, &M->cnt[b][0][0], (size_t)cap * 8 * KAI_MW_CNT, hipMemcpyDeviceToHost, core->xstream) != hipSuccess) return KAI_ERR_HIP;
        if (hipStreamSynchronize(core->xstream) != hipSuccess) return KAI_ERR_HIP;
        std::memcpy(hdr, p, 32); std::memcpy(res, p + 64, (size_t)cap * 4); std::memcpy(cnt, p + 64 + (size_t)KAI_MW_WAVE * 4, (size_t)cap * 8 * KAI_MW_CNT);
        return 0;
    }
    int push(int b, const int32_t* res, const int64_t* cnt, int cap, int hit, int xrun, int fault) {
        MultiCtx* M = core->d_mw; unsigned char* p = core->xpin;
        std::memcpy(p + 64, res, (size_t)cap * 4); std::memcpy(p + 64 + (size_t)KAI_MW_WAVE * 4, cnt, (size_t)cap * 8 * KAI_MW_CNT);
        int32_t* sc = reinterpret_cast<int32_t*>(p + 32); sc[0] = hit; sc[1] = xrun; sc[2] = 1;
        if (hipMemcpyAsync(&M->res[b][0], p + 64, (size_t)cap * 4, hipMemcpyHostToDevice, core->xstream) != hipSuccess) return KAI_ERR_HIP;
        if (hipMemcpyAsync(&M->cnt[b][0][0], p + 64 + (size_t)KAI_MW_WAVE * 4, (size_t)cap * 8 * KAI_MW_CNT, hipMemcpyHostToDevice, core->xstream) != hipSuccess) return KAI_ERR_HIP;
        if (hipMemcpyAsync(&M->hit[b], &sc[0], 4, hipMemcpyHostToDevice, core->xstream) != hipSuccess) return KAI_ERR_HIP;
        if (hipMemcpyAsync(&M->xrun[b], &sc[1], 4, hipMemcpyHostToDevice, core->xstream) != hipSuccess) return KAI_ERR_HIP;
        if (fault && hipMemcpyAsync(&M->fault, &sc[2], 4, hipMemcpyHostToDevice, core->xstream) != hipSuccess) return KAI_ERR_HIP;
        return hipStreamSynchronize(core->xstream) == hipSuccess ? 0 : (int)KAI_ERR_HIP;
    }
    int raise_fault() { int32_t* sc = reinterpret_cast<int32_t*>(core->xpin + 32); sc[2] = 1; (void)hipMemcpyAsync(&core->d_mw->fault, &sc[2], 4, hipMemcpyHostToDevice, core->xstream); return hipStreamSynchronize(core->xstream) == hipSuccess ? 0 : (int)KAI_ERR_HIP; }
    int allgather(const void* send, void* recv, int64_t bytes) {
        if (core->xag_fn) return core->xag_fn(core->xag_user, send, recv, bytes) == 0 ? 0 : (int)KAI_ERR_COMM;
        if (!core->rccl_comm || (size_t)bytes * core->world > core->xd_bytes) return KAI_ERR_COMM;
        if (hipMemcpyAsync(core->xd_send, send, (size_t)bytes, hipMemcpyHostToDevice, core->xstream) != hipSuccess) return KAI_ERR_HIP;
        if (rccl_api()->AllGather(core->xd_send, core->xd_recv, (size_t)bytes, /*ncclUint8*/ 1, core->rccl_comm, core->xstream)) return KAI_ERR_COMM;
        if (hipMemcpyAsync(recv, core->xd_recv, (size_t)bytes * core->world, hipMemcpyDeviceToHost, core->xstream) != hipSuccess) return KAI_ERR_HIP;
        return hipStreamSynchronize(core->xstream) == hipSuccess ? 0 : (int)KAI_ERR_HIP;
    }
};
// mailbox, second stream, staging: once per handle
int xshard_setup(kai_core* core) {
    if (!core->xstream) HIP_TRY(core, hipStreamCreateWithFlags(&core->xstream, hipStreamNonBlocking));
    if (!core->mail) {
        void* m = nullptr; HIP_TRY(core, hipHostMalloc(&m, 4096, hipHostMallocCoherent | hipHostMallocMapped));
        core->mail = static_cast<XMail*>(m); std::memset(m, 0, 4096);
        void* d = nullptr; HIP_TRY(core, hipHostGetDevicePointer(&d, m, 0)); core->d_mail = static_cast<XMail*>(d);
    }
    if (!core->xpin) { const size_t want = 64 + (size_t)KAI_MW_WAVE * 4 + (size_t)KAI_MW_WAVE * 8 * KAI_MW_CNT; void* p = nullptr; HIP_TRY(core, hipHostMalloc(&p, want, hipHostMallocDefault)); core->xpin = static_cast<unsigned char*>(p); core->xpin_bytes = want; }
    if (core->rccl_comm && !core->xag_fn && !core->xd_send) {
        const size_t one = xw_msg_bytes(KAI_MW_WAVE, 1); void* a = nullptr; void* b = nullptr;
        HIP_TRY(core, hipMalloc(&a, one)); HIP_TRY(core, hipMalloc(&b, one * (size_t)core->world));
        core->xd_send = static_cast<unsigned char*>(a); core->xd_recv = static_cast<unsigned char*>(b); core->xd_bytes = one * (size_t)core->world;
    }
    return KAI_OK;
}
// while the action's kernel runs: answer the mailbox (one exchange per wave) until the stream is idle
int xshard_serve(kai_core* core) {
    DevXIo io{core}; int32_t served = 0; long idle = 0;
    for (;;) {
        const int32_t req = __atomic_load_n(&core->mail->req, __ATOMIC_ACQUIRE);
        if (req != served) {
            const int b = __atomic_load_n(&core->mail->buf, __ATOMIC_ACQUIRE) & 1;
            if (core->xs.wave(io, b)) (void)io.raise_fault();  // the engines give up at the barrier behind the answer
            served = req; __atomic_store_n(&core->mail->resp, served, __ATOMIC_RELEASE); idle = 0;
            continue;
        }
        if ((++idle & 63) == 0) {
            const hipError_t q = hipStreamQuery(core->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { core->err = std::string("victim action: ") + hipGetErrorString(q); return KAI_ERR_HIP; }
            if (idle > 4096) std::this_thread::yield();
        }
    }
    return KAI_OK;
}

// ---- victim actions on several workgroups: replicas of the session arrays, one per further workgroup
struct RepSeg { const char* src; unsigned long long dst_off, bytes; };
constexpr size_t REP_CHUNK = (size_t)256 << 10;
__global__ void __launch_bounds__(256) k_replicate(const RepSeg* segs, char* rep_mem, unsigned long long stride) {
    const RepSeg sg = segs[blockIdx.x];
    const uint4* s = reinterpret_cast<const uint4*>(sg.src); uint4* d = reinterpret_cast<uint4*>(rep_mem + (size_t)blockIdx.y * stride + sg.dst_off);
    const size_t n16 = (size_t)sg.bytes / 16;  // (every allocation is a multiple of 256 bytes)
    for (size_t i = threadIdx.x; i < n16; i += blockDim.x) d[i] = s[i];
}
// Builds (first call of a session) and refreshes (every call) the replicas for a victim action on `G` workgroups; G <= 1 or no memory: one workgroup.
int prepare_multi(kai_core* core, int G, int* g_out) {
    *g_out = 1;
    KaiCtx& c = core->ctx;
    c.mw = nullptr; c.mw_rank = 0; c.mw_world = 1;
    if (G <= 1 || core->shared || !core->sv_base) return KAI_OK;  // (with shared GPUs a rolled-back simulation is observable on its node: the simulations of a partial job are not independent)
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    if (core->mw_world != G) {  // the layout of a replica: the registered arrays, the solver's scratch, the shared-GPU residency groups
        if (core->mw_world != 0) return KAI_OK;  // (replicas of another width exist already: keep to one workgroup rather than rebuild)
        size_t total = 0; std::vector<RepSeg> segs;
        auto add = [&](const char* src, size_t bytes) { bytes = up(bytes); for (size_t o = 0; o < bytes; o += REP_CHUNK) segs.push_back({src + o, (unsigned long long)(total + o), (unsigned long long)std::min(REP_CHUNK, bytes - o)}); total += bytes; };
        for (const auto& a : core->allocs) add(a.base, a.bytes);
        add(core->sv_base, core->sv_bytes); add(core->xr_base, core->xr_bytes);
        const size_t need = total * (size_t)(G - 1);
        if (core->rep_buf_bytes < need) {  // the replicas of the previous session serve again while they are large enough
            if (core->rep_buf) (void)hipFree(core->rep_buf);
            core->rep_buf = nullptr; core->rep_buf_bytes = 0;
            void* mem = nullptr;
            if (hipMalloc(&mem, need) != hipSuccess) { (void)hipGetLastError(); if (c.mw_xworld > 1) { core->err = "victim action of a node-sharded group: no memory for the replicas (every rank must run the same number of engines)"; return KAI_ERR_HIP; } return KAI_OK; }
            core->rep_buf = mem; core->rep_buf_bytes = need;
        }
        core->rep_mem = static_cast<char*>(core->rep_buf); core->rep_stride = total;
        RepSeg* dsegs = nullptr; KaiCtx* dctx = nullptr; MultiCtx* dmw = nullptr;
        int rc = dalloc(core, &dsegs, segs.size()); if (rc) return rc;
        rc = dalloc(core, &dctx, (size_t)G); if (rc) return rc;
        rc = dalloc(core, &dmw, (size_t)1); if (rc) return rc;
        HIP_TRY(core, hipMemcpyAsync(dsegs, segs.data(), segs.size() * sizeof(RepSeg), hipMemcpyHostToDevice, core->stream));
        HIP_TRY(core, hipStreamSynchronize(core->stream));  // segs dies with this scope
        core->d_segs = dsegs; core->n_segs = (int)segs.size(); core->d_ctxs = dctx; core->d_mw = dmw; core->mw_world = G;
    }
    // this action's contexts: replica w = the session's context with every array pointer moved into replica w's memory
    std::vector<KaiCtx> ctxs((size_t)G, c);
    for (int w = 0; w < G; w++) {
        KaiCtx& cw = ctxs[w];
        cw.mw = core->d_mw; cw.mw_rank = w; cw.mw_world = G;
        if (w == 0) continue;
        char* rb = core->rep_mem + (size_t)(w - 1) * core->rep_stride; size_t off = 0;
        for (const auto& a : core->allocs) { *reinterpret_cast<char**>(reinterpret_cast<char*>(&cw) + a.field_off) = rb + off; off += up(a.bytes); }
        solver_scratch_bind(cw.sv, rb + off, c.N, c.P, c.S, c.J, c.Q, c.W, c.D + c.T, c.TL, c.G); off += up(core->sv_bytes);
        cw.sv.xr_group = reinterpret_cast<int32_t*>(rb + off); off += up(core->xr_bytes);
        cw.bt.enabled = 0;  // (the batch path's pools are not replicated; a victim action never reads them)
    }
    HIP_TRY(core, hipMemcpyAsync(core->d_ctxs, ctxs.data(), (size_t)G * sizeof(KaiCtx), hipMemcpyHostToDevice, core->stream));
    if (core->mw_host.empty()) core->mw_host.resize(1);
    MultiCtx& m0 = core->mw_host[0]; std::memset(&m0, 0, sizeof m0); m0.world = G; m0.hit[0] = m0.hit[1] = 0x7fffffff;
    HIP_TRY(core, hipMemcpyAsync(core->d_mw, &m0, sizeof(MultiCtx), hipMemcpyHostToDevice, core->stream));
    hipLaunchKernelGGL(k_replicate, dim3(core->n_segs, G - 1), dim3(256), 0, core->stream, (const RepSeg*)core->d_segs, core->rep_mem, (unsigned long long)core->rep_stride);
    HIP_TRY(core, hipGetLastError());
    HIP_TRY(core, hipStreamSynchronize(core->stream));  // ctxs / m0 die with this scope
    *g_out = G;
    return KAI_OK;
}
}  // namespace

extern "C" {

const char* kai_version(void) { return "kai_core abi 5 gfx950 (HIP; batch plan/fill/apply path, node-axis sharding, device-resident sequential engine, class index)"; }

const char* kai_last_error(kai_core* core) { return core ? core->err.c_str() : "null handle"; }

int kai_core_create(const kai_config* cfg, int n_gpus, const int* gpu_ids, kai_core** out) {
    if (!cfg || !out || cfg->abi_version != KAI_ABI_VERSION || n_gpus < 1) return KAI_ERR_INVALID_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return KAI_ERR_NO_DEVICE;
    int dev = gpu_ids ? gpu_ids[0] : 0;
    if (dev < 0 || dev >= count) return KAI_ERR_INVALID_ARG;
    kai_core* core = new kai_core();
    core->cfg = *cfg; core->device = dev; core->world = n_gpus;  // n_gpus > 1: this handle is ONE rank of a node-sharded group (kai_shard_attach names the rank)
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&core->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&core->ev0) != hipSuccess || hipEventCreate(&core->ev1) != hipSuccess || hipEventCreate(&core->bev[0]) != hipSuccess ||
        hipEventCreate(&core->bev[1]) != hipSuccess || hipEventCreate(&core->bev[2]) != hipSuccess || hipEventCreate(&core->bev[3]) != hipSuccess) {
        delete core;
        return KAI_ERR_HIP;
    }
    *out = core;
    return KAI_OK;
}

int kai_core_destroy(kai_core* core) {
    if (!core) return KAI_ERR_INVALID_ARG;
    (void)hipSetDevice(core->device);
    free_session(core, true);
    if (core->pin_buf) { (void)hipHostFree(core->pin_buf); core->pin_buf = nullptr; core->pin_bytes = 0; }
    if (core->up_pin) { (void)hipStreamSynchronize(core->stream); (void)hipHostFree(core->up_pin); core->up_pin = nullptr; core->up_pin_bytes = 0; }
    if (core->rccl_comm) { (void)hipStreamSynchronize(core->stream); if (RcclApi* a = rccl_api()) (void)a->CommDestroy(core->rccl_comm); core->rccl_comm = nullptr; }
    if (core->mail) (void)hipHostFree(core->mail);
    if (core->xpin) (void)hipHostFree(core->xpin);
    if (core->xd_send) (void)hipFree(core->xd_send);
    if (core->xd_recv) (void)hipFree(core->xd_recv);
    if (core->xstream) (void)hipStreamDestroy(core->xstream);
    if (core->ev0) (void)hipEventDestroy(core->ev0);
    if (core->ev1) (void)hipEventDestroy(core->ev1);
    for (hipEvent_t e : core->bev) if (e) (void)hipEventDestroy(e);
    for (auto& slot : core->rev) for (hipEvent_t e : slot) if (e) (void)hipEventDestroy(e);
    if (core->rpin) (void)hipHostFree(core->rpin);
    if (core->stream) (void)hipStreamDestroy(core->stream);
    delete core;
    return KAI_OK;
}

int kai_session_close(kai_core* core) {
    if (!core) return KAI_ERR_INVALID_ARG;
    (void)hipSetDevice(core->device);
    free_session(core);
    return KAI_OK;
}

static int session_open_impl(kai_core* core, const kai_snapshot_soa* s) {
    if (s->abi_version != KAI_ABI_VERSION || s->n_res < 4 || s->n_res > KAI_MAX_RES) return fail(core, KAI_ERR_INVALID_ARG, "bad abi_version / n_res");
    HIP_TRY(core, hipSetDevice(core->device));
    const bool prof_open = std::getenv("KAI_PROF") != nullptr;  // host clocks of the open: where a production cycle's per-cycle cost goes (stderr)
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_begin = tnow();
    free_session(core);
    const auto t_freed = tnow();
    const int N = s->n_nodes, P = s->n_pods, S = s->n_podsets, J = s->n_jobs, Q = s->n_queues, R = s->n_res;
    if (N < 0 || P < 0 || S < 0 || J < 0 || Q < 0) return fail(core, KAI_ERR_INVALID_ARG, "negative dimension");
    if (const char* m = HostPrep::missing_array(s)) return fail(core, KAI_ERR_INVALID_ARG, m);  // (before anything reads the snapshot's arrays; HostPrep::build reports the same)
    bool any_legacy_mig = false;
    if (s->pod_flags) {  // one pass over the pods' flags on the host's cores: the two fallback rules (the first one wins, as when checked one after the other) and whether any pod is a legacy MIG task
        const int K = chunk_count((size_t)P);
        std::vector<unsigned char> seen((size_t)K, 0);
        parallel_chunks((size_t)P, [&](int ci, size_t p0, size_t p1) {
            unsigned char f = 0;
            for (size_t p = p0; p < p1; p++) {
                const uint32_t fl = s->pod_flags[p];
                if (!(fl & (KAI_POD_CPU_FALLBACK | KAI_POD_GPU_UNMODELLED | KAI_POD_LEGACY_MIG))) continue;
                if ((fl & KAI_POD_CPU_FALLBACK) && s->pod_status[p] == KAI_POD_PENDING) f |= 1;
                if ((fl & KAI_POD_GPU_UNMODELLED) && (s->pod_status[p] & (KAI_POD_ALLOCATED | KAI_POD_PIPELINED | KAI_POD_BINDING | KAI_POD_BOUND | KAI_POD_RUNNING | KAI_POD_RELEASING))) f |= 2;
                if (fl & KAI_POD_LEGACY_MIG) f |= 4;
            }
            seen[(size_t)ci] = f;
        });
        unsigned char all = 0; for (unsigned char f : seen) all |= f;
        if (all & 1) return fail(core, KAI_ERR_UNSUPPORTED, "a pending pod is flagged KAI_POD_CPU_FALLBACK: leave its job to the host path");
        if (all & 2) return fail(core, KAI_ERR_UNSUPPORTED, "an active pod holds GPU state the device does not model (gpu-memory / several fractional devices / MIG / DRA): its node's idle GPUs would be overstated");
        any_legacy_mig = all & 4;
    }
    // shared GPUs (ABI v4): fractions of one device.  One GPU memory size for the whole cluster keeps the queue-capacity step node independent.
    SharedPods& sp = core->sp;
    try { if (!sp.build(core->cfg, s)) return fail(core, KAI_ERR_UNSUPPORTED, sp.err.c_str()); }
    catch (const std::bad_alloc&) { throw; }  // (kai_session_open reports it as what it is)
    catch (const std::exception& e) { core->err = std::string("host preparation: ") + e.what(); return KAI_ERR_INVALID_ARG; }
    const bool shared = sp.any;
    core->shared = shared;

    HIP_TRY(core, hipEventRecord(core->ev0, core->stream));
    KaiCtx& c = core->ctx;
    c = KaiCtx{};
    c.N = N; c.P = P; c.S = S; c.J = J; c.Q = Q; c.R = R; c.n_pod_classes = std::max(1, s->n_pod_classes); c.n_node_classes = std::max(1, s->n_node_classes);
    c.plugins = core->cfg.plugins; c.gpu_strategy = core->cfg.gpu_strategy; c.cpu_strategy = core->cfg.cpu_strategy;
    c.restrict_nodes = core->cfg.restrict_node_scheduling; c.k_value = core->cfg.k_value <= 0.0 ? 0.0 : core->cfg.k_value;  // proportion.go:77-84

    // ---- the snapshot's own arrays: allocated first, sent from pinned staging while the host preparation below runs (UploadStage)
    const auto t_shared = tnow();
    {
        UploadStage up{core};
        int rcu = 0;
#define UP(field, host, n) do { if (!rcu) rcu = up.add(field, host, (size_t)(n)); } while (0)
        UP(c.p_req, s->pod_req, (size_t)R * P); UP(c.p_job, s->pod_job, P); UP(c.p_podset, s->pod_podset, P);
        if (s->pod_flags) UP(c.p_flags, s->pod_flags, P); else if (!rcu) rcu = dzero_f(core, c.p_flags, (size_t)P);  // (an optional array that is absent: zeros written on the device, not 4 MB of host zeros sent over)
        if (s->pod_class) UP(c.p_class, s->pod_class, P); else if (!rcu) rcu = dzero_f(core, c.p_class, (size_t)P);
        UP(c.p_status, s->pod_status, P);
        UP(c.s_job, s->podset_job, S); UP(c.s_min, s->podset_min_available, S); UP(c.s_name_rank, s->podset_name_rank, S);
        UP(c.j_queue, s->job_queue, J); UP(c.j_prio, s->job_priority, J); UP(c.j_preempt, s->job_preemptible, J); UP(c.j_created, s->job_created_ns, J); UP(c.j_uid_rank, s->job_uid_rank, J);
        UP(c.j_first_pod, s->job_first_pod, J); UP(c.j_n_pods, s->job_n_pods, J); UP(c.j_first_ps, s->job_first_podset, J); UP(c.j_n_ps, s->job_n_podsets, J);
#undef UP
        if (!rcu) rcu = up.flush();
        if (rcu) return rcu;
    }
    // ---- index structures (pure re-orderings / groupings of the input; kai_host_prep.hpp)
    const auto t_staged = tnow();
    HostPrep& prep = core->prep;
    prep.shared_pods = sp.any ? &sp : nullptr;
    try { if (prep.build(core->cfg, s, core->err)) return KAI_ERR_INVALID_ARG; }  // (the staged copies read the handle's pinned buffer: kai_session_open drains the stream on every failure)
    catch (const std::bad_alloc&) { throw; }  // (a worker thread's included: kai_parallel.hpp carries it here)
    catch (const std::exception& e) { core->err = std::string("host preparation: ") + e.what(); return KAI_ERR_INVALID_ARG; }
    const auto t_prep = tnow();
    if (any_legacy_mig) for (int p = 0; p < P; p++)  // NodeInfo.LegacyMIGTasks (node_info.go:407-409): a node that holds a legacy MIG task takes no MIG request
        if ((s->pod_flags[p] & KAI_POD_LEGACY_MIG) && prep.pod_node[p] >= 0 && (s->pod_status[p] & (KAI_POD_ALLOCATED | KAI_POD_PIPELINED | KAI_POD_BINDING | KAI_POD_BOUND | KAI_POD_RUNNING | KAI_POD_RELEASING))) prep.node_flags[prep.pod_node[p]] |= KAI_NODE_LEGACY_MIG_I;
    core->perm = prep.perm;

    int rc;
#define TRY(x) do { rc = (x); if (rc) return rc; } while (0)
    // ---- static arrays (optional ones get neutral defaults); nodes go up in name-rank order
    uint8_t one = 1;
    TRY(dupload_f(core, c.n_alloc, prep.node_alloc.data(), (size_t)R * N));
    TRY(dupload_f(core, c.n_flags, prep.node_flags.data(), (size_t)N));
    TRY(dupload_f(core, c.n_gpu_count, prep.node_gpu_count.data(), (size_t)N));
    TRY(dupload_f(core, c.n_class, prep.node_class.data(), (size_t)N));
    const bool full_uploads = std::getenv("KAI_OPEN_FULL_UPLOADS") != nullptr;  // (diagnostic: every array sent from the host, none of the constants below written on the device)
    if (prep.any_nominated || full_uploads) TRY(dupload_f(core, c.p_nominated, prep.pod_nominated.data(), (size_t)P));
    else { TRY(dalloc_f(core, c.p_nominated, (size_t)P)); HIP_TRY(core, hipMemsetAsync(KAI_VP(c.p_nominated), 0xFF, (size_t)std::max(P, 1) * sizeof(int32_t), core->stream)); }  // no nominated node anywhere: -1 in every element
    TRY(dupload_f(core, c.p_scls, prep.pod_scls.data(), (size_t)P));
    TRY(dupload_f(core, c.q_parent, s->queue_parent, (size_t)Q));
    TRY(dupload_f(core, c.q_prio, s->queue_priority, (size_t)Q));
    TRY(dupload_f(core, c.q_created, s->queue_created_ns, (size_t)Q));
    TRY(dupload_f(core, c.q_uid_rank, s->queue_uid_rank, (size_t)Q));
    if (s->class_fit && s->n_pod_classes > 0 && s->n_node_classes > 0) TRY(dupload_f(core, c.class_fit, s->class_fit, (size_t)s->n_pod_classes * s->n_node_classes));
    else TRY(dupload_f(core, c.class_fit, &one, (size_t)1));
    TRY(dupload_f(core, c.j_pods_sorted, prep.sorted.data(), (size_t)P));
    TRY(dupload_f(core, c.q_child_off, prep.child_off.data(), (size_t)Q + 2));
    TRY(dupload_f(core, c.q_children, prep.children.data(), (size_t)std::max(Q, 1)));
    TRY(dupload_f(core, c.q_job_off, prep.job_off.data(), (size_t)Q + 1));
    TRY(dupload_f(core, c.jobs_static, prep.jobs_static.data(), (size_t)std::max(J, 1)));
    TRY(dupload_f(core, c.q_depth_order, prep.depth_order.data(), (size_t)Q));
    { const int32_t* t; TRY(dupload(core, &t, prep.slot_queue.data(), (size_t)std::max(J, 1))); core->d_slot_queue = const_cast<int32_t*>(t);
      TRY(dupload(core, &t, prep.lvl_off.data(), prep.lvl_off.size())); core->d_lvl_off = const_cast<int32_t*>(t);
      TRY(dupload(core, &t, prep.lvl_parents.data(), prep.lvl_parents.size())); core->d_lvl_parents = const_cast<int32_t*>(t); }
    core->n_levels = prep.n_levels; core->h_lvl_off = prep.lvl_off; core->n_heights = prep.n_heights;
    { const int32_t* t; TRY(dupload(core, &t, prep.h_off.data(), prep.h_off.size())); core->d_h_off = const_cast<int32_t*>(t);
      TRY(dupload(core, &t, prep.h_nodes.data(), prep.h_nodes.size())); core->d_h_nodes = const_cast<int32_t*>(t); }
    // ---- scan classes + class index
    c.C = (int)prep.classes.size(); c.NB = (N + KAI_BLOCK - 1) / KAI_BLOCK; c.NSB = (c.NB + 63) / 64;
    c.use_index = c.C > 0 ? 1 : 0; c.all_tracked = prep.all_tracked; c.fast_ok = prep.fast_ok; core->fast_ok0 = prep.fast_ok; c.exact_sums = prep.exact_sums;
    { int d = core->cfg.queue_depth[KAI_ACTION_ALLOCATE]; c.queue_depth = d > 0 ? d : 0; }
    c.action = KAI_ACTION_ALLOCATE; c.max_consolidation_preemptees = core->cfg.max_consolidation_preemptees; c.allow_consolidating_reclaim = core->cfg.allow_consolidating_reclaim;
    c.saturation_multiplier = core->cfg.reclaimer_saturation_multiplier; c.sv = SolverCtx{}; core->solver_ready = false;
    c.use_signatures = core->cfg.use_scheduling_signatures ? 1 : 0; c.j_signature = nullptr;
    if (s->job_signature) TRY(dupload_f(core, c.j_signature, s->job_signature, (size_t)J));
    c.j_last_start = nullptr; c.q_preempt_mr = nullptr; c.q_reclaim_mr = nullptr;
    if (s->job_last_start_ns) TRY(dupload_f(core, c.j_last_start, s->job_last_start_ns, (size_t)J));
    if (s->queue_preempt_min_runtime_ns) TRY(dupload_f(core, c.q_preempt_mr, s->queue_preempt_min_runtime_ns, (size_t)Q));
    if (s->queue_reclaim_min_runtime_ns) TRY(dupload_f(core, c.q_reclaim_mr, s->queue_reclaim_min_runtime_ns, (size_t)Q));
    c.now_ns = core->cfg.now_ns; c.def_preempt_mr = core->cfg.default_preempt_min_runtime_ns; c.def_reclaim_mr = core->cfg.default_reclaim_min_runtime_ns; c.reclaim_method = core->cfg.reclaim_resolve_method;
    TRY(dupload_f(core, c.cls, prep.classes.data(), prep.classes.size()));
    TRY(dzero_f(core, c.sum1_key, (size_t)std::max(c.C, 1) * std::max(c.NB, 1))); TRY(dzero_f(core, c.sum1_node, (size_t)std::max(c.C, 1) * std::max(c.NB, 1)));

    // ---- shared GPUs: per-pod portion / group, per-node GPU memory and group tables (api/node_info/gpu_sharing_node_info.go)
    {
        std::vector<int64_t> gm((size_t)std::max(N, 1), 100);
        int32_t next_new = KAI_NEW_GROUP;
        for (int n = 0; n < N; n++) gm[n] = s->node_gpu_memory ? s->node_gpu_memory[prep.perm[n]] : 100;
        c.quota_on = sp.on ? 1 : 0; c.mig_on = sp.mig ? 1 : 0;
        for (int r = 0; r < KAI_MAX_RES; r++) { c.res_mig_g[r] = sp.mig_g[r]; c.res_mig_m[r] = sp.mig_m[r]; }
        // (both ways allocate the arrays in ONE order — the session's layout in HBM does not depend on what is sent; tests/test_open_uploads.py compares the images)
        TRY(dupload_f(core, c.p_shared, sp.shared.data(), (size_t)P));
        const size_t P1 = (size_t)std::max(P, 1);
        bool por_zero = true;  // pod_gpu_portion absent, or +0.0 in every element (what a packer that always fills the array sends for a cluster without fractions)
        if (!sp.on && s->pod_gpu_portion) {
            std::vector<char> nz((size_t)chunk_count((size_t)P), 0);
            parallel_chunks((size_t)P, [&](int ci, size_t p0, size_t p1) { for (size_t p = p0; p < p1; p++) { uint64_t b; std::memcpy(&b, &s->pod_gpu_portion[p], 8); if (b) { nz[(size_t)ci] = 1; return; } } });
            for (char x : nz) if (x) por_zero = false;
        }
        // A snapshot without shared-GPU requests and without MIG rows (SharedPods: neither `any` nor `mig`): the per-pod quantities of the shared-GPU model are constants —
        // no memory of a device asked for, no MIG quota, every GPU quantity the pod's GPU request (the row of p_req that is in HBM already), no group anywhere.  They are
        // written where they are read instead of being sent: at config 5 these arrays are 73 of the open's 190 MB over PCIe.  (kai_hostsim_run checks the same statement
        // about SharedPods on every snapshot the CPU suite runs: tests/host_sim/host_sim.cpp lean_shared_pods_hold.)
        const bool lean_pods = !sp.on && por_zero && !full_uploads;
        std::vector<double> por; std::vector<int32_t> grp, minus1;
        if (!lean_pods) {
            por.assign(P1, 0.0); grp.assign(P1, -1); minus1.assign(P1, -1);
            for (int p = 0; p < P; p++) { por[p] = s->pod_gpu_portion ? s->pod_gpu_portion[p] : 0.0; grp[p] = (s->pod_gpu_group && sp.shared[p]) ? s->pod_gpu_group[p] : -1; if (grp[p] >= next_new) next_new = grp[p] + 1; }
        }
        auto ones = [&](auto& field, size_t n) -> int { int rc2 = dalloc_f(core, field, n); if (rc2) return rc2; HIP_TRY(core, hipMemsetAsync(KAI_VP(field), 0xFF, std::max<size_t>(n, 1) * sizeof(*field), core->stream)); return KAI_OK; };  // -1 in every element
        auto gpu_row = [&](auto& field) -> int {  // the pods' GPU requests: a device-to-device copy of that row of p_req
            int rc2 = dalloc_f(core, field, (size_t)P); if (rc2) return rc2;
            if (P) HIP_TRY(core, hipMemcpyAsync(KAI_VP(field), (const double*)KAI_VP(c.p_req) + (size_t)KAI_RES_GPU * P, (size_t)P * sizeof(double), hipMemcpyDeviceToDevice, core->stream));
            return KAI_OK; };
        if (lean_pods) {
            TRY(dzero_f(core, c.p_mem, (size_t)P)); TRY(dzero_f(core, c.p_gmem, (size_t)P));
            TRY(gpu_row(c.p_acc_gpu)); TRY(gpu_row(c.p_pend_gpu)); TRY(gpu_row(c.p_quota_gpu));
            TRY(dzero_f(core, c.p_mig_q, (size_t)P));
        } else {
            TRY(dupload_f(core, c.p_mem, sp.mem.data(), (size_t)P)); TRY(dupload_f(core, c.p_gmem, sp.gmem.data(), (size_t)P));
            TRY(dupload_f(core, c.p_acc_gpu, sp.acc_gpu.data(), (size_t)P)); TRY(dupload_f(core, c.p_pend_gpu, sp.pend_gpu.data(), (size_t)P));
            TRY(dupload_f(core, c.p_quota_gpu, sp.quota_gpu.data(), (size_t)P)); TRY(dupload_f(core, c.p_mig_q, sp.mig_q.data(), (size_t)P));
        }
        TRY(dupload_f(core, c.p_kind, sp.kind.data(), (size_t)P));
        if (lean_pods) { TRY(dzero_f(core, c.p_portion, (size_t)P)); TRY(ones(c.p_group, (size_t)P)); TRY(ones(c.p_on_group, (size_t)P)); }
        else { TRY(dupload_f(core, c.p_portion, por.data(), (size_t)P)); TRY(dupload_f(core, c.p_group, grp.data(), (size_t)P)); TRY(dupload_f(core, c.p_on_group, minus1.data(), (size_t)P)); }
        TRY(dupload_f(core, c.n_gpu_mem, gm.data(), (size_t)N));
        if (lean_pods) { int32_t* t = nullptr; TRY(dalloc(core, &t, P1)); HIP_TRY(core, hipMemsetAsync(t, 0xFF, P1 * sizeof(int32_t), core->stream)); core->d_group0 = t; }
        else { const int32_t* t; TRY(dupload(core, &t, grp.data(), P1)); core->d_group0 = const_cast<int32_t*>(t); }
        TRY(dalloc_f(core, c.ng_id, (size_t)N * KAI_GMAX)); TRY(dzero_f(core, c.ng_used, (size_t)N * KAI_GMAX)); TRY(dzero_f(core, c.ng_rel, (size_t)N * KAI_GMAX)); TRY(dzero_f(core, c.ng_alloc, (size_t)N * KAI_GMAX));
        TRY(dzero_f(core, c.ng_mark, (size_t)N)); TRY(dzero_f(core, c.ng_has_alloc, (size_t)N));
        TRY(dupload_f(core, c.next_new_group, &next_new, (size_t)1)); core->next_group0 = next_new;
        c.shared_on = shared ? 1 : 0;
        // shared GPUs: the pods that ask for a fraction are in no class (brute-force scans over the nodes' GPU groups), the other classes stay indexed with the gpusharingorder bit in their
        // keys (kai_engine.hpp key_shared_layout; KAI_SHARED_INDEX=0: every scan by brute force, as before round 6); MIG rows: every scan by brute force (the class keys do not know the MIG predicates)
        if (shared || sp.mig) { c.all_tracked = 0; const char* e = std::getenv("KAI_SHARED_INDEX"); const int lvl = e ? std::atoi(e) : 2; if (sp.mig || lvl == 0) c.use_index = 0; if (sp.mig || lvl < 2) { c.fast_ok = 0; core->fast_ok0 = 0; } }  // (2: gangs without a fraction pod also keep the staged job path)
        // each node's active pods in UID order: the shared-GPU guards of addTaskResources are order sensitive (nodes_fake/nodes.go:289-302 adds tasks by UID)
        std::vector<int32_t> np_off((size_t)N + 1, 0), np_pods;
        if (shared) {
            std::vector<int> order; for (int p = 0; p < P; p++) { const int st = s->pod_status[p], n = prep.pod_node[p]; if (n >= 0 && (st & (KAI_POD_ALLOCATED | KAI_POD_PIPELINED | KAI_POD_BINDING | KAI_POD_BOUND | KAI_POD_RUNNING | KAI_POD_RELEASING))) { order.push_back(p); np_off[n + 1]++; } }
            std::sort(order.begin(), order.end(), [&](int a, int b) { return s->pod_uid_rank[a] < s->pod_uid_rank[b]; });
            for (int n = 0; n < N; n++) np_off[n + 1] += np_off[n];
            np_pods.resize(order.size()); std::vector<int32_t> fillp(np_off.begin(), np_off.end() - 1);
            for (int p : order) np_pods[fillp[prep.pod_node[p]]++] = p;
        }
        { const int32_t* t; TRY(dupload(core, &t, np_off.data(), np_off.size())); core->d_np_off = const_cast<int32_t*>(t);
          TRY(dupload(core, &t, np_pods.data(), np_pods.size())); core->d_np_pods = const_cast<int32_t*>(t); }
    }
    // ---- topologies + sub-group tree
    c.T = prep.T; c.TL = prep.TL; c.D = prep.D; c.G = prep.G; c.W = (N + 31) / 32;
    TRY(dupload_f(core, c.topo_level_off, prep.topo_level_off.data(), prep.topo_level_off.size())); TRY(dupload_f(core, c.node_domain, prep.node_domain.data(), prep.node_domain.size()));
    TRY(dupload_f(core, c.dom_level, prep.dom_level.data(), prep.dom_level.size())); TRY(dupload_f(core, c.dom_topo, prep.dom_topo.data(), prep.dom_topo.size()));
    TRY(dupload_f(core, c.dom_parent, prep.dom_parent.data(), prep.dom_parent.size())); TRY(dupload_f(core, c.dom_id_rank, prep.dom_id_rank.data(), prep.dom_id_rank.size()));
    TRY(dupload_f(core, c.dom_child_off, prep.dom_child_off.data(), prep.dom_child_off.size())); TRY(dupload_f(core, c.dom_children, prep.dom_children.data(), prep.dom_children.size()));
    { const size_t DT = (size_t)prep.D + prep.T;
      TRY(dzero_f(core, c.dom_alloc_pods, DT)); TRY(dzero_f(core, c.dom_free, DT * KAI_MAX_RES)); TRY(dzero_f(core, c.dom_tmp, 3 * DT + 4)); TRY(dzero_f(core, c.dom_ratio, DT)); TRY(dzero_f(core, c.dom_key, 2 * DT));
      TRY(dzero_f(core, c.ns_bits, (size_t)KAI_TDEPTH * std::max(c.W, 1))); TRY(dzero_f(core, c.ns_sets, (size_t)KAI_TDEPTH * (DT + 1)));
      TRY(dzero_f(core, c.sg_score, (size_t)KAI_TKEYS * std::max<size_t>(DT, 1))); TRY(dzero_f(core, c.sg_key, (size_t)KAI_TKEYS)); TRY(dzero_f(core, c.sg_row, (size_t)KAI_TKEYS)); }
    {   // no sub-group tree in the snapshot (HostPrep::build_topology's identity tables: one root group per job, G = J): no parent, no topology, no required / preferred level
        // anywhere (-1), name rank 0, no children, and a pod-set's group is its job — constants written on the device, the pod-sets' groups copied from s_job (in HBM already).
        // (One allocation order either way: the session's layout in HBM does not depend on what is sent.)
        const bool lean_groups = prep.groups_default && !full_uploads;
        auto minus_one = [&](auto& field, size_t n) -> int { int rc2 = dalloc_f(core, field, n); if (rc2) return rc2; HIP_TRY(core, hipMemsetAsync(KAI_VP(field), 0xFF, std::max<size_t>(n, 1) * sizeof(*field), core->stream)); return KAI_OK; };
        TRY(dupload_f(core, c.g_job, prep.g_job.data(), prep.g_job.size()));
        if (lean_groups) { TRY(minus_one(c.g_parent, (size_t)J)); TRY(dzero_f(core, c.g_name_rank, (size_t)J)); TRY(minus_one(c.g_topo, (size_t)J)); TRY(minus_one(c.g_req, (size_t)J)); TRY(minus_one(c.g_pref, (size_t)J)); }
        else { TRY(dupload_f(core, c.g_parent, prep.g_parent.data(), prep.g_parent.size())); TRY(dupload_f(core, c.g_name_rank, prep.g_name_rank.data(), prep.g_name_rank.size())); TRY(dupload_f(core, c.g_topo, prep.g_topo.data(), prep.g_topo.size()));
               TRY(dupload_f(core, c.g_req, prep.g_req.data(), prep.g_req.size())); TRY(dupload_f(core, c.g_pref, prep.g_pref.data(), prep.g_pref.size())); }
        TRY(dupload_f(core, c.j_root_group, prep.j_root_group.data(), prep.j_root_group.size()));
        if (lean_groups) TRY(dzero_f(core, c.g_child_off, (size_t)J + 1)); else TRY(dupload_f(core, c.g_child_off, prep.g_child_off.data(), prep.g_child_off.size()));
        TRY(dupload_f(core, c.g_children, prep.g_children.data(), prep.g_children.size()));
        if (lean_groups) {
            TRY(dalloc_f(core, c.s_group, (size_t)S));
            if (S) HIP_TRY(core, hipMemcpyAsync(KAI_VP(c.s_group), KAI_VP(c.s_job), (size_t)S * sizeof(int32_t), hipMemcpyDeviceToDevice, core->stream));
            TRY(minus_one(c.s_topo, (size_t)S)); TRY(minus_one(c.s_req, (size_t)S)); TRY(minus_one(c.s_pref, (size_t)S));
            TRY(dzero_f(core, c.j_has_topology, (size_t)std::max(J, 1)));
        } else {
            TRY(dupload_f(core, c.s_group, prep.s_group.data(), prep.s_group.size())); TRY(dupload_f(core, c.s_topo, prep.s_topo.data(), prep.s_topo.size()));
            TRY(dupload_f(core, c.s_req, prep.s_req.data(), prep.s_req.size())); TRY(dupload_f(core, c.s_pref, prep.s_pref.data(), prep.s_pref.size()));
            TRY(dupload_f(core, c.j_has_topology, prep.j_has_topology.data(), prep.j_has_topology.size()));
        }
    }

    // ---- dynamic state
    double* d;
    TRY(dalloc_f(core, c.n_idle, (size_t)R * N)); (void)d;
    if ((size_t)R * N) HIP_TRY(core, hipMemcpyAsync(KAI_VP(c.n_idle), KAI_VP(c.n_alloc), (size_t)R * N * sizeof(double), hipMemcpyDeviceToDevice, core->stream));  // NewNodeInfo: Idle = Allocatable
    TRY(dzero_f(core, c.n_rel, (size_t)R * N)); TRY(dzero_f(core, c.n_used, (size_t)R * N));
    TRY(dupload_f(core, c.p_node, prep.pod_node.data(), (size_t)P));
    TRY(dzero_f(core, c.p_on_node, (size_t)P)); TRY(dzero_f(core, c.p_on_node_status, (size_t)P)); TRY(dzero_f(core, c.p_virtual, (size_t)P)); TRY(dzero_f(core, c.p_accepted, (size_t)P));
    TRY(dzero_f(core, c.s_active_alloc, (size_t)S)); TRY(dzero_f(core, c.s_active_used, (size_t)S)); TRY(dzero_f(core, c.s_alive, (size_t)S)); TRY(dzero_f(core, c.s_gated, (size_t)S)); TRY(dzero_f(core, c.s_pipelined, (size_t)S));
    TRY(dzero_f(core, c.j_n_pending, (size_t)J)); TRY(dzero_f(core, c.j_tta_valid, (size_t)J)); TRY(dzero_f(core, c.j_tta_n, (size_t)J)); TRY(dzero_f(core, c.tta, (size_t)P));
    TRY(dzero_f(core, c.j_tta_res, (size_t)4 * J)); TRY(dzero_f(core, c.j_allocated, (size_t)4 * J));
    TRY(dzero_f(core, c.lq_sorted, (size_t)J)); TRY(dzero_f(core, c.lq_side, (size_t)J)); TRY(dzero_f(core, c.lq_cur, (size_t)Q)); TRY(dzero_f(core, c.lq_end, (size_t)Q)); TRY(dzero_f(core, c.lq_side_len, (size_t)Q));
    TRY(dzero_f(core, c.j_state, (size_t)J));
    TRY(dzero_f(core, c.qheap, (size_t)Q + 1)); TRY(dzero_f(core, c.root_heap, (size_t)Q + 1)); TRY(dzero_f(core, c.qn, (size_t)Q + 1));
    c.ops_cap = 4 * P + 64; TRY(dalloc_f(core, c.ops, (size_t)c.ops_cap));
    c.out_cap = (int64_t)2 * P + 64; TRY(dalloc_f(core, c.out_ops, (size_t)c.out_cap));
    TRY(dzero_f(core, c.scratch, (size_t)P + 64));
    TRY(dzero_f(core, c.st, (size_t)1));
    TRY(dzero(core, &core->d_jsum, (size_t)9 * J));
    TRY(dzero(core, &core->d_weight, (size_t)3 * Q)); TRY(dzero(core, &core->d_rem_amt, (size_t)3 * Q)); TRY(dzero(core, &core->d_rem_has, (size_t)3 * Q));
    TRY(dzero(core, &core->d_best_out, (size_t)2));
    TRY(dupload_f(core, c.q_share, prep.shares.data(), prep.shares.size()));
    // batch path of the allocate action (kai_batch.hpp): its pools and tables, when the snapshot admits it
    core->shape = prep.shape;
    {   int rcb = batch_bind(c, prep,
            [&](size_t bytes) -> void* { char* p = nullptr; if (dalloc(core, &p, bytes)) return nullptr; if (hipMemsetAsync(p, 0, bytes, core->stream) != hipSuccess) return nullptr; return p; },
            [&](void* d, const void* h, size_t n) -> int { return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, core->stream) == hipSuccess ? 0 : (int)KAI_ERR_HIP; },
            core->world, core->rank, core->shard_k);
        if (rcb) return fail(core, rcb, "batch path buffers");
        if (shared) c.bt.enabled = 0; }
    // keep the initial dynamic state in HBM so that kai_session_reset needs no host traffic
    TRY(dalloc(core, &core->d_status0, (size_t)P)); TRY(dalloc(core, &core->d_node0, (size_t)P)); TRY(dalloc(core, &core->d_shares0, (size_t)std::max(Q, 1) * 3));
#undef TRY
    if (P) { HIP_TRY(core, hipMemcpyAsync(core->d_status0, KAI_VP(c.p_status), (size_t)P * 4, hipMemcpyDeviceToDevice, core->stream));
             HIP_TRY(core, hipMemcpyAsync(core->d_node0, KAI_VP(c.p_node), (size_t)P * 4, hipMemcpyDeviceToDevice, core->stream)); }
    HIP_TRY(core, hipMemcpyAsync(core->d_shares0, KAI_VP(c.q_share), (size_t)std::max(Q, 1) * 3 * sizeof(QShare), hipMemcpyDeviceToDevice, core->stream));
    { KaiCtx* t = nullptr; int rc3 = dalloc(core, &t, (size_t)1); if (rc3) return rc3; core->d_ctx = t; }
    HIP_TRY(core, hipMemcpyAsync(core->d_ctx, &core->ctx, sizeof(KaiCtx), hipMemcpyHostToDevice, core->stream));
    if (std::getenv("KAI_OPEN_DIGEST")) {  // diagnostic: what the open put into every array of the session context, before its first kernel (FNV-1a per KaiCtx field; tests/test_open_uploads.py)
        HIP_TRY(core, hipStreamSynchronize(core->stream));
        for (const kai_core::AllocRec& a : core->allocs) {
            std::vector<unsigned char> h(a.bytes);
            HIP_TRY(core, hipMemcpyAsync(h.data(), a.base, a.bytes, hipMemcpyDeviceToHost, core->stream)); HIP_TRY(core, hipStreamSynchronize(core->stream));
            uint64_t f = 1469598103934665603ull; for (unsigned char x : h) { f ^= x; f *= 1099511628211ull; }
            std::fprintf(stderr, "kai open digest: field %zu bytes %zu fnv %016llx\n", a.field_off, a.bytes, (unsigned long long)f);
        }
    }
    { int rc2 = launch_open_kernels(core); if (rc2) return rc2; core->index_stale = false; }
    HIP_TRY(core, hipEventRecord(core->ev1, core->stream));
    const auto t_enq = tnow();
    HIP_TRY(core, hipStreamSynchronize(core->stream));  // prep's host buffers die with this scope
    if (prof_open) std::fprintf(stderr, "kai open: free %.2f ms, checks + shared pods %.2f, the snapshot's arrays staged %.2f, host prep %.2f, allocate + enqueue uploads %.2f, wait %.2f | total %.2f ms\n",
                                tms(t_begin, t_freed), tms(t_freed, t_shared), tms(t_shared, t_staged), tms(t_staged, t_prep), tms(t_prep, t_enq), tms(t_enq, tnow()), tms(t_begin, tnow()));
    if (prof_open) std::fprintf(stderr, "kai open: host prep by phase: range checks %.2f, nodes %.2f, pods %.2f, task order %.2f, queues + job lists %.2f, shares + topology %.2f, classes %.2f, batch shape %.2f ms on %d host threads\n",
                                prep.phase_ms[0], prep.phase_ms[1], prep.phase_ms[2], prep.phase_ms[3], prep.phase_ms[4], prep.phase_ms[5], prep.phase_ms[6], prep.phase_ms[7], host_threads());
    float ms = 0; HIP_TRY(core, hipEventElapsedTime(&ms, core->ev0, core->ev1));
    std::memset(&core->stats, 0, sizeof(core->stats));
    core->stats.upload_ms = ms;
    // slabs of earlier (larger) sessions this open did not take: keep two for what the session's actions still allocate (solver scratch, batch pools), release the rest
    // ... and the smallest one that holds the LARGEST request the last session made after its open (post_open_max: a victim action's replicas, the batch pools) — or that request
    // would pay a hipMalloc every cycle and a hipFree at the next trim, which is what keeping slabs is there to avoid
    if (core->spare.size() > 2) {
        std::sort(core->spare.begin(), core->spare.end(), [](const std::pair<void*, size_t>& a, const std::pair<void*, size_t>& b) { return a.second < b.second; });
        size_t keep_big = core->spare.size();
        for (size_t i = 2; i < core->spare.size(); i++) if (core->post_open_max > 0 && core->spare[i].second >= core->post_open_max) { keep_big = i; break; }
        if (core->post_open_max > 0 && (core->spare[0].second >= core->post_open_max || core->spare[1].second >= core->post_open_max)) keep_big = core->spare.size();  // (one of the two small ones already holds it)
        std::vector<std::pair<void*, size_t>> kept;
        for (size_t i = 0; i < core->spare.size(); i++) { if (i < 2 || i == keep_big) kept.push_back(core->spare[i]); else (void)hipFree(core->spare[i].first); }
        core->spare.swap(kept);
    }
    core->post_open_max = 0;  // (counted from here on: dalloc)
    core->open = true; core->err = "ok";
    return KAI_OK;
}
// Nothing leaves this function by an exception (the host preparation's loops allocate: std::bad_alloc, also from a worker thread — kai_parallel.hpp carries it here — and
// std::system_error from thread creation), and every failure takes ONE way out: the stream is drained — the staged copies read the handle's pinned buffer, which the next open
// writes again — and the slabs this open took go back to the handle, so that a failed open leaves the handle as a closed session does.
int kai_session_open(kai_core* core, const kai_snapshot_soa* s) {
    if (!core || !s) return KAI_ERR_INVALID_ARG;
    int rc;
    try { rc = session_open_impl(core, s); }
    catch (const std::bad_alloc&) { core->err = "kai_session_open: out of host memory"; rc = KAI_ERR_NO_MEMORY; }
    catch (const std::exception& e) { core->err = std::string("kai_session_open: ") + e.what(); rc = KAI_ERR_INVALID_ARG; }
    catch (...) { core->err = "kai_session_open: unknown exception"; rc = KAI_ERR_INVALID_ARG; }
    if (rc != KAI_OK) {
        const std::string why = core->err;
        if (core->stream) (void)hipStreamSynchronize(core->stream);
        free_session(core);
        core->err = why;
    }
    return rc;
}

int kai_session_reset(kai_core* core) {
    if (!core) return KAI_ERR_INVALID_ARG;
    if (!core->open) return fail(core, KAI_ERR_STATE, "no open session");
    HIP_TRY(core, hipSetDevice(core->device));
    KaiCtx& c = core->ctx;
    const size_t RN = (size_t)c.R * c.N;
    HIP_TRY(core, hipEventRecord(core->ev0, core->stream));
    if (RN) { HIP_TRY(core, hipMemcpyAsync(KAI_VP(c.n_idle), KAI_VP(c.n_alloc), RN * 8, hipMemcpyDeviceToDevice, core->stream));
              HIP_TRY(core, hipMemsetAsync(KAI_VP(c.n_rel), 0, RN * 8, core->stream)); HIP_TRY(core, hipMemsetAsync(KAI_VP(c.n_used), 0, RN * 8, core->stream)); }
    if (c.P) { HIP_TRY(core, hipMemcpyAsync(KAI_VP(c.p_status), core->d_status0, (size_t)c.P * 4, hipMemcpyDeviceToDevice, core->stream));
               HIP_TRY(core, hipMemcpyAsync(KAI_VP(c.p_node), core->d_node0, (size_t)c.P * 4, hipMemcpyDeviceToDevice, core->stream)); }
    HIP_TRY(core, hipMemcpyAsync(KAI_VP(c.q_share), core->d_shares0, (size_t)std::max(c.Q, 1) * 3 * sizeof(QShare), hipMemcpyDeviceToDevice, core->stream));
    c.fast_ok = core->fast_ok0;
    if (c.P) HIP_TRY(core, hipMemcpyAsync(KAI_VP(c.p_group), core->d_group0, (size_t)c.P * 4, hipMemcpyDeviceToDevice, core->stream));
    HIP_TRY(core, hipMemcpyAsync(KAI_VP(c.next_new_group), &core->next_group0, 4, hipMemcpyHostToDevice, core->stream));
    { const size_t NG = (size_t)std::max(c.N, 1) * KAI_GMAX;
      HIP_TRY(core, hipMemsetAsync(KAI_VP(c.ng_used), 0, NG * 8, core->stream)); HIP_TRY(core, hipMemsetAsync(KAI_VP(c.ng_rel), 0, NG * 8, core->stream)); HIP_TRY(core, hipMemsetAsync(KAI_VP(c.ng_alloc), 0, NG * 8, core->stream));
      HIP_TRY(core, hipMemsetAsync(KAI_VP(c.ng_mark), 0, (size_t)std::max(c.N, 1) * 4, core->stream)); HIP_TRY(core, hipMemsetAsync(KAI_VP(c.ng_has_alloc), 0, (size_t)std::max(c.N, 1) * 4, core->stream)); }
    if (core->solver_ready) HIP_TRY(core, hipMemsetAsync(c.sv.xr_key, 0xFF, sizeof(int64_t) * ((size_t)c.sv.xr_mask + 1), core->stream));
    HIP_TRY(core, hipMemsetAsync(KAI_VP(c.st), 0, sizeof(EngineState), core->stream));
    int rc = launch_open_kernels(core); if (rc) return rc;
    core->index_stale = false;
    HIP_TRY(core, hipEventRecord(core->ev1, core->stream));
    HIP_TRY(core, hipStreamSynchronize(core->stream));
    float ms = 0; HIP_TRY(core, hipEventElapsedTime(&ms, core->ev0, core->ev1));
    core->stats.upload_ms = ms;
    return KAI_OK;
}

int kai_queue_shares(kai_core* core, kai_queue_share* out, int cap) {
    if (!core || !out) return KAI_ERR_INVALID_ARG;
    if (!core->open) return fail(core, KAI_ERR_STATE, "no open session");
    const int Q = core->ctx.Q;
    if (cap < Q) return fail(core, KAI_ERR_CAPACITY, "kai_queue_shares: cap < n_queues");
    HIP_TRY(core, hipSetDevice(core->device));
    std::vector<QShare> h((size_t)std::max(Q, 1) * 3);
    HIP_TRY(core, hipMemcpyAsync(h.data(), KAI_VP(core->ctx.q_share), (size_t)Q * 3 * sizeof(QShare), hipMemcpyDeviceToHost, core->stream));
    HIP_TRY(core, hipStreamSynchronize(core->stream));
    for (int q = 0; q < Q; q++) for (int k = 0; k < 3; k++) {
        const QShare& x = h[(size_t)q * 3 + k];
        out[q].fair_share[k] = x.fair; out[q].allocated[k] = x.allocated; out[q].allocated_non_preemptible[k] = x.allocated_np;
        out[q].request[k] = x.request; out[q].deserved[k] = x.deserved; out[q].max_allowed[k] = x.max_allowed;
    }
    return KAI_OK;
}

int kai_action_execute(kai_core* core, int action, kai_op* ops_out, int64_t ops_cap, int64_t* n_ops) {
    if (!core || !n_ops) return KAI_ERR_INVALID_ARG;
    if (!core->open) return fail(core, KAI_ERR_STATE, "no open session");
    if (action < KAI_ACTION_ALLOCATE || action > KAI_ACTION_PREEMPT) return fail(core, KAI_ERR_INVALID_ARG, "unknown action");
    const bool victim = action != KAI_ACTION_ALLOCATE;
    if (victim && core->cfg.use_scheduling_signatures && !core->ctx.j_signature && core->ctx.J > 0) return fail(core, KAI_ERR_UNSUPPORTED, "use_scheduling_signatures needs kai_snapshot_soa.job_signature (actions/common/minimal_job_comparison.go)");
    HIP_TRY(core, hipSetDevice(core->device));
    const auto ta0 = std::chrono::steady_clock::now();
    auto ta_ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    KaiCtx& c = core->ctx;
    if (victim && !core->solver_ready) {  // scratch of the victim search, kept for the rest of the session
        char* base = nullptr; size_t bytes = solver_scratch_bytes(c.N, c.P, c.S, c.J, c.Q, c.W, c.D + c.T, c.TL, c.G);
        int rc0 = dalloc(core, &base, bytes); if (rc0) return rc0;
        HIP_TRY(core, hipMemsetAsync(base, 0, bytes, core->stream));
        solver_scratch_bind(c.sv, base, c.N, c.P, c.S, c.J, c.Q, c.W, c.D + c.T, c.TL, c.G);
        core->sv_base = base; core->sv_bytes = bytes;
        HIP_TRY(core, hipMemsetAsync(c.sv.xr_key, 0xFF, sizeof(int64_t) * ((size_t)c.sv.xr_mask + 1), core->stream));  // empty residency table
        { int32_t* xg = nullptr; int rcx = dalloc(core, &xg, (size_t)c.sv.xr_mask + 1); if (rcx) return rcx; c.sv.xr_group = xg; core->xr_base = (char*)xg; core->xr_bytes = ((size_t)c.sv.xr_mask + 1) * 4; }
        core->solver_ready = true;
    }
    { int d = core->cfg.queue_depth[action]; c.queue_depth = d > 0 ? d : 0; c.action = action; }
    HIP_TRY(core, hipMemcpyAsync(core->d_ctx, &core->ctx, sizeof(KaiCtx), hipMemcpyHostToDevice, core->stream));
    // reset the per-action scalars, keep the proportion totals
    EngineState st{};
    HIP_TRY(core, hipMemcpyAsync(&st, KAI_VP(c.st), sizeof(st), hipMemcpyDeviceToHost, core->stream));
    HIP_TRY(core, hipStreamSynchronize(core->stream));
    double total[3] = {st.total[0], st.total[1], st.total[2]};
    st = EngineState{}; st.total[0] = total[0]; st.total[1] = total[1]; st.total[2] = total[2];
    HIP_TRY(core, hipMemcpyAsync(KAI_VP(c.st), &st, sizeof(st), hipMemcpyHostToDevice, core->stream));
    HIP_TRY(core, hipEventRecord(core->ev0, core->stream));
    const int TB = 256;
    if (core->index_stale) { if (c.use_index && c.NB) hipLaunchKernelGGL(k_index_build, dim3((c.NB + 3) / 4), dim3(TB), 0, core->stream, c); core->index_stale = false; }
    if (c.J) hipLaunchKernelGGL(k_job_init, dim3((c.J + TB - 1) / TB), dim3(TB), 0, core->stream, c);
    if (c.Q) hipLaunchKernelGGL(k_leaf_init, dim3((c.Q + 3) / 4), dim3(TB), 0, core->stream, c);
    BatchStats bs; core->batch_plan_ms = core->batch_fill_ms = core->batch_apply_ms = 0; int g_run = 1, scan_wgs_used = 1; bool xsh = false;
    if (!victim) {  // the batch path (plan / fill / apply rounds, kai_batch.hpp) when the action qualifies
        DevLauncher dl{core};
        int rcb = batch_allocate(dl, c, core->shape, bs);
        if (rcb) { if (core->err == "ok") core->err = "batch path failed"; return rcb; }
        if (bs.ran && bs.buckets) core->index_stale = true;
        if (bs.ran && !bs.st_on_device) {
            EngineState sb{};
            HIP_TRY(core, hipMemcpyAsync(&sb, KAI_VP(c.st), sizeof(sb), hipMemcpyDeviceToHost, core->stream));
            HIP_TRY(core, hipStreamSynchronize(core->stream));
            sb.decisions += bs.decisions; sb.jobs_attempted += bs.attempted; sb.jobs_committed += bs.committed; sb.rollbacks += bs.rollbacks; sb.out_len += bs.ops; sb.stmts += bs.committed;
            sb.index_queries += bs.decisions; sb.drain_pending = bs.drain;
            HIP_TRY(core, hipMemcpyAsync(KAI_VP(c.st), &sb, sizeof(sb), hipMemcpyHostToDevice, core->stream));
        }
    }
    const auto ta1 = std::chrono::steady_clock::now();
    if (!bs.ran && core->world > 1) {
        // A node-sharded group shards the batch path's fill.  Every other action — sub-group trees, topology, elastic jobs, the victim actions — runs REPLICATED: every rank
        // holds the whole session (only the fill's index is sharded), runs the same engine on it and commits the same operations; nothing is exchanged and nothing gets
        // faster, but a cycle that mixes both kinds of action (BASELINE config 4: allocate, consolidation, reclaim) runs on the group with the one-rank results.
        // The sharded fill left a class index of the own slice only: rebuild it over all nodes first.
        if (c.use_index && c.NB) hipLaunchKernelGGL(k_index_build, dim3((c.NB + 3) / 4), dim3(TB), 0, core->stream, c);
    }
    if (!bs.ran) {  // dynamic LDS: upper levels of the class index, plus the job-order tree when it fits beside them (160 KiB per CU)
        size_t idx_b = lds_index_bytes(c.C, c.NSB), tree_b = lds_tree_bytes(c.Q);
        const size_t budget = 160 * 1024 - 16384;  // static LDS of the kernel (mailbox, context, engine scalars, frame: 6.8 KB, llvm-readelf .group_segment_fixed_size) + margin
        int tree_in_lds = (idx_b + tree_b <= budget && !std::getenv("KAI_TREE_IN_HBM")) ? 1 : 0;
        size_t dyn = idx_b + (tree_in_lds ? tree_b : 0);
        // a victim action runs on several workgroups, each on a replica of the session arrays made right here (after k_job_init / k_leaf_init on the stream)
        if (victim) {
            int want = 32; if (const char* e = std::getenv("KAI_VICTIM_WGS")) want = std::atoi(e);  // (measured on C4: 32 workgroups — four replicas per XCD, hot in its L2 — beat 64 and more, whose waves are no shorter)
            int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, core->device) != hipSuccess || cus <= 0) cus = 64;
            want = std::max(1, std::min(std::min(want, (int)KAI_MW_MAX), cus));  // every workgroup must be resident: they meet at a grid barrier
            { int per_cu = 0;  // ... as the runtime's occupancy calculation sees it for this kernel, its workgroup size and its dynamic LDS (not only one per compute unit)
              if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_action<true, false>), WG, dyn) == hipSuccess && per_cu > 0) want = std::min(want, per_cu * cus);
              else (void)hipGetLastError(); }
            // a node-sharded group with an exchange for it deals the waves out over its ranks (kai_victim_shard.hpp); every rank takes the same decision here
            xsh = core->world > 1 && (core->xag_fn || core->rccl_comm) && !core->shared && want > 1 && (core->mw_world == 0 || core->mw_world == want) && !std::getenv("KAI_VICTIM_REPLICATED");
            if (xsh) {
                if (int rcx = xshard_setup(core)) return rcx;
                int cap = xw_default_cap(core->world, want); if (const char* e = std::getenv("KAI_VICTIM_XCAP")) { const int v = std::atoi(e); if (v > 0) cap = std::max(core->world, std::min(v, (int)KAI_MW_WAVE)); }
                c.mw_xworld = core->world; c.mw_xrank = core->rank; c.mw_xcap = cap; c.mw_mail = core->d_mail;
                core->xs.begin(core->world, core->rank, cap);
                __atomic_store_n(&core->mail->req, 0, __ATOMIC_RELEASE); __atomic_store_n(&core->mail->resp, 0, __ATOMIC_RELEASE);
            }
            int rcm = prepare_multi(core, want, &g_run);
            c.mw_xworld = 0; c.mw_xrank = 0; c.mw_xcap = 0; c.mw_mail = nullptr;  // (the host copy is what every other kernel of the session gets by value)
            if (rcm) return rcm;
            if (xsh && g_run <= 1) xsh = false;  // (one engine: the action runs replicated; the conditions are the same on every rank — a failed allocation is an error above)
        }
        // the allocate action of a large cluster: helper workgroups beside the engine's take the passes over the nodes (ScanGrid, kai_kernels.hpp)
        int scan_wgs = 1;
        if (!victim) {
            int want = c.N >= 4096 ? 32 : c.N >= 1024 ? 16 : 1; if (const char* e = std::getenv("KAI_SCAN_WGS")) want = std::atoi(e);  // (measured on the mixed config 5 and on config 3 with fractions: 16 .. 32 workgroups are the fastest, 64 and more pay for the table they all watch; from about a thousand nodes a pass on the grid beats one on this workgroup alone: 1 500 nodes 355 -> 303 ms, 2 500: 707 -> 517, 3 500: 1 143 -> 725 per cycle of config 3 with fractions)
            int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, core->device) != hipSuccess || cus <= 0) cus = 64;
            want = std::max(1, std::min(std::min(want, (int)KAI_SG_MAX), cus / 2));  // every workgroup must be resident (the control lane waits for the helpers); half the chip leaves room for a neighbour
            if (want > 1 && !core->d_sg) { ScanGrid* g = nullptr; int rcg = dalloc(core, &g, (size_t)1); if (rcg) return rcg; core->d_sg = g; }
            if (want > 1) {
                HIP_TRY(core, hipMemsetAsync(core->d_sg, 0, sizeof(ScanGrid), core->stream));
                c.sg = core->d_sg; c.sg_wgs = want; c.sg_per = (((c.N + want - 1) / want + 63) / 64) * 64;
                HIP_TRY(core, hipMemcpyAsync(core->d_ctx, &core->ctx, sizeof(KaiCtx), hipMemcpyHostToDevice, core->stream));
                HIP_TRY(core, hipStreamSynchronize(core->stream));
                c.sg = nullptr; c.sg_wgs = 0; c.sg_per = 0;  // (the host copy is what every other kernel of the session gets by value)
                scan_wgs = want;
            }
        }
        auto launch = [&](auto kernel) -> int {
            HIP_TRY(core, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            hipLaunchKernelGGL(kernel, dim3(victim ? g_run : scan_wgs), dim3(WG), dyn, core->stream, g_run > 1 ? (const KaiCtx*)core->d_ctxs : (const KaiCtx*)core->d_ctx, action, tree_in_lds, victim ? 1 : scan_wgs);
            return KAI_OK;
        };
        int rcl = victim ? launch(k_action<true, false>) : tree_in_lds ? launch(k_action<false, true>) : launch(k_action<false, false>);
        if (rcl) return rcl;
        scan_wgs_used = scan_wgs;
        if (xsh) { HIP_TRY(core, hipGetLastError()); if (int rcs = xshard_serve(core)) return rcs; }  // the kernel is running: carry its waves' exchanges until it ends
    }
    if (c.J && !victim) hipLaunchKernelGGL(k_drain, dim3(std::min(2048, (c.J + TB - 1) / TB)), dim3(TB), 0, core->stream, c, core->d_slot_queue);
    HIP_TRY(core, hipGetLastError());
    HIP_TRY(core, hipEventRecord(core->ev1, core->stream));
    HIP_TRY(core, hipMemcpyAsync(&st, KAI_VP(c.st), sizeof(st), hipMemcpyDeviceToHost, core->stream));
    HIP_TRY(core, hipStreamSynchronize(core->stream));
    float ms = 0; HIP_TRY(core, hipEventElapsedTime(&ms, core->ev0, core->ev1));
    double upload = core->stats.upload_ms;
    std::memset(&core->stats, 0, sizeof(core->stats));
    core->stats.upload_ms = upload; core->stats.kernel_ms = ms; core->stats.decisions = st.decisions; core->stats.node_scans = st.node_scans;
    core->stats.nodes_scanned = st.nodes_scanned; core->stats.jobs_attempted = st.jobs_attempted; core->stats.jobs_committed = st.jobs_committed; core->stats.rollbacks = st.rollbacks;
    core->stats.reserved[0] = st.index_queries; core->stats.reserved[1] = st.index_refreshes; core->stats.reserved[2] = victim ? st.scenarios : st.drained_jobs; core->stats.reserved[3] = victim ? st.simulations : st.drained_decisions;
    // reserved[4] = plan/fill rounds of the batch path (0 = the sequential engine ran the action); batch path: [5] fill-wave cycles, [6] mispredicted jobs,
    // [7] plan / fill / apply time in microseconds, 21 bits each; sequential engine: [5..7] control-lane cycles allocate / commit+discard / total
    if (bs.ran) {
        core->stats.reserved[4] = bs.rounds; core->stats.reserved[5] = bs.fill_cycles; core->stats.reserved[6] = bs.mismatches;
        auto us = [](double ms) { int64_t v = (int64_t)(ms * 1000.0); return v < 0 ? (int64_t)0 : v > 0x1fffff ? (int64_t)0x1fffff : v; };
        core->stats.reserved[7] = (us(core->batch_plan_ms) << 42) | (us(core->batch_fill_ms) << 21) | us(core->batch_apply_ms);
        core->stats.reserved[1] = (bs.block_loads & ((1ll << 48) - 1)) | ((int64_t)(bs.buckets ? 1 : 0) << 62) | ((int64_t)(bs.buckets >= 2 ? 1 : 0) << 61) | ((int64_t)(bs.buckets == 3 ? 1 : 0) << 60) | ((int64_t)(bs.dev_loop ? 1 : 0) << 59); /* bit 59: the round loop's state lived on the device (rounds without the host), bit 62: the fill ran on the sets by free devices (kai_fill_buckets.hpp), bit 61: behind a counting machine (kai_fill_counts.hpp), bit 60: with a wavefront per level (kai_fill_levels.hpp) */ core->stats.reserved[0] = core->world > 1 ? bs.exchanges : core->stats.reserved[0];
        if (std::getenv("KAI_PROF")) std::fprintf(stderr, "kai batch%s: rounds %lld mismatches %lld planned %lld max_h %d | fill cycles %lld load %lld update %lld rescan %lld | block loads %lld rescans %lld %lld %lld | plan %.3f ms fill %.3f ms apply %.3f ms\n",
            bs.buckets ? " (bucket fill)" : "", (long long)bs.rounds, (long long)bs.mismatches, (long long)bs.planned, bs.max_h, (long long)bs.fill_cycles, (long long)bs.fill_load, (long long)bs.fill_update, (long long)bs.fill_rescan,
            (long long)bs.block_loads, (long long)bs.rescans1, (long long)bs.rescans2, (long long)bs.rescans3, core->batch_plan_ms, core->batch_fill_ms, core->batch_apply_ms);
    } else { core->stats.reserved[4] = 0; core->stats.reserved[5] = st.prof[2]; core->stats.reserved[6] = st.prof[3]; core->stats.reserved[7] = st.prof[7]; }
    if (!victim && !bs.ran && scan_wgs_used > 1) {  // sequential allocate with a scan grid: bits 48.. of [1] = workgroups that took the passes over the nodes (the engine's + the helpers that signed on)
        int32_t nreg = 0; HIP_TRY(core, hipMemcpyAsync(&nreg, &core->d_sg->n_reg, sizeof nreg, hipMemcpyDeviceToHost, core->stream)); HIP_TRY(core, hipStreamSynchronize(core->stream));
        core->stats.reserved[1] |= (int64_t)(nreg + 1) << 48;
        if (std::getenv("KAI_PROF")) std::fprintf(stderr, "kai scan grid: %d workgroups launched, %d helpers signed on\n", scan_wgs_used, nreg);
    }
    if (xsh) {  // the action's closing message (every rank sends exactly one; skipped once another rank's was seen): a fault anywhere is everybody's
        DevXIo io{core}; int32_t mfault = 0;
        HIP_TRY(core, hipMemcpyAsync(&mfault, &core->d_mw->fault, 4, hipMemcpyDeviceToHost, core->stream)); HIP_TRY(core, hipStreamSynchronize(core->stream));
        const int rcf = core->xs.finish(io, (st.fault || mfault) ? 1 : 0);
        core->stats.reserved[7] = core->xs.exchanges;  // collectives of this action
        if (rcf && !st.fault) return fail(core, rcf, "victim action of a node-sharded group: another rank ended it with a fault");
    }
    if (victim) {  // victim actions: [1] workgroups the action ran on, [5] waves, [6] simulations run (speculative ones included) << 32 | simulations the reference's order reached; a group that dealt the waves out over its ranks: [7] collectives
        core->stats.reserved[1] = g_run;
        if (g_run > 1) {
            if (core->mw_host.empty()) core->mw_host.resize(1);
            MultiCtx& m = core->mw_host[0]; HIP_TRY(core, hipMemcpyAsync(&m, core->d_mw, sizeof(MultiCtx), hipMemcpyDeviceToHost, core->stream)); HIP_TRY(core, hipStreamSynchronize(core->stream));
            core->stats.reserved[5] = m.waves; core->stats.reserved[6] = (m.sims_run << 32) | (m.sims_used & 0xffffffffll);
            if (std::getenv("KAI_PROF")) std::fprintf(stderr, "kai victim: %d workgroups, waves %lld, simulations run %lld / counted %lld, replays %lld, fault %d\n", g_run, (long long)m.waves, (long long)m.sims_run, (long long)m.sims_used, (long long)m.replays, m.fault);
        }
    }
    if (std::getenv("KAI_PROF")) { std::fprintf(stderr, "kai prof:"); for (int i = 0; i < KAI_NPROF; i++) std::fprintf(stderr, " %lld", (long long)st.prof[i]); std::fprintf(stderr, "\n"); }
    if (st.non_allocate_commits) c.fast_ok = 0;  // the staged job path assumes nothing releasing / pipelined in the session (kai_host_prep.hpp); until the next open / reset
    if (st.fault) { char buf[96]; std::snprintf(buf, sizeof buf, "device engine fault code %d (engine source line %d)", st.fault, st.fault_line); core->err = buf; return KAI_ERR_DEVICE_FAULT; }
    *n_ops = st.out_len;
    const auto ta2 = std::chrono::steady_clock::now();
    if (ops_out) {
        if (st.out_len > ops_cap) return fail(core, KAI_ERR_CAPACITY, "kai_action_execute: ops_cap too small");
        // through the handle's own pinned buffer: a copy of some MB straight into the caller's pageable memory makes the runtime pin and unpin that memory around
        // the transfer, and the unpinning lands in a later call (measured: 22 ms in every other kai_session_reset of the C5 bench, profiles/r04e_*)
        const size_t bytes = (size_t)st.out_len * sizeof(kai_op);
        if (bytes > core->pin_bytes) {
            if (core->pin_buf) (void)hipHostFree(core->pin_buf);
            core->pin_buf = nullptr; core->pin_bytes = 0;
            const size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20);
            HIP_TRY(core, hipHostMalloc(&core->pin_buf, want, hipHostMallocDefault));
            core->pin_bytes = want;
        }
        if (st.out_len) HIP_TRY(core, hipMemcpyAsync(core->pin_buf, KAI_VP(c.out_ops), bytes, hipMemcpyDeviceToHost, core->stream));
        HIP_TRY(core, hipStreamSynchronize(core->stream));
        const kai_op* src = static_cast<const kai_op*>(core->pin_buf);
        const int32_t* perm = core->perm.data();
        parallel_chunks((size_t)st.out_len, [&](int, size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) { kai_op o = src[i]; if (o.node >= 0) o.node = perm[o.node]; ops_out[i] = o; } });  // name rank → caller's index (config 5: 150 k operations, 4.8 MB)
    }
    if (std::getenv("KAI_PROF")) { const auto ta3 = std::chrono::steady_clock::now(); std::fprintf(stderr, "kai action host clocks: setup + batch path %.2f ms, engine / drain / stats %.2f, operations to the caller %.2f | total %.2f ms\n", ta_ms(ta0, ta1), ta_ms(ta1, ta2), ta_ms(ta2, ta3), ta_ms(ta0, ta3)); }
    return KAI_OK;
}

int kai_best_node(kai_core* core, int32_t pod_idx, const uint32_t* nodeset_bitmap, int pipeline_only, int32_t* node_idx_out, int* is_pipeline_out) {
    if (!core || !node_idx_out) return KAI_ERR_INVALID_ARG;
    if (!core->open) return fail(core, KAI_ERR_STATE, "no open session");
    if (pod_idx < 0 || pod_idx >= core->ctx.P) return fail(core, KAI_ERR_INVALID_ARG, "pod index out of range");
    HIP_TRY(core, hipSetDevice(core->device));
    uint32_t* d_bits = nullptr;
    if (nodeset_bitmap) {  // caller's node indices → engine order (name rank), then to HBM for this call
        const int N = core->ctx.N, W = (N + 31) / 32;
        std::vector<uint32_t> bits((size_t)std::max(W, 1), 0u);
        for (int i = 0; i < N; i++) { int o = core->perm[i]; if ((nodeset_bitmap[o >> 5] >> (o & 31)) & 1u) bits[i >> 5] |= 1u << (i & 31); }
        HIP_TRY(core, hipMalloc(reinterpret_cast<void**>(&d_bits), bits.size() * 4));
        HIP_TRY(core, hipMemcpyAsync(d_bits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, core->stream));
        HIP_TRY(core, hipStreamSynchronize(core->stream));
    }
    hipLaunchKernelGGL(k_best_node, dim3(1), dim3(WG), 16, core->stream, core->ctx, (int)pod_idx, pipeline_only, core->d_best_out, (const uint32_t*)d_bits);
    HIP_TRY(core, hipGetLastError());
    int32_t h[2] = {-1, 0};
    HIP_TRY(core, hipMemcpyAsync(h, core->d_best_out, sizeof h, hipMemcpyDeviceToHost, core->stream));
    HIP_TRY(core, hipStreamSynchronize(core->stream));
    if (d_bits) (void)hipFree(d_bits);
    *node_idx_out = h[0] >= 0 ? core->perm[h[0]] : -1;
    if (is_pipeline_out) *is_pipeline_out = h[1];
    return KAI_OK;
}

int kai_pod_states(kai_core* core, int32_t* status_out, int32_t* node_out, int cap) {
    if (!core) return KAI_ERR_INVALID_ARG;
    if (!core->open) return fail(core, KAI_ERR_STATE, "no open session");
    const int P = core->ctx.P;
    if (cap < P) return fail(core, KAI_ERR_CAPACITY, "kai_pod_states: cap < n_pods");
    HIP_TRY(core, hipSetDevice(core->device));
    if (status_out && P) HIP_TRY(core, hipMemcpyAsync(status_out, KAI_VP(core->ctx.p_status), (size_t)P * 4, hipMemcpyDeviceToHost, core->stream));
    if (node_out && P) HIP_TRY(core, hipMemcpyAsync(node_out, KAI_VP(core->ctx.p_node), (size_t)P * 4, hipMemcpyDeviceToHost, core->stream));
    HIP_TRY(core, hipStreamSynchronize(core->stream));
    if (node_out) for (int p = 0; p < P; p++) if (node_out[p] >= 0) node_out[p] = core->perm[node_out[p]];
    return KAI_OK;
}

int kai_node_states(kai_core* core, kai_node_state* out, int cap) {
    if (!core || !out) return KAI_ERR_INVALID_ARG;
    if (!core->open) return fail(core, KAI_ERR_STATE, "no open session");
    const int N = core->ctx.N, R = core->ctx.R;
    if (cap < N) return fail(core, KAI_ERR_CAPACITY, "kai_node_states: cap < n_nodes");
    HIP_TRY(core, hipSetDevice(core->device));
    std::vector<double> idle((size_t)R * N + 1), rel((size_t)R * N + 1), used((size_t)R * N + 1);
    if (N) {
        HIP_TRY(core, hipMemcpyAsync(idle.data(), KAI_VP(core->ctx.n_idle), (size_t)R * N * 8, hipMemcpyDeviceToHost, core->stream));
        HIP_TRY(core, hipMemcpyAsync(rel.data(), KAI_VP(core->ctx.n_rel), (size_t)R * N * 8, hipMemcpyDeviceToHost, core->stream));
        HIP_TRY(core, hipMemcpyAsync(used.data(), KAI_VP(core->ctx.n_used), (size_t)R * N * 8, hipMemcpyDeviceToHost, core->stream));
    }
    HIP_TRY(core, hipStreamSynchronize(core->stream));
    for (int i = 0; i < N; i++) {
        kai_node_state& o = out[core->perm[i]];
        std::memset(&o, 0, sizeof(kai_node_state));
        for (int r = 0; r < R; r++) { o.idle[r] = idle[(size_t)r * N + i]; o.releasing[r] = rel[(size_t)r * N + i]; o.used[r] = used[(size_t)r * N + i]; }
    }
    return KAI_OK;
}

int kai_shard_attach(kai_core* core, int rank, int world, int offers_per_class, kai_allgather_fn fn, void* user) {
    if (!core || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) return KAI_ERR_INVALID_ARG;
    if (core->open) return fail(core, KAI_ERR_STATE, "kai_shard_attach: before kai_session_open");
    core->world = world; core->rank = rank; core->shard_k = offers_per_class; core->ag_fn = fn; core->ag_user = user;
    return KAI_OK;
}

int kai_shard_attach_host(kai_core* core, kai_allgather_fn fn, void* user) {
    if (!core) return KAI_ERR_INVALID_ARG;
    core->xag_fn = fn; core->xag_user = user;
    return KAI_OK;
}

int kai_shard_rccl_id(kai_core* core, void* id_out) {
    if (!core || !id_out) return KAI_ERR_INVALID_ARG;
    RcclApi* a = rccl_api();
    if (!a) return fail(core, KAI_ERR_COMM, "kai_shard_rccl_id: librccl could not be loaded");
    RcclId id; const int rc = a->GetUniqueId(&id);
    if (rc) return rccl_fail(core, "ncclGetUniqueId", rc);
    std::memcpy(id_out, id.internal, sizeof id.internal);
    return KAI_OK;
}

int kai_shard_attach_rccl(kai_core* core, int rank, int world, int offers_per_class, const void* id) {
    if (!core || !id || world < 1 || rank < 0 || rank >= world) return KAI_ERR_INVALID_ARG;
    if (core->open) return fail(core, KAI_ERR_STATE, "kai_shard_attach_rccl: before kai_session_open");
    RcclApi* a = rccl_api();
    if (!a) return fail(core, KAI_ERR_COMM, "kai_shard_attach_rccl: librccl could not be loaded");
    HIP_TRY(core, hipSetDevice(core->device));
    if (core->rccl_comm) { (void)a->CommDestroy(core->rccl_comm); core->rccl_comm = nullptr; }
    RcclId uid; std::memcpy(uid.internal, id, sizeof uid.internal);
    void* comm = nullptr; const int rc = a->CommInitRank(&comm, world, uid, rank);
    if (rc) return rccl_fail(core, "ncclCommInitRank", rc);
    core->rccl_comm = comm; core->world = world; core->rank = rank; core->shard_k = offers_per_class; core->ag_fn = nullptr; core->ag_user = nullptr;
    return KAI_OK;
}

int kai_shard_allgather_probe(kai_core* core, const void* send, void* recv, int64_t bytes_per_rank) {
    if (!core || !send || !recv || bytes_per_rank <= 0) return KAI_ERR_INVALID_ARG;
    HIP_TRY(core, hipSetDevice(core->device));
    DevLauncher dl{core};
    if (int rc = dl.allgather(send, recv, bytes_per_rank)) return rc;
    HIP_TRY(core, hipStreamSynchronize(core->stream));
    return KAI_OK;
}

int kai_pod_gpu_groups(kai_core* core, int32_t* out, int cap) {
    if (!core || !out) return KAI_ERR_INVALID_ARG;
    if (!core->open) return fail(core, KAI_ERR_STATE, "no open session");
    const int P = core->ctx.P;
    if (cap < P) return fail(core, KAI_ERR_CAPACITY, "kai_pod_gpu_groups: cap < n_pods");
    HIP_TRY(core, hipSetDevice(core->device));
    std::vector<int32_t> st((size_t)std::max(P, 1)); std::vector<uint8_t> por((size_t)std::max(P, 1));
    if (P) { HIP_TRY(core, hipMemcpyAsync(out, KAI_VP(core->ctx.p_group), (size_t)P * 4, hipMemcpyDeviceToHost, core->stream));
             HIP_TRY(core, hipMemcpyAsync(st.data(), KAI_VP(core->ctx.p_status), (size_t)P * 4, hipMemcpyDeviceToHost, core->stream));
             HIP_TRY(core, hipMemcpyAsync(por.data(), KAI_VP(core->ctx.p_shared), (size_t)P, hipMemcpyDeviceToHost, core->stream)); }
    HIP_TRY(core, hipStreamSynchronize(core->stream));
    for (int p = 0; p < P; p++) if (!(core->shared && por[p] && (st[p] & (KAI_POD_ALLOCATED | KAI_POD_PIPELINED | KAI_POD_BINDING | KAI_POD_BOUND | KAI_POD_RUNNING | KAI_POD_RELEASING)))) out[p] = -1;
    return KAI_OK;
}

int kai_action_stats_get(kai_core* core, kai_action_stats* out) {
    if (!core || !out) return KAI_ERR_INVALID_ARG;
    *out = core->stats;
    return KAI_OK;
}

}  // extern "C"
