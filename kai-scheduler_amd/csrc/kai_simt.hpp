// kai_simt.hpp — the few SIMT primitives the batch-path kernels (kai_plan_kernels.hpp) are written against.
//
// Device pass (hipcc, gfx950): thin wrappers over the hardware — threadIdx / blockIdx, s_barrier, ballot, ds_bpermute shuffles, the DPP
// max-scan of kai_wave.hpp, global atomics.  That is the product.
//
// Plain g++ (tests/host_sim only): the same names are backed by a small lock-step emulator — every thread of a workgroup is a fiber with its
// own stack, a collective (barrier, ballot, shuffle) parks the fiber until all live lanes of its wave / workgroup have arrived — so that the
// kernels' bodies, scans and index arithmetic can be debugged against the oracle in the `-m "not gpu"` suite of a container without a GPU.
// It is TEST INFRASTRUCTURE: libkai_core.so never contains it (hipcc defines __HIPCC__ in both of its passes) and nothing in the package
// loads it; every parity claim is made by the `-m gpu` tests through the C ABI on a real MI355X.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>

#include "kai_wave.hpp"
#define KW_DEV __device__ __forceinline__
#define KW_BODY __device__ __forceinline__  // kernel bodies and their helpers: device code only in the product library
#define KW_SHARED __shared__
#define KW_LDS_PTR(T) T __attribute__((address_space(3)))*  // pointer into LDS: ds_read / ds_write instead of flat accesses
namespace kw {
KW_DEV int tid() { return (int)threadIdx.x; }
KW_DEV int bid() { return (int)blockIdx.x; }
KW_DEV int bdim() { return (int)blockDim.x; }
KW_DEV int gdim() { return (int)gridDim.x; }
KW_DEV int lane() { return (int)(threadIdx.x & 63); }
KW_DEV void sync() { __syncthreads(); }
KW_DEV uint64_t ballot(bool p) { return (uint64_t)__ballot(p); }
template <class T> KW_DEV T shfl(T v, int src) { return __shfl(v, src, 64); }
KW_DEV uint64_t shfl(uint64_t v, int src) { return (uint64_t)__shfl((unsigned long long)v, src, 64); }
KW_DEV int64_t shfl(int64_t v, int src) { return (int64_t)__shfl((long long)v, src, 64); }
template <class T> KW_DEV T shfl_up(T v, int d) { return __shfl_up(v, d, 64); }  // lanes < d keep their own value
KW_DEV uint64_t wave_max_u64(uint64_t v) { return (uint64_t)kai::wave_max_u64((unsigned long long)v); }
// value of lane `src` in every lane, `src` being the same in all lanes: v_readlane_b32 per dword (no LDS round trip)
KW_DEV int bcast(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
KW_DEV uint32_t bcast(uint32_t v, int src) { return (uint32_t)bcast((int)v, src); }
KW_DEV uint64_t bcast(uint64_t v, int src) { const int s = __builtin_amdgcn_readfirstlane(src); return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), s) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, s); }
KW_DEV double bcast(double v, int src) { return __longlong_as_double((long long)bcast((uint64_t)__double_as_longlong(v), src)); }
KW_DEV int atomic_add(int32_t* p, int v) { return atomicAdd(p, v); }
KW_DEV int atomic_min(int32_t* p, int v) { return atomicMin(p, v); }
KW_DEV int atomic_max(int32_t* p, int v) { return atomicMax(p, v); }
KW_DEV unsigned long long atomic_add(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
KW_DEV unsigned long long atomic_or(unsigned long long* p, unsigned long long v) { return atomicOr(p, v); }
KW_DEV double atomic_add(double* p, double v) { return atomicAdd(p, v); }
KW_DEV uint32_t atomic_or32(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
KW_DEV int64_t clock() { return (int64_t)clock64(); }
KW_DEV unsigned char* dyn_lds() { extern __shared__ __align__(16) unsigned char kw_dyn_lds_[]; return kw_dyn_lds_; }
KW_DEV void fence() { __threadfence(); }
// lanes of one wave exchange data through LDS / memory: on the hardware they run in lock-step, so ordering the accesses is enough
KW_DEV void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
// lanes of ONE wave exchange data through LDS only: the LDS serves a wave's accesses in issue order, so keeping the compiler from reordering is enough
// (no s_waitcnt on the vector-memory counter: stores to HBM stay in flight)
KW_DEV void lds_order() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
// read-modify-write of one LDS word in ONE ds instruction (ds_xor_rtn_b64); returns the old value.  Wavefront scope: the word belongs to the calling lane
KW_DEV uint64_t lds_xor(KW_LDS_PTR(uint64_t) p, uint64_t v) { return __hip_atomic_fetch_xor(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
KW_DEV void expect_uniform(long long) {}
KW_DEV void fence_wg() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
// two wavefronts of one workgroup hand data over through LDS (kai_fill_counts.hpp: a command ring): the producer's stores, then its release store of the counter; the consumer's
// acquire load of the counter, then its loads.  relax(): the polling wavefront steps aside for a few cycles
KW_DEV int lds_load_acq(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
KW_DEV void lds_store_rel(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
KW_DEV void relax() { __builtin_amdgcn_s_sleep(1); }
// ... the same hand-over without waiting for the producer's own stores to land: the LDS executes one wavefront's instructions in issue order, so a payload stored before the counter is
// in place before it; all the producer must do is keep the compiler from reordering the two stores (kai_fill_levels.hpp: a worker's hand-over entries)
KW_DEV void lds_store_ordered(int32_t* p, int v) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
}  // namespace kw

#else  // ---------------------------------------------------------------------------------------------- host emulator (tests only)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#define KW_DEV inline
#define KW_BODY inline
#define KW_LDS_PTR(T) T*
#define KW_SHARED static  // workgroups run one after the other, so one static object per declaration is "the LDS of the running workgroup"
namespace kw {

extern "C" void kw_switch_ctx(void** save_sp, void* load_sp);
#if defined(__x86_64__)
asm(R"(
.text
.globl kw_switch_ctx
.type kw_switch_ctx,@function
kw_switch_ctx:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size kw_switch_ctx,.-kw_switch_ctx
)");
#else
#error "the host emulator's context switch is written for x86-64"
#endif

enum FiberState : int { F_RUN = 0, F_WAIT_BLOCK = 1, F_WAIT_WAVE = 2, F_DONE = 3 };
struct Fiber { void* sp = nullptr; int state = F_DONE; char* stack = nullptr; int line = 0; };  // line: source line of the collective the fiber is parked in (deadlock report)
struct Emu {
    std::vector<Fiber> fibers; std::vector<char*> stacks;
    void* sched_sp = nullptr;
    int cur = -1, block = 0, bdim = 0, gdim = 0;
    const std::function<void()>* body = nullptr;
    uint64_t xchg[64 * 64];       // per wave: 64 slots of 8 bytes (up to 64 waves per workgroup)
    std::vector<unsigned char> lds;
    static constexpr size_t STACK = 128 * 1024;
};
inline Emu& emu() { static Emu e; return e; }
inline void fiber_yield() { Emu& e = emu(); kw_switch_ctx(&e.fibers[e.cur].sp, e.sched_sp); }
inline void fiber_entry() {
    Emu& e = emu();
    (*e.body)();
    e.fibers[e.cur].state = F_DONE;
    fiber_yield();
    std::abort();  // never resumed
}
// Runs `grid` workgroups of `block` threads, one workgroup at a time, its threads as fibers in lock-step at collectives.
inline void launch(int grid, int block, size_t dyn_lds_bytes, const std::function<void()>& body) {
    Emu& e = emu();
    if (grid <= 0 || block <= 0) return;
    if (block > 64 * 64) { std::fprintf(stderr, "kw::launch: workgroup too large\n"); std::abort(); }
    e.lds.assign(dyn_lds_bytes + 64, 0);
    e.body = &body; e.bdim = block; e.gdim = grid;
    if ((int)e.fibers.size() < block) e.fibers.resize(block);
    while ((int)e.stacks.size() < block) e.stacks.push_back(static_cast<char*>(std::malloc(Emu::STACK)));
    for (int b = 0; b < grid; b++) {
        e.block = b;
        for (int t = 0; t < block; t++) {
            Fiber& f = e.fibers[t]; f.stack = e.stacks[t]; f.state = F_RUN;
            // initial frame: six callee-saved registers, then the return address = fiber_entry; at its first instruction rsp % 16 == 8
            uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + Emu::STACK) & ~uintptr_t(15);
            void** sp = reinterpret_cast<void**>(top) - 2;  // keeps the alignment rule after `ret`
            *sp = reinterpret_cast<void*>(&fiber_entry);
            for (int i = 0; i < 6; i++) *--sp = nullptr;
            f.sp = sp;
        }
        int live = block;
        // KW_EMU_ORDER: how the waves of a workgroup take turns between their collectives — unset / 0: in order; 1: in reverse; 2: every wave sits
        // out a pass with probability 1/2 (seeded), so the waves drift apart as far as the barriers allow.  Races between waves show up as
        // results that depend on this setting.
        static const int order = [] { const char* v = std::getenv("KW_EMU_ORDER"); return v ? std::atoi(v) : 0; }();
        static uint64_t rng = [] { const char* v = std::getenv("KW_EMU_SEED"); return 0x9e3779b97f4a7c15ull + (v ? (uint64_t)std::atoll(v) * 0x100000001b3ull : 0); }();
        while (live > 0) {
            bool progress = false;
            const int nwv = (block + 63) / 64;
            for (int wi = 0; wi < nwv; wi++) {
                const int w = order == 1 ? nwv - 1 - wi : wi;
                if (order == 2) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; if ((rng >> 40) & 1) continue; }
                for (int t = w * 64; t < block && t < w * 64 + 64; t++) {
                    if (e.fibers[t].state != F_RUN) continue;
                    e.cur = t; kw_switch_ctx(&e.sched_sp, e.fibers[t].sp);
                    progress = true;
                    if (e.fibers[t].state == F_DONE) live--;
                }
            }
            if (order == 2 && !progress) { bool any = false; for (int t = 0; t < block; t++) if (e.fibers[t].state == F_RUN) any = true; if (any) continue; }  // every wave sat out
            // release the barriers every live participant has reached
            bool all_block = true; int waiting = 0;
            for (int t = 0; t < block; t++) { int s = e.fibers[t].state; if (s == F_RUN || s == F_WAIT_WAVE) all_block = false; if (s == F_WAIT_BLOCK) waiting++; }
            if (all_block && waiting) { for (int t = 0; t < block; t++) if (e.fibers[t].state == F_WAIT_BLOCK) e.fibers[t].state = F_RUN; progress = true; }
            for (int w = 0; w * 64 < block; w++) {
                bool all = true; int nw = 0;
                for (int t = w * 64; t < block && t < w * 64 + 64; t++) { int s = e.fibers[t].state; if (s == F_RUN || s == F_WAIT_BLOCK) all = false; if (s == F_WAIT_WAVE) nw++; }
                if (all && nw) { for (int t = w * 64; t < block && t < w * 64 + 64; t++) if (e.fibers[t].state == F_WAIT_WAVE) e.fibers[t].state = F_RUN; progress = true; }
            }
            if (!progress && live > 0) {
                std::fprintf(stderr, "kw::launch: deadlock (a collective not reached by every live lane); block %d, lanes parked at source lines:", b);
                for (int t = 0; t < block && t < 64; t++) std::fprintf(stderr, " %d:%s%d", t, e.fibers[t].state == F_WAIT_BLOCK ? "B" : e.fibers[t].state == F_WAIT_WAVE ? "W" : e.fibers[t].state == F_DONE ? "done" : "run", e.fibers[t].line);
                std::fprintf(stderr, "\n"); std::abort();
            }
        }
    }
    e.body = nullptr;
}
inline int tid() { return emu().cur; }
inline int bid() { return emu().block; }
inline int bdim() { return emu().bdim; }
inline int gdim() { return emu().gdim; }
inline int lane() { return emu().cur & 63; }
inline void sync(int line = __builtin_LINE()) { Emu& e = emu(); e.fibers[e.cur].line = line; e.fibers[e.cur].state = F_WAIT_BLOCK; fiber_yield(); }
inline void wave_bar(int line = 0) { Emu& e = emu(); if (line) e.fibers[e.cur].line = line; e.fibers[e.cur].state = F_WAIT_WAVE; fiber_yield(); }
inline uint64_t* wave_slots() { Emu& e = emu(); return e.xchg + (size_t)(e.cur >> 6) * 64; }
inline bool lane_live(int l) { Emu& e = emu(); int t = (e.cur & ~63) + l; return t < e.bdim && e.fibers[t].state != F_DONE; }
template <class T> inline T shfl(T v, int src, int line = __builtin_LINE()) {
    static_assert(sizeof(T) <= 8, "shfl: 8 bytes at most");
    uint64_t raw = 0; std::memcpy(&raw, &v, sizeof(T));
    wave_slots()[lane()] = raw; wave_bar(line);
    uint64_t got = wave_slots()[src & 63]; wave_bar();
    T r; std::memcpy(&r, &got, sizeof(T)); return r;
}
template <class T> inline T shfl_up(T v, int d) { int l = lane(); T r = shfl(v, l >= d ? l - d : l); return l >= d ? r : v; }
inline uint64_t ballot(bool p, int line = __builtin_LINE()) {
    wave_slots()[lane()] = p ? 1 : 0; wave_bar(line);
    uint64_t m = 0; for (int l = 0; l < 64; l++) if (lane_live(l) && wave_slots()[l]) m |= 1ull << l;
    wave_bar(); return m;
}
inline uint64_t wave_max_u64(uint64_t v, int line = __builtin_LINE()) {
    wave_slots()[lane()] = v; wave_bar(line);
    uint64_t m = 0; for (int l = 0; l < 64; l++) if (lane_live(l) && wave_slots()[l] > m) m = wave_slots()[l];
    wave_bar(); return m;
}
template <class T> inline T bcast(T v, int src, int line = __builtin_LINE()) { return shfl(v, src, line); }
inline int atomic_add(int32_t* p, int v) { int o = *p; *p = o + v; return o; }
inline int atomic_min(int32_t* p, int v) { int o = *p; if (v < o) *p = v; return o; }
inline int atomic_max(int32_t* p, int v) { int o = *p; if (v > o) *p = v; return o; }
inline unsigned long long atomic_add(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline unsigned long long atomic_or(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o | v; return o; }
inline double atomic_add(double* p, double v) { double o = *p; *p = o + v; return o; }
inline uint32_t atomic_or32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
inline int64_t clock() { return 0; }
inline unsigned char* dyn_lds() { Emu& e = emu(); return reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(e.lds.data()) + 15) & ~uintptr_t(15)); }
inline void fence() {}
inline void fence_wg() {}
inline int lds_load_acq(const int32_t* p) { return *p; }
inline void lds_store_rel(int32_t* p, int v) { *p = v; }
inline void lds_store_ordered(int32_t* p, int v) { *p = v; }
inline void relax(int line = __builtin_LINE()) { wave_bar(line); }  // the whole wave parks: the scheduler lets the other wavefronts of the workgroup run
inline void wave_sync(int line = __builtin_LINE()) { wave_bar(line); }
inline void lds_order(int line = __builtin_LINE()) { wave_bar(line); }
inline uint64_t lds_xor(uint64_t* p, uint64_t v) { const uint64_t o = *p; *p = o ^ v; return o; }
// debug aid: every lane must hold the same value here (a branch on it is meant to be uniform)
inline void expect_uniform(long long v, int line = __builtin_LINE()) { const long long v0 = shfl(v, 0, line); if (v != v0) { std::fprintf(stderr, "kw: value not uniform at line %d: lane %d has %lld, lane 0 has %lld\n", line, lane(), v, v0); std::abort(); } }  // the emulator's lanes are fibers: they meet here
}  // namespace kw
#endif

namespace kw {
// arg-max of (key, n) over the wave, ties to the lowest lane; every lane returns the winner (key 0 = none).  Same contract as
// kai::wave_argmax_first (kai_wave.hpp), which it is on the device.
KW_DEV void wave_argmax_first(uint64_t& key, int& n) {
    const uint64_t m = wave_max_u64(key);
    const uint64_t win = ballot(key == m);
    const int l = __builtin_ctzll(win);
    n = shfl(n, l);
    key = m;
}
// inclusive prefix sums over the 64 lanes (Hillis-Steele with shfl_up)
template <class T> KW_DEV T wave_scan_add(T v) {
    for (int d = 1; d < 64; d <<= 1) { T o = shfl_up(v, d); if (lane() >= d) v = v + o; }
    return v;
}
#if defined(__HIPCC__)
// ... on the device with DPP instead of ds_bpermute round trips (six dependent LDS-crossbar trips per scan, ~100 cycles each, against six DPP moves): row_shr 1 / 2 / 4 / 8 inside
// the rows of 16 lanes, then row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3 — the scan of kai_wave.hpp's wave_max_u64.  A lane without a source adds 0.  The
// additions are grouped differently from the shuffle form: integers always, and the f64 sums of the plan kernels are exact in any order (HostPrep::batch_units).
#define KW_DPP_MOV(x, ctrl, rmask) __builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rmask, 0xf, false)
template <> KW_DEV int wave_scan_add<int>(int v) {
    v += KW_DPP_MOV(v, 0x111, 0xf); v += KW_DPP_MOV(v, 0x112, 0xf); v += KW_DPP_MOV(v, 0x114, 0xf); v += KW_DPP_MOV(v, 0x118, 0xf);
    v += KW_DPP_MOV(v, 0x142, 0xa); v += KW_DPP_MOV(v, 0x143, 0xc);
    return v;
}
// the DPP move of one scan step for 32- and 64-bit values (CTRL / RMASK are instruction immediates); a lane without a source, or outside the row mask, reads 0
template <int CTRL, int RMASK> KW_DEV int dpp_mov32(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, RMASK, 0xf, false); }
template <int CTRL, int RMASK> KW_DEV uint64_t dpp_mov64(uint64_t x) { return ((uint64_t)(unsigned)dpp_mov32<CTRL, RMASK>((int)(x >> 32)) << 32) | (unsigned)dpp_mov32<CTRL, RMASK>((int)x); }
template <> KW_DEV double wave_scan_add<double>(double v) {
#define KW_DPP_ADD_F64(ctrl, rmask) { const long long b_ = __double_as_longlong(v); const unsigned lo_ = (unsigned)KW_DPP_MOV((unsigned)b_, ctrl, rmask), hi_ = (unsigned)KW_DPP_MOV((unsigned)((unsigned long long)b_ >> 32), ctrl, rmask); v = v + __longlong_as_double((long long)(((unsigned long long)hi_ << 32) | lo_)); }
    KW_DPP_ADD_F64(0x111, 0xf) KW_DPP_ADD_F64(0x112, 0xf) KW_DPP_ADD_F64(0x114, 0xf) KW_DPP_ADD_F64(0x118, 0xf) KW_DPP_ADD_F64(0x142, 0xa) KW_DPP_ADD_F64(0x143, 0xc)
#undef KW_DPP_ADD_F64
    return v;
}
// one step of an inclusive scan over 64-bit words: the word of the lane `ctrl` names (0 for a lane without a source)
#endif
}  // namespace kw
