// fill_bench.hip — replays ONE dumped fill launch (tests/host_sim, KAI_HOSTSIM_FILL_DUMP) on the MI355X through k_fill_levels / k_fill_counts / k_fill_buckets and times it with HIP events.
// A development aid for the fill kernels (the product library takes ≈ 7 min to build; this file a few seconds): the dump holds the planned order of a round, the sets before the launch
// and — from the emulated kernel, which the scalar C++ shadow and the oracle vouch for — every output, all of which are compared here.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/micro/fill_bench tools/micro/fill_bench.hip
//   tools/micro/fill_bench tools/micro/data/c5_2 [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../kai-scheduler_amd/csrc/kai_batch_kernels.hpp"

using namespace kai;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

template <class T> static T* dev(const std::vector<T>& h, size_t n = 0) { T* d = nullptr; const size_t m = std::max(n, h.size()); CK(hipMalloc(&d, std::max<size_t>(m, 1) * sizeof(T))); CK(hipMemset(d, 0, std::max<size_t>(m, 1) * sizeof(T))); if (!h.empty()) CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
template <class T> static void rd(FILE* f, std::vector<T>& v, size_t n) { v.resize(n); if (n && std::fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } }

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: fill_bench <dump prefix> [reps]\n"); return 2; }
    const std::string pre = argv[1]; const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
    FILE* f = std::fopen((pre + ".in").c_str(), "rb"); if (!f) { std::perror("in"); return 2; }
    int32_t hdr[16]; RoundParams rp; BucketParams bp;
    if (std::fread(hdr, 4, 16, f) != 16 || std::fread(&rp, sizeof rp, 1, f) != 1 || std::fread(&bp, sizeof bp, 1, f) != 1 || hdr[0] != 0x4b464c31) { std::fprintf(stderr, "bad header\n"); return 2; }
    const int C = hdr[1], Q = hdr[2], P = hdr[3], V = hdr[4], LV = hdr[5], NW = hdr[6];
    std::vector<double> qd; rd(f, qd, 64);
    std::vector<uint8_t> g_flag; std::vector<int32_t> g_first, g_nt, g_ucls, t_cls; std::vector<uint64_t> words;
    rd(f, g_flag, V); rd(f, g_first, V); rd(f, g_nt, V); rd(f, g_ucls, V); rd(f, t_cls, P); rd(f, words, (size_t)LV * NW); std::fclose(f);
    f = std::fopen((pre + ".out").c_str(), "rb"); if (!f) { std::perror("out"); return 2; }
    FillStatus efs; std::vector<uint8_t> e_out; std::vector<int32_t> e_opoff, e_stmt, e_node; std::vector<uint64_t> e_words;
    if (std::fread(&efs, sizeof efs, 1, f) != 1) return 2;
    rd(f, e_out, V); rd(f, e_opoff, V); rd(f, e_stmt, V); rd(f, e_node, P); rd(f, e_words, (size_t)LV * NW); std::fclose(f);
    std::printf("dump %s: C %d Q %d P %d planned %d (from %d) levels %d words/level %d | expected: executed %d mismatch %d decisions %lld committed %lld commands %lld\n", pre.c_str(), C, Q, P, V, rp.start, LV, NW,
                efs.n_done, efs.mismatch, (long long)efs.decisions, (long long)efs.committed, (long long)efs.rescans2);

    std::vector<ClassRec> cls(64); for (int k = 0; k < 64; k++) { std::memset(&cls[k], 0, sizeof(ClassRec)); cls[k].req[KAI_RES_GPU] = qd[k]; }
    std::vector<int32_t> q_valid(Q + 1, 0); q_valid[Q] = V;
    KaiCtx c; std::memset((void*)&c, 0, sizeof c);
    c.C = C; c.Q = Q; c.P = P; c.NB = NW;
    c.cls = (KAI_GP(const ClassRec))dev(cls);
    BatchCtx& b = c.bt;
    b.q_valid = (KAI_GP(int32_t))dev(q_valid);
    b.g_flag = (KAI_GP(uint8_t))dev(g_flag); b.g_first = (KAI_GP(int32_t))dev(g_first); b.g_nt = (KAI_GP(int32_t))dev(g_nt); b.g_ucls = (KAI_GP(int32_t))dev(g_ucls);
    b.t_cls = (KAI_GP(int32_t))dev(t_cls);
    uint64_t* d_words = dev(words, (size_t)KBK_GMAX * NW); b.bk_words = (KAI_GP(uint64_t))d_words;
    std::vector<uint8_t> z8; std::vector<int32_t> z32; std::vector<uint64_t> z64; std::vector<FillStatus> zfs(1);
    uint8_t* d_out = dev(z8, V + 64); int32_t* d_opoff = dev(z32, V + 64); int32_t* d_stmt = dev(z32, V + 64); int32_t* d_node = dev(z32, P + 64);
    b.g_out = (KAI_GP(uint8_t))d_out; b.g_opoff = (KAI_GP(int32_t))d_opoff; b.g_stmt = (KAI_GP(int32_t))d_stmt; b.t_node = (KAI_GP(int32_t))d_node;
    FillStatus* d_fs = dev(zfs); b.fs = (KAI_GP(FillStatus))d_fs; b.dead_mask = (KAI_GP(uint64_t))dev(z64, 1);

    const size_t dyn = ((size_t)LV * NW + (size_t)LV * bp.nw1 + KBK_GMAX) * 8 + 16;  // the sets and their first summaries (kai_batch_driver.hpp batch_bucket_params; a plain cluster has no class bitmaps)
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill_levels), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill_counts), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill_buckets), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"k_fill_levels", "k_fill_counts", "k_fill_buckets"};
    for (int kern = 0; kern < 3; kern++) {
        if (kern == 0 && LV > KFL_LMAX) continue;
        if (const char* only = std::getenv("FILL_BENCH_ONLY")) if (std::atoi(only) != kern) continue;
        double best = 1e30, sum = 0; FillStatus fs{};
        for (int r = 0; r < reps; r++) {
            CK(hipMemcpy(d_words, words.data(), words.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemset(d_node, 0xff, (size_t)P * 4)); CK(hipMemset(d_fs, 0, sizeof(FillStatus)));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (kern == 0) hipLaunchKernelGGL(k_fill_levels, dim3(1), dim3(64 * (LV + 2)), dyn, 0, c, rp, bp);
            else if (kern == 1) hipLaunchKernelGGL(k_fill_counts, dim3(1), dim3(256), dyn, 0, c, rp, bp);
            else hipLaunchKernelGGL(k_fill_buckets, dim3(1), dim3(256), dyn, 0, c, rp, bp);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, (double)ms); sum += ms;
        }
        CK(hipMemcpy(&fs, d_fs, sizeof fs, hipMemcpyDeviceToHost));
        std::vector<uint8_t> o(V); std::vector<int32_t> oo(V), os(V), on(P); std::vector<uint64_t> ow((size_t)LV * NW);
        CK(hipMemcpy(o.data(), d_out, V, hipMemcpyDeviceToHost)); CK(hipMemcpy(oo.data(), d_opoff, (size_t)V * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(os.data(), d_stmt, (size_t)V * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(on.data(), d_node, (size_t)P * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ow.data(), d_words, ow.size() * 8, hipMemcpyDeviceToHost));
        long long bad = 0;
        if (fs.n_done != efs.n_done || fs.mismatch != efs.mismatch || fs.decisions != efs.decisions || fs.attempted != efs.attempted || fs.committed != efs.committed || fs.rollbacks != efs.rollbacks || fs.ops != efs.ops || fs.dead_mask != efs.dead_mask || fs.all_dead != efs.all_dead) bad++;
        for (int gi = rp.start; gi < efs.n_done; gi++) {
            if (o[gi] != e_out[gi]) { bad++; continue; }
            if (o[gi] != BF_OK || g_flag[gi] == BF_GATE) continue;
            if (oo[gi] != e_opoff[gi] || os[gi] != e_stmt[gi]) bad++;
            for (int t = 0; t < g_nt[gi]; t++) if (on[g_first[gi] + t] != e_node[g_first[gi] + t]) bad++;
        }
        for (size_t i = 0; i < ow.size(); i++) if (ow[i] != e_words[i]) bad++;
        std::printf("%-15s best %8.3f ms  mean %8.3f ms  | executed %d decisions %lld committed %lld commands %lld | counting-wave cycles %lld (waited for the ring %lld) workers idle %lld / total %lld, busiest busy %lld (level %lld) | %s (%lld differences)\n",
                    names[kern], best, sum / reps, fs.n_done, (long long)fs.decisions, (long long)fs.committed, (long long)fs.rescans2, (long long)fs.cycles_total, (long long)fs.cycles_load,
                    (long long)fs.cycles_update, (long long)fs.cycles_rescan, (long long)fs.block_loads, (long long)fs.rescans1, bad ? "DIFFERS from the dump" : "outputs = the dump's", bad);
    }
    return 0;
}
