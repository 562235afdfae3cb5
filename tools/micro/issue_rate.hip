// issue_rate.hip — what one wavefront pays per instruction on gfx950, alone on its SIMD: dependent / independent scalar chains, scalar <-> vector crossings, v_readlane, taken branches.
// Development aid for the counting machine of kai_fill_levels.hpp.  hipcc --offload-arch=gfx950 -O3 -o tools/micro/issue_rate tools/micro/issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void k(long long* out, int n, int seed) {
    long long t[12]; int a = seed, b2 = seed + 1, c2 = seed + 2, d2 = seed + 3; int v = threadIdx.x + seed; unsigned long long m;
    // 0: dependent s_add chain
    t[0] = clock64();
    for (int i = 0; i < n; i++) { asm volatile(REP16("s_add_i32 %0, %0, 1\n") : "+s"(a) :: "scc"); }
    t[1] = clock64();
    // 1: independent s_adds (4 chains)
    for (int i = 0; i < n; i++) { asm volatile(REP16("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1\n") : "+s"(a), "+s"(b2), "+s"(c2), "+s"(d2) :: "scc"); }
    t[2] = clock64();
    // 2: dependent v_add chain
    for (int i = 0; i < n; i++) { asm volatile(REP16("v_add_u32 %0, %0, 1\n") : "+v"(v)); }
    t[3] = clock64();
    // 3: crossing: v_readlane (s <- v) then v_add with that s (v <- s): a dependent s/v ping-pong, 2 instructions per pair
    for (int i = 0; i < n; i++) { asm volatile(REP16("v_readlane_b32 %0, %1, 0\n s_nop 0\n v_add_u32 %1, %1, %0\n") : "+s"(a), "+v"(v) :: "scc"); }
    t[4] = clock64();
    // 4: v_cmp -> sgpr pair -> s_ff1 (dependent): VALU writes SGPR, SALU reads it, VALU reads the SALU result
    for (int i = 0; i < n; i++) { asm volatile(REP16("v_cmp_lt_i32 %2, 0, %1\n s_ff1_i32_b64 %0, %2\n v_add_u32 %1, %1, %0\n") : "+s"(a), "+v"(v), "=s"(m) :: "scc"); }
    t[5] = clock64();
    // 5: taken branches: 16 jumps over nothing
    for (int i = 0; i < n; i++) { asm volatile(REP16("s_branch 1f\n s_nop 0\n 1:\n") ::: "memory", "scc"); }
    t[6] = clock64();
    // 6: s_cmp + not-taken conditional branch + s_add
    for (int i = 0; i < n; i++) { asm volatile(REP16("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_i32 %0, %0, 1\n 1:\n") : "+s"(a) :: "scc"); }
    t[7] = clock64();
    // 7: independent mix of salu and valu (no dependencies between them)
    for (int i = 0; i < n; i++) { asm volatile(REP16("s_add_i32 %0, %0, 1\n v_add_u32 %1, %1, 1\n") : "+s"(a), "+v"(v) :: "scc"); }
    t[8] = clock64();
    // 8: ds_write fire-and-forget + s_add
    __shared__ int sh[256];
    for (int i = 0; i < n; i++) { asm volatile(REP16("ds_write_b32 %1, %2\n s_add_i32 %0, %0, 1\n") : "+s"(a) : "v"((int)(threadIdx.x * 4)), "v"(v) : "scc", "memory"); }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t[9] = clock64();
    if (threadIdx.x == 0) { for (int i = 0; i < 9; i++) out[i] = t[i + 1] - t[i]; out[9] = a + b2 + c2 + d2 + v + sh[1]; }
}
int main() {
    long long* d; hipMalloc(&d, 16 * 8); long long h[16];
    const int n = 1000;
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, 1); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 16 * 8, hipMemcpyDeviceToHost);
    const char* nm[9] = {"dependent s_add", "4 independent s_add chains (per instr)", "dependent v_add", "v_readlane + s_nop + v_add (per triple)", "v_cmp->sgpr, s_ff1, v_add (per triple)", "taken s_branch + skipped s_nop (per jump)", "s_cmp + untaken cbranch + s_add (per triple)", "s_add + v_add independent (per pair)", "ds_write + s_add (per pair)"};
    const int per[9] = {16, 64, 16, 16, 16, 16, 16, 16, 16};
    for (int i = 0; i < 9; i++) std::printf("%-50s %7.2f cycles\n", nm[i], (double)h[i] / ((double)n * per[i]));
    return 0;
}
