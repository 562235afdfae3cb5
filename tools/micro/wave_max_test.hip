// micro-test: DPP wave-wide max of u64 + first-lane arg-max vs the compiler's scalar-loop builtin (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../kai-scheduler_amd/csrc/kai_wave.hpp"
__global__ void k(const unsigned long long* in, unsigned long long* out, int* outl, long long* cyc) {
    unsigned long long v = in[blockIdx.x * 64 + threadIdx.x];
    long long t0 = clock64();
    unsigned long long m = kai::wave_max_u64(v);
    long long t1 = clock64();
    unsigned long long r = __builtin_amdgcn_wave_reduce_max_u64(v, 0);
    long long t2 = clock64();
    int n = (int)threadIdx.x + 1000; unsigned long long kk = v; kai::wave_argmax_first(kk, n);
    long long t3 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = m; out[blockIdx.x * 3 + 1] = r; out[blockIdx.x * 3 + 2] = kk; outl[blockIdx.x] = n; cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main() {
    const int B = 4096; unsigned long long *h = (unsigned long long*)malloc(B * 64 * 8), *d, *o; int* ol; long long* cy;
    srand(1);
    for (int i = 0; i < B * 64; i++) { unsigned long long x = ((unsigned long long)rand() << 40) ^ ((unsigned long long)rand() << 20) ^ rand(); int m = i / 64 % 4; h[i] = m == 0 ? x : m == 1 ? (x & 7) : m == 2 ? (x | (1ull << 63)) : (x & 0xffffffff00000000ull); }
    for (int i = 0; i < 64; i++) h[i] = 0;               // all-zero block
    for (int i = 64; i < 128; i++) h[i] = 5;             // all-equal block → first lane
    hipMalloc(&d, B * 64 * 8); hipMalloc(&o, B * 3 * 8); hipMalloc(&ol, B * 4); hipMalloc(&cy, 24);
    hipMemcpy(d, h, B * 64 * 8, hipMemcpyHostToDevice);
    k<<<B, 64>>>(d, o, ol, cy); hipDeviceSynchronize();
    unsigned long long* ho = (unsigned long long*)malloc(B * 3 * 8); int* hl = (int*)malloc(B * 4); long long hc[3];
    hipMemcpy(ho, o, B * 3 * 8, hipMemcpyDeviceToHost); hipMemcpy(hl, ol, B * 4, hipMemcpyDeviceToHost); hipMemcpy(hc, cy, 24, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < B; b++) {
        unsigned long long mx = 0; int first = 0; for (int i = 0; i < 64; i++) if (h[b * 64 + i] > mx) { mx = h[b * 64 + i]; first = i; }
        if (mx == 0) first = 0;
        if (ho[b * 3] != mx || ho[b * 3 + 1] != mx || ho[b * 3 + 2] != mx || hl[b] != first + 1000) { if (bad < 5) printf("block %d: dpp %llx builtin %llx argmax %llx lane %d want %llx %d\n", b, ho[b * 3], ho[b * 3 + 1], ho[b * 3 + 2], hl[b], mx, first + 1000); bad++; }
    }
    printf("bad=%d cycles dpp=%lld builtin=%lld argmax=%lld\n", bad, hc[0], hc[1], hc[2]);
    return bad != 0;
}
