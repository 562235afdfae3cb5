// micro-benchmark: single-wave memory latencies on one CU (what the action kernel's control lane / service waves see)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k(const int* __restrict__ chain, const double* arr, double* wr, long long* out, int n_chain, int stride) {
    __shared__ int lds[4096];
    int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += blockDim.x) lds[i] = (i * 17 + 1) & 4095;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    long long t0, t1; int idx = 0; double acc = 0;
    // (0) dependent chain of global loads, lane 0 only (pointer chasing through a 64 MB array → L2/HBM mix)
    if (lane == 0) { t0 = clock64(); for (int i = 0; i < n_chain; i++) idx = chain[idx]; t1 = clock64(); out[0] = (t1 - t0) / n_chain; out[20] = idx; }
    // (1) the same chain a second time (now L2-resident part)
    if (lane == 0) { idx = 0; t0 = clock64(); for (int i = 0; i < n_chain; i++) idx = chain[idx]; t1 = clock64(); out[1] = (t1 - t0) / n_chain; out[21] = idx; }
    // (2) 13 independent coalesced loads by the whole wave (like load_node), cold then warm
    for (int rep = 0; rep < 2; rep++) {
        t0 = clock64();
        double v[13];
        #pragma unroll
        for (int r = 0; r < 13; r++) v[r] = arr[(size_t)r * stride + lane + 64 * 7];
        #pragma unroll
        for (int r = 0; r < 13; r++) acc += v[r];
        t1 = clock64() + (acc == 1.2345 ? 1 : 0);
        if (lane == 0) out[2 + rep] = t1 - t0;
    }
    // (4) LDS dependent chain, lane 0
    if (lane == 0) { idx = 0; t0 = clock64(); for (int i = 0; i < 256; i++) idx = lds[idx]; t1 = clock64(); out[4] = (t1 - t0) / 256; out[22] = idx; }
    // (5) store then load of the same address by lane 0 (RMW through memory), 64 times
    if (lane == 0) { t0 = clock64(); for (int i = 0; i < 64; i++) { double x = wr[i * 37]; wr[i * 37] = x + 1.0; } t1 = clock64(); out[5] = (t1 - t0) / 64; }
    // (6) 16 stores then a fence (what __syncthreads waits for)
    t0 = clock64();
    if (lane == 0) for (int i = 0; i < 16; i++) wr[4096 + i * 64] = (double)i;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    t1 = clock64();
    if (lane == 0) out[6] = t1 - t0;
    // (7) single warm load latency by lane 0 of an address another lane wrote a while ago
    if (lane == 0) { t0 = clock64(); double x = wr[4096]; t1 = clock64() + (x == 1.2345 ? 1 : 0); out[7] = t1 - t0; }
    // (8) load right after a store to a DIFFERENT address (does the load wait for the store?)
    if (lane == 0) { t0 = clock64(); wr[8192] = 3.0; double x = arr[12345]; t1 = clock64() + (x == 1.2345 ? 1 : 0); out[8] = t1 - t0; }
    if (lane == 0) out[23] = (long long)acc;
}
int main() {
    const int NC = 16 << 20;  // 64 MB of ints
    std::vector<int> h(NC); unsigned x = 12345; for (int i = 0; i < NC; i++) { x = x * 1664525u + 1013904223u; h[i] = (int)((x >> 4) % NC); }
    int* dc; double *da, *dw; long long* dout;
    CK(hipMalloc(&dc, (size_t)NC * 4)); CK(hipMalloc(&da, (size_t)13 * 65536 * 8)); CK(hipMalloc(&dw, 1 << 20)); CK(hipMalloc(&dout, 64 * 8));
    CK(hipMemcpy(dc, h.data(), (size_t)NC * 4, hipMemcpyHostToDevice)); CK(hipMemset(da, 0, (size_t)13 * 65536 * 8)); CK(hipMemset(dw, 0, 1 << 20)); CK(hipMemset(dout, 0, 64 * 8));
    for (int threads : {64, 512}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, dc, da, dw, dout, 2000, 65536); CK(hipDeviceSynchronize());
        long long o[64]; CK(hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost));
        printf("threads=%d: chain cold %lld  chain again %lld | 13 loads cold %lld warm %lld | lds chain %lld | RMW %lld | 16 stores+fence %lld | warm load %lld | store+load %lld\n",
               threads, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8]);
    }
    return 0;
}
