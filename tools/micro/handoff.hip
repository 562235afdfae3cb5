// micro-benchmark: the action kernel's hand-off pattern.  Wave 0 lane 0 ("control") stores a node's state, barrier,
// wave 1 ("service") loads 13 columns of the node's 64-node block, reduces, writes a result to LDS, barrier, control reads it.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ctx { int N, R; double *idle, *rel, *alloc; unsigned* flags; };
template <int MODE>  // 0: pointers as kernel args (global_load)  1: pointers from a struct in memory (flat_load)  2: as 1 but hoisted before the loop
__global__ void k(const Ctx* cp, Ctx cv, long long* out, int iters) {
    __shared__ double res[16]; __shared__ int cmd;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long t_ctl = 0, t_svc = 0, t_ld = 0;
    const Ctx& c = MODE == 0 ? cv : *cp;
    const int N0 = c.N; double* const idle0 = c.idle; double* const rel0 = c.rel; double* const alloc0 = c.alloc; unsigned* const fl0 = c.flags;
    if (threadIdx.x == 0) cmd = 0;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
        int n = (it * 977) % (N0 - 64);
        if (threadIdx.x == 0) {  // control: RMW the node's state (like node_apply), then hand over
            long long t0 = clock64();
            for (int r = 0; r < 4; r++) { double* p = (MODE == 2 ? idle0 : c.idle) + (size_t)r * N0 + n; *p = *p - 1.0; }
            cmd = n;
            __syncthreads();
            __syncthreads();
            double x = res[1];
            t_ctl += clock64() - t0 + (x == 1.25 ? 1 : 0);
        } else if (threadIdx.x >= 64) {
            __syncthreads();
            long long s0 = clock64();
            int b = cmd & ~63, nn = b + lane;
            double acc = 0;
            if (MODE == 2) {
                #pragma unroll
                for (int r = 0; r < 4; r++) acc += idle0[(size_t)r * N0 + nn] + rel0[(size_t)r * N0 + nn];
                acc += alloc0[nn] + alloc0[(size_t)2 * N0 + nn] + (double)fl0[nn];
            } else {
                #pragma unroll
                for (int r = 0; r < 4; r++) acc += c.idle[(size_t)r * c.N + nn] + c.rel[(size_t)r * c.N + nn];
                acc += c.alloc[nn] + c.alloc[(size_t)2 * c.N + nn] + (double)c.flags[nn];
            }
            long long s1 = clock64() + (acc == 1.25 ? 1 : 0);
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (lane == 0) res[wave] = acc;
            if (threadIdx.x == 64) { t_ld += s1 - s0; t_svc += clock64() - s0; }
            __syncthreads();
        } else { __syncthreads(); __syncthreads(); }
    }
    if (threadIdx.x == 0) out[0] = t_ctl / iters;
    if (threadIdx.x == 64) { out[1] = t_svc / iters; out[2] = t_ld / iters; }
}
int main() {
    const int N = 65536; Ctx h; h.N = N; h.R = 4;
    CK(hipMalloc(&h.idle, (size_t)4 * N * 8)); CK(hipMalloc(&h.rel, (size_t)4 * N * 8)); CK(hipMalloc(&h.alloc, (size_t)4 * N * 8)); CK(hipMalloc(&h.flags, (size_t)N * 4));
    CK(hipMemset(h.idle, 0, (size_t)4 * N * 8)); CK(hipMemset(h.rel, 0, (size_t)4 * N * 8)); CK(hipMemset(h.alloc, 0, (size_t)4 * N * 8)); CK(hipMemset(h.flags, 0, (size_t)N * 4));
    Ctx* d; long long* dout; CK(hipMalloc(&d, sizeof(Ctx))); CK(hipMemcpy(d, &h, sizeof(Ctx), hipMemcpyHostToDevice)); CK(hipMalloc(&dout, 64));
    for (int threads : {128, 512}) for (int mode = 0; mode < 3; mode++) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, d, h, dout, 2000);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, d, h, dout, 2000);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, d, h, dout, 2000);
        CK(hipDeviceSynchronize());
        long long o[8]; CK(hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost));
        printf("threads=%d mode=%d: control round trip %lld cycles | service busy %lld (loads %lld)\n", threads, mode, o[0], o[1], o[2]);
    }
    return 0;
}
