// fake_hip.cpp — a stand-in for the HIP runtime on a machine WITHOUT a GPU, for one purpose: running the HOST side of libkai_core.so (kai_core_create, kai_session_open: host
// preparation, device allocations, uploads, the constants kai_session_open writes on the device instead of sending them) and looking at the image it leaves in "device" memory.
// Device memory is host memory, copies and memsets are memcpy / memset at the call (one stream order), kernel launches do NOTHING — so the image is what the open put there before
// its first kernel.  tests/test_open_uploads.py builds kai_core.hip host-only (hipcc --cuda-host-only, seconds), links it with this file instead of libamdhip64 and compares the
// image of a default open with the image of KAI_OPEN_FULL_UPLOADS=1 (every array sent from the host): byte for byte the same.  Test infrastructure; never part of the product.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <sys/mman.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

namespace {
struct Block { void* p; size_t n; };
std::mutex g_m; std::vector<Block> g_live; uint64_t g_tok = 0x1000; int64_t g_launches = 0, g_h2d = 0, g_d2d = 0, g_memset = 0, g_h2d_bytes = 0, g_pinned_bytes = 0;
thread_local struct { dim3 g, b; size_t sh; hipStream_t s; } g_cfg;
}

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { std::lock_guard<std::mutex> lk(g_m); *s = (hipStream_t)(g_tok += 16); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { std::lock_guard<std::mutex> lk(g_m); *e = (hipEvent_t)(g_tok += 16); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
// Device memory comes from ONE arena at a fixed address, handed out by a bump pointer: two processes that allocate the same sizes in the same order get the same addresses, so the
// pointers the library stores IN device memory (the session context, replica tables) do not make their images differ.  (Fresh pages read as zeros: the padding between the library's
// sub-allocations is part of the image.)  hipFree gives nothing back — the test opens a handful of sessions.
hipError_t hipMalloc(void** p, size_t n) {
    std::lock_guard<std::mutex> lk(g_m);
    static char* base = nullptr; static size_t used = 0; const size_t cap = (size_t)64 << 30;
    if (!base) { void* m = mmap((void*)0x7e0000000000ull, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED_NOREPLACE, -1, 0); if (m == MAP_FAILED) return hipErrorOutOfMemory; base = (char*)m; }
    const size_t need = ((n ? n : 1) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (used + need > cap) return hipErrorOutOfMemory;
    void* m = base + used; used += need;
    g_live.push_back({m, n}); *p = m; return hipSuccess;
}
hipError_t hipFree(void* p) { std::lock_guard<std::mutex> lk(g_m); for (size_t i = 0; i < g_live.size(); i++) if (g_live[i].p == p) { g_live.erase(g_live.begin() + (long)i); break; } return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); { std::lock_guard<std::mutex> lk(g_m); g_pinned_bytes += (int64_t)n; } return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) {
    std::memmove(d, s, n);
    std::lock_guard<std::mutex> lk(g_m); if (k == hipMemcpyHostToDevice) { g_h2d++; g_h2d_bytes += (int64_t)n; } else if (k == hipMemcpyDeviceToDevice) g_d2d++;
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); std::lock_guard<std::mutex> lk(g_m); g_memset++; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fake_hip"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { std::lock_guard<std::mutex> lk(g_m); g_launches++; return hipSuccess; }
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t s) { g_cfg.g = g; g_cfg.b = b; g_cfg.sh = sh; g_cfg.s = s; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* s) { *g = g_cfg.g; *b = g_cfg.b; *sh = g_cfg.sh; *s = g_cfg.s; return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}

// the image: FNV-1a over the live device allocations in allocation order; out[9] = {hash, allocations, bytes, kernel launches, H2D copies, H2D bytes, D2D copies, memsets, pinned host bytes ever allocated}
int fakehip_image(uint64_t* out) {
    std::lock_guard<std::mutex> lk(g_m);
    uint64_t h = 1469598103934665603ull, bytes = 0;
    for (const Block& b : g_live) { const unsigned char* p = (const unsigned char*)b.p; for (size_t i = 0; i < b.n; i++) { h ^= p[i]; h *= 1099511628211ull; } bytes += b.n; h ^= b.n; h *= 1099511628211ull; }
    out[0] = h; out[1] = g_live.size(); out[2] = bytes; out[3] = (uint64_t)g_launches; out[4] = (uint64_t)g_h2d; out[5] = (uint64_t)g_h2d_bytes; out[6] = (uint64_t)g_d2d; out[7] = (uint64_t)g_memset; out[8] = (uint64_t)g_pinned_bytes;
    return 0;
}
}
