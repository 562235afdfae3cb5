// kai_parallel.hpp — the host side's one parallel primitive: a range cut into contiguous chunks, one thread per chunk, chunk i handed to f(i, begin, end).
// kai_session_open's host preparation (kai_host_prep.hpp) is loops over 10^6 pods and 10^5 jobs; every result that depends on the order of the input is
// combined over the chunks in chunk order, so the output does not depend on the number of threads (KAI_HOST_THREADS, default: the machine's cores, at most 16).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <exception>
#include <memory>
#include <new>
#include <thread>
#include <vector>

namespace kai {

inline int host_threads() {
    static const int n = [] {
        if (const char* e = std::getenv("KAI_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1) return std::min(v, 64); }
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(hc ? hc : 1u, 16u));
    }();
    return n;
}
// number of chunks parallel_chunks will use for a range of n items (>= 1)
inline int chunk_count(size_t n, size_t min_chunk = 16384) {
    const size_t want = n / std::max<size_t>(min_chunk, 1);
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), want));
}
template <class F>
inline void parallel_chunks(size_t n, F&& f, size_t min_chunk = 16384) {
    const int k = chunk_count(n, min_chunk);
    if (k <= 1) { f(0, (size_t)0, n); return; }
    std::vector<std::thread> th; th.reserve((size_t)k - 1);
    auto bound = [&](int i) { return (size_t)((unsigned __int128)n * (unsigned)i / (unsigned)k); };
    // an exception in a worker (std::bad_alloc) would end the process there: it is carried to the calling thread and rethrown after the join
    std::vector<std::exception_ptr> err((size_t)k);
    auto run = [&](int i) { try { f(i, bound(i), bound(i + 1)); } catch (...) { err[(size_t)i] = std::current_exception(); } };
    for (int i = 1; i < k; i++) th.emplace_back([&, i] { run(i); });
    run(0);
    for (auto& t : th) t.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
}

// std::vector whose resize() leaves new elements uninitialised (trivial types only): the parallel loop that follows writes every element, so the
// zero fill of a plain vector — a serial pass of page faults over tens of megabytes — is not paid first
template <class T> struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <class T> using raw_vector = std::vector<T, DefaultInitAlloc<T>>;

}  // namespace kai
