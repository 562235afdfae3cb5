// kai_parallel.hpp — the host side's one parallel primitive: a range cut into contiguous chunks, one thread per chunk, chunk i handed to f(i, begin, end).
// kai_session_open's host preparation (kai_host_prep.hpp) is loops over 10^6 pods and 10^5 jobs; every result that depends on the order of the input is
// combined over the chunks in chunk order, so the output does not depend on the number of threads (KAI_HOST_THREADS, 1 .. 64; default: the CPUs this process may run on —
// its affinity mask and its cgroup's CPU quota, not the machine's core count —, at most 16).
//
// The chunks run on a pool of worker threads that is started on first use and kept (HostPool): the preparation is ~30 such loops per session and a scheduler opens
// a session per cycle, so starting and joining 15 threads per loop (100 - 300 us each time) was a third of what was left of it.  A loop that finds the pool taken —
// two handles opening sessions on two threads of one process, a loop inside a loop — or switched off (KAI_HOST_POOL=0) starts its own threads as before.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <sched.h>
#include <unistd.h>

namespace kai {

inline int host_threads() {
    static const int n = [] {
        if (const char* e = std::getenv("KAI_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1) return std::min(v, 64); }
        // the CPUs this process may use: the affinity mask (a container pinned to 2 CPUs of a 128-core host gets 2 threads, not 15) and the cgroup's quota where one is set
        unsigned hc = std::thread::hardware_concurrency();
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a >= 1) hc = hc ? std::min<unsigned>(hc, (unsigned)a) : (unsigned)a; }
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
            long long quota = 0, period = 0;
            if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) hc = std::min<unsigned>(hc ? hc : 1u, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
            std::fclose(f);
        }
        return (int)std::max(1u, std::min(hc ? hc : 1u, 16u));
    }();
    return n;
}
// number of chunks parallel_chunks will use for a range of n items (>= 1)
inline int chunk_count(size_t n, size_t min_chunk = 16384) {
    const size_t want = n / std::max<size_t>(min_chunk, 1);
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), want));
}

// host_threads() - 1 workers waiting for a job: run(k, call, ctx) has workers 1 .. k-1 call `call(ctx, i)`, the caller runs i = 0, and returns when all are done.
// One job at a time (`busy`); the object is never destroyed (its workers are detached and wait on it until the process ends) and belongs to the process that made it:
// after a fork the child finds another pid in it and starts its own (the parent's workers do not exist there).
class HostPool {
public:
    static HostPool* get() {
        static std::atomic<HostPool*> g{nullptr};
        static const bool off = [] { const char* e = std::getenv("KAI_HOST_POOL"); return e && e[0] == '0'; }();
        if (off || host_threads() < 2) return nullptr;
        HostPool* p = g.load(std::memory_order_acquire);
        if (p && p->pid_ == getpid()) return p;
        HostPool* fresh = new (std::nothrow) HostPool();
        if (!fresh) return nullptr;
        if (!fresh->start(host_threads() - 1)) { fresh->retire(); return nullptr; }
        if (g.compare_exchange_strong(p, fresh, std::memory_order_acq_rel)) return fresh;  // (p: the pool of another process image, or none — left where it is)
        fresh->retire();  // another thread of this process installed one first
        p = g.load(std::memory_order_acquire);
        return (p && p->pid_ == getpid()) ? p : nullptr;
    }
    // false: the pool is taken (or too small for k): the caller starts its own threads
    bool run(int k, void (*call)(void*, int), void* ctx) {
        if (k - 1 > n_workers_) return false;
        if (busy_.exchange(true, std::memory_order_acquire)) return false;
        {
            std::lock_guard<std::mutex> lk(m_);
            call_ = call; ctx_ = ctx; k_ = k; remaining_ = k - 1; gen_++;
        }
        cv_.notify_all();
        call(ctx, 0);
        {
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [&] { return remaining_ == 0; });
            call_ = nullptr; ctx_ = nullptr; k_ = 0;
        }
        busy_.store(false, std::memory_order_release);
        return true;
    }

private:
    HostPool() : pid_(getpid()) {}
    bool start(int n) {
        try { for (int w = 1; w <= n; w++) { std::thread([this, w] { work(w); }).detach(); n_workers_ = w; } }
        catch (...) { return n_workers_ > 0; }  // (a machine that refuses more threads: the pool is as large as it got)
        return n_workers_ > 0;
    }
    void retire() { { std::lock_guard<std::mutex> lk(m_); quit_ = true; } cv_.notify_all(); }  // its workers leave; the object stays allocated (they may still be reading it)
    void work(int w) {
        uint64_t seen = 0;
        for (;;) {
            void (*call)(void*, int) = nullptr; void* ctx = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || gen_ != seen; });
                if (quit_) return;
                seen = gen_;
                if (w < k_) { call = call_; ctx = ctx_; }
            }
            if (!call) continue;
            call(ctx, w);  // (never throws: parallel_chunks catches inside)
            bool last;
            { std::lock_guard<std::mutex> lk(m_); last = --remaining_ == 0; }
            if (last) done_.notify_one();
        }
    }
    const pid_t pid_;
    int n_workers_ = 0;
    std::atomic<bool> busy_{false};
    std::mutex m_;
    std::condition_variable cv_, done_;
    uint64_t gen_ = 0; int k_ = 0, remaining_ = 0; bool quit_ = false;
    void (*call_)(void*, int) = nullptr; void* ctx_ = nullptr;
};

template <class F>
inline void parallel_chunks(size_t n, F&& f, size_t min_chunk = 16384) {
    const int k = chunk_count(n, min_chunk);
    if (k <= 1) { f(0, (size_t)0, n); return; }
    auto bound = [&](int i) { return (size_t)((unsigned __int128)n * (unsigned)i / (unsigned)k); };
    // an exception in a worker (std::bad_alloc) would end the process there: it is carried to the calling thread and rethrown after the join
    std::vector<std::exception_ptr> err((size_t)k);
    auto run = [&](int i) { try { f(i, bound(i), bound(i + 1)); } catch (...) { err[(size_t)i] = std::current_exception(); } };
    bool done = false;
    if (HostPool* pool = HostPool::get()) done = pool->run(k, [](void* c, int i) { (*static_cast<decltype(run)*>(c))(i); }, &run);
    if (!done) {
        // own threads; a machine that refuses one (std::system_error) has this thread run the chunks that got none — never an unwinding vector of joinable threads (std::terminate)
        std::vector<std::thread> th; th.reserve((size_t)k - 1);
        int started = 1;
        try { for (; started < k; started++) th.emplace_back([&run, i = started] { run(i); }); } catch (...) {}
        run(0);
        for (int i = started; i < k; i++) run(i);
        for (auto& t : th) t.join();
    }
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
