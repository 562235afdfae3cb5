// Stress of kai_parallel.hpp (the host preparation's parallel primitive and its pool of waiting workers): sequential loops, concurrent callers (the pool is taken: own threads),
// a loop inside a loop, an exception in a chunk, forked children (a child finds another pid in the pool and starts its own), forks while another thread runs loops.
// Built and run by tests/test_host_pool.py; prints "pool stress ok".
#include "../../kai-scheduler_amd/csrc/kai_parallel.hpp"
#include <cstdio>
#include <numeric>
#include <sys/wait.h>
using namespace kai;
static long job(size_t n) {
    std::vector<long> part((size_t)chunk_count(n), 0);
    parallel_chunks(n, [&](int ci, size_t a, size_t b) { long s = 0; for (size_t i = a; i < b; i++) s += (long)i; part[(size_t)ci] = s; });
    return std::accumulate(part.begin(), part.end(), 0L);
}
int main() {
    const size_t n = 1 << 20; const long want = (long)n * (long)(n - 1) / 2;
    // sequential
    for (int i = 0; i < 2000; i++) if (job(n) != want) { puts("BAD seq"); return 1; }
    // concurrent callers (pool contention -> fallback)
    { std::vector<std::thread> th; std::atomic<int> bad{0};
      for (int t = 0; t < 6; t++) th.emplace_back([&] { for (int i = 0; i < 500; i++) if (job(n) != want) bad++; });
      for (auto& t : th) t.join(); if (bad) { puts("BAD conc"); return 1; } }
    // nested
    { std::atomic<long> tot{0}; parallel_chunks(n, [&](int, size_t a, size_t b) { if (job(1 << 18) != (long)(1 << 18) * ((1 << 18) - 1) / 2) tot += 1; (void)a; (void)b; }); if (tot) { puts("BAD nested"); return 1; } }
    // exceptions
    { bool caught = false; try { parallel_chunks(n, [&](int ci, size_t, size_t) { if (ci == chunk_count(n) - 1) throw std::bad_alloc(); }); } catch (const std::bad_alloc&) { caught = true; } if (!caught) { puts("BAD exc"); return 1; }
      if (job(n) != want) { puts("BAD after exc"); return 1; } }
    // fork: child uses the pool
    for (int r = 0; r < 5; r++) { pid_t c = fork(); if (c == 0) { for (int i = 0; i < 200; i++) if (job(n) != want) _exit(2); _exit(0); } int st = 0; waitpid(c, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) { puts("BAD fork"); return 1; } }
    // fork while another thread is running jobs
    { std::atomic<bool> stop{false}; std::thread bg([&] { while (!stop) job(n); });
      for (int r = 0; r < 20; r++) { pid_t c = fork(); if (c == 0) { for (int i = 0; i < 50; i++) if (job(n) != want) _exit(2); _exit(0); } int st = 0; waitpid(c, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) { puts("BAD fork2"); stop = true; bg.join(); return 1; } }
      stop = true; bg.join(); }
    puts("pool stress ok");
}
