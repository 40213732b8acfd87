// host_pool_harness.cpp -- CPU-only checks of apex-solver_amd/csrc/host_parallel.h (tests/test_host_pool.py):
// exceptions thrown inside a pooled loop come back on the calling thread, the pool stays usable, nested loops of
// DIFFERENT instantiations run serially instead of deadlocking on the pool.
#include <atomic>
#include <new>
#include <stdexcept>
#include <vector>

#include "host_parallel.h"

extern "C" {

// a loop of n rows in which row `bad` throws; which = 0: std::bad_alloc, 1: std::runtime_error.
// returns 1 = bad_alloc caught on the caller, 2 = runtime_error caught, 0 = nothing thrown, -1 = something else.
// *rows_done: rows that completed; *after_ok: 1 when a following loop over the same pool visited every row once.
int hp_throw(long n, long bad, int which, long* rows_done, int* after_ok) {
    std::atomic<long> done(0);
    int rc = 0;
    try {
        apex::parallel_rows(n, [&](int64_t i) {
            std::vector<int> scratch(64, (int)i);   // (the real bodies allocate)
            if (i == bad) {
                if (which == 0) throw std::bad_alloc();
                throw std::runtime_error("row failed");
            }
            done.fetch_add(1);
        }, 4);
    } catch (const std::bad_alloc&) { rc = 1; }
    catch (const std::runtime_error&) { rc = 2; }
    catch (...) { rc = -1; }
    *rows_done = done.load();
    std::vector<std::atomic<int>> hits(n);
    for (auto& h : hits) h.store(0);
    apex::parallel_rows(n, [&](int64_t i) { hits[i].fetch_add(1); }, 4);
    int ok = 1;
    for (auto& h : hits) ok &= (h.load() == 1);
    *after_ok = ok;
    return rc;
}

// an outer loop (one lambda type) whose body starts inner loops of two OTHER lambda types: must finish, sum checked
long hp_nested(long n_outer, long n_inner) {
    std::atomic<long> sum(0);
    apex::parallel_rows(n_outer, [&](int64_t o) {
        apex::parallel_ranges(n_inner, 8, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) sum.fetch_add(1); });
        apex::parallel_rows(n_inner, [&](int64_t i) { sum.fetch_add((long)(i & 1)); }, 8);
    }, 1);
    return sum.load();
}

unsigned hp_threads() { return apex::host_threads(); }
}
