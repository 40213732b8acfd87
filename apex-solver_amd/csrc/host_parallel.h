// host_parallel.h -- a minimal fork-join loop over std::thread for the one-off structure set-up (no OpenMP runtime
// dependency in the product library).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace apex {

inline unsigned host_threads() {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    return std::min<unsigned>(nt, 64);
}

// f(begin, end) over [0, n) in dynamic chunks of `grain`; serial when the range is small
template <typename F>
void parallel_ranges(int64_t n, int64_t grain, F&& f) {
    const unsigned nt = host_threads();
    if (n <= grain || nt == 1) { if (n > 0) f((int64_t)0, n); return; }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> th;
    const unsigned use = (unsigned)std::min<int64_t>(nt, (n + grain - 1) / grain);
    for (unsigned t = 0; t < use; ++t)
        th.emplace_back([&] {
            for (;;) {
                const int64_t b = next.fetch_add(grain);
                if (b >= n) break;
                f(b, std::min<int64_t>(n, b + grain));
            }
        });
    for (auto& t : th) t.join();
}

// f(i) for i in [0, n)
template <typename F>
void parallel_rows(int64_t n, F&& f, int64_t grain = 16) {
    parallel_ranges(n, grain, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) f(i); });
}

}  // namespace apex
