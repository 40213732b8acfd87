// host_parallel.h -- a minimal fork-join loop over std::thread for the one-off structure set-up (no OpenMP runtime
// dependency in the product library).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <sched.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdio>
#include <exception>
#include <memory>
#include <new>
#include <thread>
#include <utility>
#include <vector>

namespace apex {

// Threads of the set-up's loops: the CPUs this process may really use -- the affinity mask cut by the cgroup CPU quota --, not
// the hardware threads the machine shows (a GPU box of this pool shows 256 and grants 16: 128 pooled threads took 0.17 s to
// start and were throttled by the quota for the rest of the set-up; round 5).  APEX_HOST_THREADS overrides.
inline unsigned host_threads() {
    static const unsigned n = [] {
        if (const char* e = getenv("APEX_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return (unsigned)std::min(v, 256); }
        unsigned nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 4;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) nt = std::min<unsigned>(nt, (unsigned)c); }
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
            char q[64]; long long period = 0;
            if (fscanf(f, "%63s %lld", q, &period) == 2 && q[0] != 'm' && period > 0) {
                const long long quota = atoll(q);
                if (quota > 0) nt = std::min<unsigned>(nt, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
            }
            fclose(f);
        } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
            long long quota = -1, period = 0;
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 0; fclose(h); }
            if (quota > 0 && period > 0) nt = std::min<unsigned>(nt, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        }
        return std::min<unsigned>(std::max<unsigned>(nt, 1), 128);
    }();
    return n;
}

// A persistent pool: the set-up runs some thirty parallel loops and creating 128 threads for each costs more than most of
// the loops themselves.  Workers are created on first use and live until the process ends; one loop at a time (callers
// are serialised), the calling thread works too.
class HostPool {
   public:
    static HostPool& get() { static HostPool p; return p; }
    // run job(worker) on `use` workers (including the caller) and wait.  Exception-safe: a throw in ANY participant
    // (the bodies allocate: std::bad_alloc) is caught where it happens, the first one is kept, every worker is still
    // waited for -- `job` and the caller's loop state live on this stack frame -- and the exception is rethrown on the
    // calling thread, where capi.cpp's guarded() turns it into a status code.  Nothing ever escapes a worker thread
    // (that would be std::terminate in the host process).
    template <typename J>
    void run(unsigned use, J&& job) {
        std::lock_guard<std::mutex> serial(run_mu_);
        if (use <= 1) { job(); return; }
        ensure(use - 1);     // (may throw std::system_error before anything is published: nothing to undo)
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = [&job]() { job(); };
            want_ = use - 1; started_ = 0; pending_ = use - 1;
            error_ = nullptr;
            ++epoch_;
        }
        cv_.notify_all();
        std::exception_ptr mine;
        try { job(); } catch (...) { mine = std::current_exception(); }
        std::exception_ptr first;
        {
            std::unique_lock<std::mutex> lk(mu_);
            done_cv_.wait(lk, [&] { return pending_ == 0; });
            fn_ = nullptr;
            first = error_ ? error_ : mine;
            error_ = nullptr;
        }
        if (first) std::rethrow_exception(first);
    }
    // one flag per thread for ALL loop instantiations: a loop started from inside any other loop must run serially
    // (run_mu_ is not recursive, and a worker that waits for the pool waits for itself)
    static bool& inside_loop() { static thread_local bool inside = false; return inside; }

   private:
    HostPool() = default;
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void ensure(unsigned n) {
        while (th_.size() < n) th_.emplace_back([this] { loop(); });
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && started_ < want_); });
                if (stop_) return;
                seen = epoch_;
                ++started_;
                fn = fn_;
            }
            std::exception_ptr err;
            try { fn(); } catch (...) { err = std::current_exception(); }
            fn = nullptr;    // (drop the reference to the caller's frame before reporting completion)
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (err && !error_) error_ = err;
                if (--pending_ == 0) done_cv_.notify_all();
            }
        }
    }
    std::mutex run_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> th_;
    std::function<void()> fn_;
    std::exception_ptr error_;
    unsigned want_ = 0, started_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

// f(begin, end) over [0, n) in dynamic chunks of `grain`; serial when the range is small
template <typename F>
void parallel_ranges(int64_t n, int64_t grain, F&& f) {
    const unsigned nt = host_threads();
    bool& inside = HostPool::inside_loop();   // a loop started from inside a loop (of any instantiation) runs serially
    if (n <= grain || nt == 1 || inside) { if (n > 0) f((int64_t)0, n); return; }
    std::atomic<int64_t> next(0);
    const unsigned use = (unsigned)std::min<int64_t>(nt, (n + grain - 1) / grain);
    HostPool::get().run(use, [&] {
        bool& in = HostPool::inside_loop();
        struct Reset { bool& b; ~Reset() { b = false; } } reset{in};   // also on the exceptional path
        in = true;
        for (;;) {
            const int64_t b = next.fetch_add(grain);
            if (b >= n) break;
            try {
                f(b, std::min<int64_t>(n, b + grain));
            } catch (...) {
                next.store(n);     // the other participants stop at their next chunk
                throw;
            }
        }
    });
}

// f(i) for i in [0, n)
template <typename F>
void parallel_rows(int64_t n, F&& f, int64_t grain = 16) {
    parallel_ranges(n, grain, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) f(i); });
}

// APEX_SETUP_TRACE=1: sub-phase times of the structure set-up on stderr
struct SetupTrace {
    bool on; double t0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    SetupTrace() : on(getenv("APEX_SETUP_TRACE") != nullptr), t0(now()) {}
    void mark(const char* what) {
        if (!on) return;
        const double t = now();
        fprintf(stderr, "[setup] %-28s %7.1f ms\n", what, (t - t0) * 1e3);
        t0 = t;
    }
};

// The big blocks of the structure set-up (observation lists, the pair list: 3.7 GB on final-13682) come from a process-wide
// CACHE and go back to it (round 5).  Returning them to the system costs 0.2 s of page zapping wherever it is done -- and done on
// a background thread it still stalls the caller: while the pages of buffers the GPU driver has seen (hipMemcpy from pageable
// memory registers them) are unmapped, the caller's next apexgpu_set_params took 0.2-0.39 s instead of 0.012
// (tools/setup_probe.py, APEX_SETUP_FREE=bg / sync / leak).  A block handed back is kept (up to kKeepBytes in all) and serves the
// next handle's set-up, which then also skips the first-touch page faults of fresh memory.
// Round 6 (what the cache may hold, and for how long): only blocks of the LAST set-up (end_setup drops what an older, larger
// structure left behind), only while a solver handle is alive (release() by the last ~Solver returns everything: the page
// zapping then falls into apexgpu_destroy), never more than kKeepBytes; apexgpu_trim_host_cache() returns everything at once,
// APEX_HOST_CACHE=0 turns the cache off, apexgpu_host_cache_bytes() says what it holds.  The object is never destroyed (a thread
// of a leaked handle may still give a block back at process exit).
class HostBlockCache {
   public:
    static HostBlockCache& get() { static HostBlockCache* c = new HostBlockCache; return *c; }
    void retain() { std::lock_guard<std::mutex> lk(mu_); ++handles_; }
    void release() { bool last; { std::lock_guard<std::mutex> lk(mu_); last = --handles_ <= 0; if (last) handles_ = 0; } if (last) (void)trim(); }
    void begin_setup() { std::lock_guard<std::mutex> lk(mu_); ++gen_; }
    size_t end_setup() {   // the blocks no set-up since begin_setup() has used go back to the system; returns the bytes released
        std::vector<Block> drop;
        size_t n = 0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t i = 0; i < kept_.size();)
                if (kept_[i].gen < gen_) { drop.push_back(kept_[i]); n += kept_[i].bytes; kept_bytes_ -= kept_[i].bytes; kept_.erase(kept_.begin() + (long)i); }
                else ++i;
        }
        for (const Block& b : drop) free(b.p);
        return n;
    }
    void* take(size_t bytes) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            int best = -1;   // the smallest kept block that fits and is not more than twice the request
            for (int i = 0; i < (int)kept_.size(); ++i)
                if (kept_[i].bytes >= bytes && kept_[i].bytes <= 2 * bytes && (best < 0 || kept_[i].bytes < kept_[best].bytes)) best = i;
            if (best >= 0) {
                void* p = kept_[best].p;
                kept_bytes_ -= kept_[best].bytes;
                kept_[best].gen = gen_;
                live_.push_back(kept_[best]);
                kept_.erase(kept_.begin() + best);
                return p;
            }
        }
        void* p = nullptr;
        if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0 || !p) throw std::bad_alloc();
        (void)madvise(p, bytes, MADV_HUGEPAGE);
        std::lock_guard<std::mutex> lk(mu_);
        live_.push_back(Block{p, bytes, gen_});
        return p;
    }
    void give(void* p, size_t) {
        std::lock_guard<std::mutex> lk(mu_);
        size_t bytes = 0;
        int gen = 0;
        for (size_t i = 0; i < live_.size(); ++i)
            if (live_[i].p == p) { bytes = live_[i].bytes; gen = live_[i].gen; live_.erase(live_.begin() + (long)i); break; }
        if (bytes == 0 || !enabled_ || handles_ <= 0 || kept_bytes_ + bytes > kKeepBytes) { free(p); return; }
        kept_.push_back(Block{p, bytes, gen});
        kept_bytes_ += bytes;
    }
    size_t trim() {   // everything kept goes back to the system; returns the bytes released
        std::vector<Block> drop;
        size_t n = 0;
        { std::lock_guard<std::mutex> lk(mu_); drop.swap(kept_); n = kept_bytes_; kept_bytes_ = 0; }
        for (const Block& b : drop) free(b.p);
        return n;
    }
    size_t kept_bytes() { std::lock_guard<std::mutex> lk(mu_); return kept_bytes_; }

   private:
    struct Block { void* p; size_t bytes; int gen; };
    HostBlockCache() { const char* e = getenv("APEX_HOST_CACHE"); enabled_ = !(e && e[0] == '0'); }
    static constexpr size_t kKeepBytes = (size_t)8 << 30;
    std::mutex mu_;
    std::vector<Block> kept_, live_;
    size_t kept_bytes_ = 0;
    int handles_ = 0, gen_ = 0;
    bool enabled_ = true;
};

// A vector whose resize() leaves trivially constructible elements UNINITIALISED: the set-up fills hundreds of megabytes
// of lists from parallel loops, and a value-initialising resize would first touch (and page in) all of it from one thread.
template <typename T>
struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U>&) {}
    // big blocks: 2 MB aligned and advised for transparent huge pages (first touch of a fresh 1.5 GB list is otherwise
    // 400 K page faults)
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes >= (size_t)32 << 20) return static_cast<T*>(HostBlockCache::get().take(bytes));
        void* p = malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t n) {
        if (n * sizeof(T) >= (size_t)32 << 20) HostBlockCache::get().give(p, n * sizeof(T));
        else free(p);
    }
    template <typename U> void construct(U* p) { ::new (static_cast<void*>(p)) U; }
    template <typename U, typename... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <typename T> using raw_vector = std::vector<T, NoInitAlloc<T>>;

// Stable counting sort of the indices 0..n-1 by key[i] in [0, n_keys): ptr[k] .. ptr[k+1] = the indices with key k, in
// increasing order.  Few keys (cameras): per-thread histograms over contiguous index chunks.
template <typename K>
void parallel_bucket_small(int64_t n, int64_t n_keys, const K* key, std::vector<int>& ptr, raw_vector<int>& idx) {
    const int64_t nth = std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / 65536));
    const int64_t per = (n + nth - 1) / nth;
    std::vector<std::vector<int>> hist(nth, std::vector<int>());
    parallel_rows(nth, [&](int64_t t) {
        auto& h = hist[t]; h.assign(n_keys, 0);
        for (int64_t i = t * per; i < std::min(n, (t + 1) * per); ++i) h[key[i]]++;
    }, 1);
    ptr.assign(n_keys + 1, 0);
    int64_t run = 0;
    for (int64_t k = 0; k < n_keys; ++k) {
        ptr[k] = (int)run;
        for (int64_t t = 0; t < nth; ++t) { const int c = hist[t][k]; hist[t][k] = (int)run; run += c; }
    }
    ptr[n_keys] = (int)run;
    idx.resize(n);
    parallel_rows(nth, [&](int64_t t) {
        auto& h = hist[t];
        for (int64_t i = t * per; i < std::min(n, (t + 1) * per); ++i) idx[h[key[i]]++] = (int)i;
    }, 1);
}

// The same with many keys (landmarks): every thread owns a contiguous KEY range, streams over all indices and takes the
// ones in its range -- its output range is contiguous, no histogram per thread.
template <typename K, typename P>
void parallel_bucket_large(int64_t n, int64_t n_keys, const K* key, std::vector<P>& ptr, raw_vector<int>& idx) {
    // already grouped (BAL files list the observations landmark by landmark): the buckets are the runs
    std::atomic<int> sorted(1);
    parallel_ranges(n, 1 << 18, [&](int64_t b, int64_t e) {
        for (int64_t i = std::max<int64_t>(b, 1); i < e; ++i)
            if (key[i] < key[i - 1]) { sorted.store(0, std::memory_order_relaxed); return; }
    });
    if (sorted.load() && n > 0) {
        ptr.assign(n_keys + 1, 0);
        idx.resize(n);
        parallel_ranges(n, 1 << 18, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) {
                idx[i] = (int)i;
                const int64_t prev = i > 0 ? (int64_t)key[i - 1] : -1;
                for (int64_t k = prev + 1; k <= (int64_t)key[i]; ++k) ptr[k] = (P)i;   // first index with key >= k
            }
        });
        for (int64_t k = (int64_t)key[n - 1] + 1; k <= n_keys; ++k) ptr[k] = (P)n;
        return;
    }
    const int64_t nth = std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / 65536));
    const int64_t per = (n_keys + nth - 1) / nth;
    ptr.assign(n_keys + 1, 0);
    parallel_rows(nth, [&](int64_t t) {
        const int64_t k0 = t * per, k1 = std::min(n_keys, (t + 1) * per);
        for (int64_t i = 0; i < n; ++i) { const int64_t k = (int64_t)key[i]; if (k >= k0 && k < k1) ptr[k + 1]++; }
    }, 1);
    for (int64_t k = 0; k < n_keys; ++k) ptr[k + 1] += ptr[k];
    idx.resize(n);
    parallel_rows(nth, [&](int64_t t) {
        const int64_t k0 = t * per, k1 = std::min(n_keys, (t + 1) * per);
        if (k0 >= k1) return;
        std::vector<P> fill(ptr.begin() + k0, ptr.begin() + k1);
        for (int64_t i = 0; i < n; ++i) { const int64_t k = (int64_t)key[i]; if (k >= k0 && k < k1) idx[fill[k - k0]++] = (int)i; }
    }, 1);
}

}  // namespace apex
