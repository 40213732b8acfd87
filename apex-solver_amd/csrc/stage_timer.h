// stage_timer.h -- per-stage device time from HIP events recorded on the solver's own stream.
//
// begin()/end() only record events (no synchronisation), so a timed region is not perturbed by the
// measurement; the pairs are resolved when times() / reset() is called.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>
#include <vector>

namespace apex {

template <int N>
class StageTimer {
   public:
    ~StageTimer() {
        resolve();
        for (hipEvent_t e : pool_) (void)hipEventDestroy(e);
    }
    void enable(bool on) { on_ = on; mask_ = ~0u; }
    void enable_only(uint32_t stage_mask) { on_ = stage_mask != 0; mask_ = stage_mask; }   // bit k: stage k is timed
    bool enabled() const { return on_; }
    void begin(int st, hipStream_t s) {
        if (!on_ || !((mask_ >> st) & 1u)) return;
        hipEvent_t a = take(), b = take();
        (void)hipEventRecord(a, s);
        open_[st] = {a, b};
    }
    void end(int st, hipStream_t s) {
        if (!on_ || !open_[st].first) return;
        (void)hipEventRecord(open_[st].second, s);
        pending_.push_back({st, open_[st]});
        open_[st] = {nullptr, nullptr};
    }
    void reset() {
        resolve();
        for (int i = 0; i < N; ++i) { ms_[i] = 0; n_[i] = 0; }
    }
    int times(double* ms, int64_t* launches) {
        resolve();
        for (int i = 0; i < N; ++i) { ms[i] = ms_[i]; launches[i] = n_[i]; }
        return N;
    }

   private:
    hipEvent_t take() {
        hipEvent_t e = nullptr;
        if (!pool_.empty()) { e = pool_.back(); pool_.pop_back(); } else (void)hipEventCreate(&e);
        return e;
    }
    void resolve() {
        for (auto& p : pending_) {
            (void)hipEventSynchronize(p.second.second);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) { ms_[p.first] += ms; n_[p.first] += 1; }
            pool_.push_back(p.second.first);
            pool_.push_back(p.second.second);
        }
        pending_.clear();
    }
    bool on_ = false;
    uint32_t mask_ = ~0u;
    std::vector<hipEvent_t> pool_;
    std::pair<hipEvent_t, hipEvent_t> open_[N] = {};
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending_;
    double ms_[N] = {0};
    int64_t n_[N] = {0};
};

}  // namespace apex
