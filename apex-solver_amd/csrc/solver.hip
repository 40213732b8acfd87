// solver.hip -- host orchestration of the MI355X bundle-adjustment backend (see solver.h).
#include "solver.h"
#include "ba_device.hpp"
#include "ba_structure.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <future>
#include <numeric>

namespace apex {

void warm_ba_kernels(hipStream_t s);
void warm_schur_pairs(hipStream_t s);
void warm_chol_kernels(hipStream_t s);

#define HIP_TRY(expr)                                         \
    do {                                                      \
        int _rc = check_hip((expr), #expr);                   \
        if (_rc != kOk) return _rc;                           \
    } while (0)

// every collective's result is surfaced as APEXGPU_ERR_DEVICE with the transport's own text (comm.h)
#define COMM_TRY(expr)                                                                  \
    do {                                                                                \
        if (!(expr)) return fail(kDeviceError, comm_->error());                         \
    } while (0)

template <typename T>
static hipError_t dev_alloc(T** p, size_t n) {
    return hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
}

// OptimizeParams<POSE, LANDMARK, INTRINSIC> of every mode as 4 POSE + 2 LANDMARK + INTRINSIC (src/factors/mod.rs:82-101)
int mode_mask(int mode) {
    static const int m[7] = {6 /* BundleAdjustment */, 7 /* SelfCalibration */, 4 /* OnlyPose */, 2 /* OnlyLandmarks */,
                             1 /* OnlyIntrinsics */, 5 /* PoseAndIntrinsics */, 3 /* LandmarksAndIntrinsics */};
    return (mode >= 0 && mode < 7) ? m[mode] : 7;
}

Solver::Solver(int64_t n_cam, int64_t n_pt, int64_t n_obs, int mode, int device)
    : n_cam_(n_cam), n_pt_(n_pt), n_obs_(n_obs), mode_(mode), dc_(mode_mask(mode) & 1 ? 9 : 6), device_(device) {
    lm_lo_ = 0; lm_hi_ = n_pt;
    HostBlockCache::get().retain();   // (the set-up's host blocks are cached only while a handle is alive: host_parallel.h)
}

Solver::~Solver() {
    if (free_thread_.joinable()) free_thread_.join();
    HostBlockCache::get().release();   // the last handle returns the cached blocks to the system
    hipSetDevice(device_);
    if (stream_) hipStreamSynchronize(stream_);
    void* ptrs[] = {poses_[0], poses_[1], intr_[0], intr_[1], pts_[0], pts_[1], camp_[0], camp_[1], ptasks_, pchunks_, pblocks_, precs_, pqdesc_, orec_, o_slot_, wg_cam_n_, wg_cam_list_, o_cam_, o_pt_, o_uv_, o_orig_, pt_ptr_,
                    cam_ptr_, cam_obs_, co_pt_, co_uv_, co_rank_, fix_pose_, fix_intr_, fix_pt_, g_c_, g_red_,
                    dcam_, hinv_, g_l_, dl_, partial_, scal_, flags_, pcg_buf_, lmu_, sd_, minv_, cam_scale_, pt_scale_, lam_mask_};
    for (void* p : ptrs)
        if (p) hipFree(p);
    comm_.reset();
    if (pcg_host_) (void)hipHostFree(pcg_host_);
    if (eager_host_) (void)hipHostFree(eager_host_);
    for (hipEvent_t e : pcg_ev_) if (e) (void)hipEventDestroy(e);
    for (int b = 0; b < 2; ++b) {
        if (pin_[b]) (void)hipHostFree(pin_[b]);
        if (pin_ev_[b]) (void)hipEventDestroy(pin_ev_[b]);
    }
    if (stream_) hipStreamDestroy(stream_);
}

// A hipMemcpy from pageable memory is staged by the runtime in small pieces (0.19 s for the 107 MB of final-13682's points:
// 0.56 GB/s); through two pinned 16 MB chunks the copy runs at memcpy speed with the DMA of one chunk under the memcpy of the
// next.  Asynchronous on stream_ like the copies it replaces (the caller synchronises).
int Solver::upload_staged(void* dst_dev, const void* src_host, size_t bytes) {
    constexpr size_t kChunk = (size_t)16 << 20;
    if (bytes <= ((size_t)1 << 20)) return check_hip(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, stream_), "upload");
    for (int b = 0; b < 2; ++b) {
        if (!pin_[b]) HIP_TRY(hipHostMalloc(&pin_[b], kChunk, hipHostMallocDefault));
        if (!pin_ev_[b]) HIP_TRY(hipEventCreateWithFlags(&pin_ev_[b], hipEventDisableTiming));
    }
    size_t i = 0;
    for (size_t off = 0; off < bytes; off += kChunk, ++i) {
        const int b = (int)(i & 1);
        const size_t n = std::min(kChunk, bytes - off);
        // a pinned chunk is rewritten only when the DMA that last read it has finished -- also the one a PREVIOUS call (or a
        // call that returned early on an error) left in flight
        if (pin_busy_[b]) { HIP_TRY(hipEventSynchronize(pin_ev_[b])); pin_busy_[b] = false; }
        memcpy(pin_[b], static_cast<const char*>(src_host) + off, n);
        HIP_TRY(hipMemcpyAsync(static_cast<char*>(dst_dev) + off, pin_[b], n, hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipEventRecord(pin_ev_[b], stream_));
        pin_busy_[b] = true;
    }
    return kOk;
}

int Solver::fail(int code, const std::string& msg) {
    err_ = msg;
    return code;
}

int Solver::check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return kOk;
    return fail(kDeviceError, std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

BAView Solver::view(int which) const {
    BAView v;
    v.n_cam = n_cam_; v.n_pt = n_pt_; v.n_obs = (int64_t)o_orig_h_.size();
    v.camp = camp_[which]; v.camq = camp_[which] + (size_t)kCamStride * n_cam_; v.pts = pts_[which];
    v.o_cam = o_cam_; v.o_pt = o_pt_; v.o_uv = o_uv_; v.pt_ptr = pt_ptr_;
    v.huber_delta = huber_delta_;
    v.mask_code = mode_mask(mode_);
    v.co_pt = co_pt_; v.co_uv = co_uv_; v.co_rank = co_rank_;
    v.cam_scale = scaled_ ? cam_scale_ : nullptr;
    v.pt_scale = scaled_ ? pt_scale_ : nullptr;
    v.lam_mask = tree_shard_ ? lam_mask_ : nullptr;
    v.o_slot = o_slot_; v.wg_cam_n = wg_cam_n_; v.wg_cam_list = wg_cam_list_;
    if (world_ > 1 && lm_hi_ > lm_lo_) {   // the rank's own landmark range, in whole workgroups of kLmWg
        v.lm_wg0 = (int)(lm_lo_ / kLmWg);
        v.lm_wgn = (int)((lm_hi_ + kLmWg - 1) / kLmWg) - v.lm_wg0;
    }
    return v;
}

TileMap Solver::tilemap() const { return tp_.tilemap(); }

void Solver::stage_begin(int st) { timer_.begin(st, stream_); }
void Solver::stage_end(int st) { timer_.end(st, stream_); }
void Solver::reset_stage_times() { timer_.reset(); }
int Solver::stage_times(double* ms, int64_t* launches) { return timer_.times(ms, launches); }

int Solver::set_shard(int rank, int world) {
    if (have_structure_) return fail(kInvalidState, "set_shard must precede set_structure");
    if (world < 1 || rank < 0 || rank >= world) return fail(kInvalidInput, "bad rank/world");
    rank_ = rank; world_ = world;
    return kOk;
}

int Solver::adopt_comm(std::unique_ptr<Communicator> c, const std::string& err) {
    if (!c) return fail(kDeviceError, err);
    comm_ = std::move(c);
    Communicator* cp = comm_.get();
    TilePlan::Comm tc;
    auto note = [this, cp](bool ok) { if (!ok) comm_err_ = cp->error(); return ok; };
    tc.sum = [cp, note](double* buf, size_t n, hipStream_t st) { return note(cp->all_reduce_sum(buf, n, st)); };
    tc.max_int = [cp, note](int* buf, size_t n, hipStream_t st) { return note(cp->all_reduce_max(buf, n, st)); };
    tp_.set_comm(std::move(tc));
    return kOk;
}

int Solver::comm_init(int world, int rank, const void* unique_id128) {
    int rc = set_shard(rank, world);
    if (rc != kOk) return rc;
    HIP_TRY(hipSetDevice(device_));
    std::string err;
    return adopt_comm(make_rccl_comm(world, rank, unique_id128, &err), err);
}

// The same multi-rank schedule over the host shared-memory transport (comm.h): ranks = processes of one node.
int Solver::comm_init_shm(int world, int rank, const char* name) {
    int rc = set_shard(rank, world);
    if (rc != kOk) return rc;
    HIP_TRY(hipSetDevice(device_));
    std::string err;
    return adopt_comm(make_shm_comm(world, rank, name, &err), err);
}

// Landmark range [lo,hi) of `rank`: contiguous, balanced by observation count.  ptr[l] = number of
// observations of landmarks < l (n_pt+1 entries).  Pure host arithmetic, identical on every rank.
void shard_range(int64_t n_pt, const int64_t* ptr, int rank, int world, int64_t* lo, int64_t* hi) {
    const int64_t n_obs = ptr[n_pt];
    auto cut = [&](int r) -> int64_t {
        if (r <= 0) return 0;
        if (r >= world) return n_pt;
        const int64_t target = (n_obs * r) / world;
        return std::min<int64_t>(std::lower_bound(ptr, ptr + n_pt + 1, target) - ptr, n_pt);
    };
    *lo = cut(rank);
    *hi = cut(rank + 1);
}

// ---------------------------------------------------------------------------------------------
// structure
// ---------------------------------------------------------------------------------------------
int Solver::set_structure(const uint32_t* cam_idx, const uint32_t* pt_idx, const double* obs_uv,
                          const int64_t* intr_col, const int64_t* pose_col, const int64_t* pt_col,
                          const uint8_t* fix_pose, const uint8_t* fix_intr, const uint8_t* fix_pt,
                          double huber_delta) {
    if (n_cam_ <= 0 || n_pt_ <= 0) return fail(kInvalidInput, n_cam_ <= 0 ? "No camera variables found" : "No landmark variables found");
    if (n_obs_ < 0 || n_obs_ > 2000000000LL) return fail(kInvalidInput, "observation count out of range");
    // The first stream this process creates costs 0.1-0.16 s (the runtime brings up its hardware queues; measured with
    // APEX_SETUP_TRACE in bench.py, torch's context already there), the code objects of the three kernel files a few ms more:
    // both on a thread, beside the argument checks and the camera order (host only), joined in front of the first device call
    // below (device_ready) -- round 5.
    double* raw_uv = nullptr;   // (device thread -> uploader thread, which joins the former through device_ready)
    const bool raw_uv_wanted = device_gathers_ && world_ == 1 && !(comm_ && world_ > 1);
    struct RawFree { double*& p; ~RawFree() { if (p) { (void)hipFree(p); p = nullptr; } } } raw_free{raw_uv};
    std::promise<hipError_t> init_p;
    std::shared_future<hipError_t> init_f = init_p.get_future().share();
    std::promise<bool> validated_p;   // (the 0.5 GB copy of the measurements does not start for a call that is about to be refused)
    std::shared_future<bool> validated_f = validated_p.get_future().share();
    std::thread warmer([this, &init_p, &raw_uv, raw_uv_wanted, obs_uv, validated_f] {
        SetupTrace wt;
        hipError_t e = hipSetDevice(device_);
        if (e == hipSuccess && !stream_) e = hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking);
        init_p.set_value(e);
        wt.mark("device thread: device, stream");
        if (e != hipSuccess) return;
        if (raw_uv_wanted && validated_f.get()) {   // the caller's measurements as they are, beside the host's list building -- once the lists are valid
            if (hipMalloc(reinterpret_cast<void**>(&raw_uv), std::max<size_t>(2 * (size_t)n_obs_, 2) * sizeof(double)) != hipSuccess ||
                hipMemcpy(raw_uv, obs_uv, 2 * (size_t)n_obs_ * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
                if (raw_uv) { (void)hipFree(raw_uv); raw_uv = nullptr; }
                (void)hipGetLastError();
            }
            wt.mark("device thread: measurements up");
        }
        hipStream_t ws = nullptr;
        if (hipStreamCreateWithFlags(&ws, hipStreamNonBlocking) != hipSuccess) return;
        warm_ba_kernels(ws); warm_schur_pairs(ws); warm_chol_kernels(ws);
        (void)hipStreamSynchronize(ws);
        (void)hipStreamDestroy(ws);
        wt.mark("device thread: code objects");
    });
    auto device_ready = [this, init_f]() -> hipError_t {   // (any thread: the calling thread's device is set as well)
        const hipError_t e = init_f.get();
        return e != hipSuccess ? e : hipSetDevice(device_);
    };
    struct WJoiner { std::thread& t; ~WJoiner() { if (t.joinable()) t.join(); } } wjoiner{warmer};
    // (declared behind the joiner: destroyed before it, so an early return releases the device thread before it is joined)
    struct Unblock { std::promise<bool>& p; bool done = false; void set(bool v) { if (!done) { done = true; p.set_value(v); } } ~Unblock() { set(false); } } validated{validated_p};
    SetupTrace tr;
    {
        std::atomic<int64_t> bad(n_obs_);   // first observation that references a missing variable
        parallel_ranges(n_obs_, 1 << 18, [&](int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i)
                if (cam_idx[i] >= (uint64_t)n_cam_ || pt_idx[i] >= (uint64_t)n_pt_) {
                    int64_t cur = bad.load();
                    while (i < cur && !bad.compare_exchange_weak(cur, i)) {}
                    return;
                }
        });
        if (bad.load() < n_obs_) return fail(kInvalidInput, "observation " + std::to_string(bad.load()) + " references a missing variable");
    }
    validated.set(true);
    huber_delta_ = huber_delta;
    intr_col_.assign(intr_col, intr_col + n_cam_);
    pose_col_.assign(pose_col, pose_col + n_cam_);
    pt_col_.assign(pt_col, pt_col + n_pt_);
    // (the caller's factor list is NOT kept: get_hessian_csc, the one reader, rebuilds it from the device's observation
    // lists on demand -- 0.1 s of set-up on final-13682 for an export the LM loop never calls)
    tr.mark("validate the index lists");

    // ---- everything derived from the observation list on the host (ba_structure.h): internal camera order (hub
    // cameras last, nested dissection of the tile graph), tile structure, landmark sharding, observation lists ------
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    HostBlockCache::get().begin_setup();
    matrix_free_only_ = matrix_free_only_opt_;   // (an automatic selection of an earlier set_structure on this handle does not stick)
    auto_fallback_ = false; fallback_reason_.clear();
    BaStructOptions so;
    so.dc = dc_; so.use_nd = use_nd_; so.nd_leaf = nd_leaf_; so.hubs_last = hubs_last_;
    so.rank = rank_; so.world = world_; so.dist_factor = dist_factor_; so.tree_sharding = tree_sharding_;
    so.dist_selftest = dist_selftest_; so.schur_form = rows_form_;
    so.device_gathers = device_gathers_ && world_ == 1;
    std::unique_ptr<BaHostStructure> hs_owner(new BaHostStructure);
    BaHostStructure& hs = *hs_owner;
    // Camera order and tile structure first; then the tile plan (symbolic fill, task lists, 1.4 GB of device allocations: 0.06-
    // 0.1 s, mostly serial) is built on a thread of its own BESIDE the observation lists (0.1 s), which do not need it (round 5).
    // A distributed plan with tree sharding previews the partition inside the list phase on the same TilePlan: no overlap then.
    hs.build_order(n_cam_, n_pt_, n_obs_, cam_idx, pt_idx, so, tp_);
    nt_ = hs.nt;
    std::vector<uint8_t> present_plan = hs.present;   // (the plan's own copy: a matrix-free handle keeps the diagonal only)
    std::string plan_err;
    double plan_seconds = 0.0;
    auto build_plan = [&]() {
        const auto t_plan = std::chrono::steady_clock::now();
        (void)hipSetDevice(device_);
        if (matrix_free_only_) {   // S is never formed: keep the diagonal tiles (Schur-Jacobi blocks are read from them), nothing else
            std::fill(present_plan.begin(), present_plan.end(), (uint8_t)0);
            for (int I = 0; I < nt_; ++I) present_plan[(size_t)I * nt_ + I] = 1;
        }
        tp_.enable_graphs(use_graphs_);
        auto_fallback_ = false; fallback_reason_.clear();
        // Variant selection by predicted cost (round 6; the reference's dispatch never fails on the fill of S and a drop-in backend
        // should not spend 10 s where it owns a 0.5 s way to the same step: levenberg_marquardt.rs:1039-1082).  The matrix-free
        // PCG costs at most its cap times one S p -- two passes over the observations, 160 bytes each at the 4.5 TB/s the two
        // kernels sustain (1.01 ms on final-13682, 0.45 ms on synthetic-10k: DESIGN section 5) -- whatever the structure; the tile
        // plan refuses to be built when its own prediction (TilePlan::predict_solve_ms) is above that.  Both numbers are host
        // arithmetic on the replicated structure: every rank decides alike.  "variant_cost_permille" scales the matrix-free side
        // (tests move the crossover onto small problems; 0: the rule is off).
        pred_mf_ms_ = 500.0 * (160.0 * (double)n_obs_ / 4.5e12 * 1e3 + 0.02) * (double)variant_cost_permille_ / 1000.0;
        pred_direct_ms_ = 0.0; variant_choice_ = matrix_free_only_ ? 3 : 0;
        tp_.set_cost_limit_ms((auto_variant_ && !matrix_free_only_ && variant_cost_permille_ > 0) ? pred_mf_ms_ : 0.0);
        std::string e = tp_.build(nt_, present_plan, stream_);
        if (!matrix_free_only_) pred_direct_ms_ = tp_.predicted_ms();
        // A structure whose direct factorisation is out of reach (a photo collection: S dense at tile granularity) is not an
        // error of the caller's: the reference's LM never fails on the fill of S.  The handle becomes matrix-free only by itself
        // and answers every variant with the matrix-free PCG (set_auto_variant).  The update-list rule is pure host arithmetic
        // on the replicated structure (every rank decides alike); the memory rule depends on the device and is single-rank only.
        const bool refused_size = tp_.refused_too_large(), refused_mem = tp_.refused_no_memory() && world_ == 1, refused_cost = tp_.refused_by_cost();
        if (!e.empty() && auto_variant_ && !matrix_free_only_ && (refused_size || refused_mem || refused_cost)) {
            auto_fallback_ = true; matrix_free_only_ = true;
            variant_choice_ = refused_cost ? 1 : 2;
            fallback_reason_ = (refused_cost ? "matrix-free PCG (IterativeSchurSolver semantics) selected by predicted cost (" : "the direct factorisation of S was refused (") + e +
                               (refused_cost ? ")" : "): matrix-free PCG (IterativeSchurSolver semantics) selected");
            tp_.set_cost_limit_ms(0.0);
            std::fill(present_plan.begin(), present_plan.end(), (uint8_t)0);
            for (int I = 0; I < nt_; ++I) present_plan[(size_t)I * nt_ + I] = 1;
            e = tp_.build(nt_, present_plan, stream_);
        }
        plan_err = e;
        plan_seconds = since(t_plan);
    };
    tr.mark("camera order, tile structure");
    std::thread planner;
    struct PJoiner { std::thread& t; ~PJoiner() { if (t.joinable()) t.join(); } } pjoiner{planner};
    const bool plan_beside_lists = !BaHostStructure::needs_owner_preview(so);
    if (plan_beside_lists)
        planner = std::thread([&] {
            try {
                if (device_ready() != hipSuccess) { plan_err = "the device could not be initialised"; return; }
                build_plan();
            } catch (const std::exception& ex) { plan_err = std::string("tile plan: ") + ex.what(); }
        });
    {
        const std::string e = hs.build_obs_lists(cam_idx, pt_idx, obs_uv, so, tp_);
        if (!e.empty()) return fail(kInvalidInput, e);
    }
    tr.mark("observation lists");
    HIP_TRY(device_ready());
    tr.mark("waited for the device thread");
    n_c_ = hs.n_c; nt_ = hs.nt; n_c_pad_ = hs.n_c_pad;
    cmap_ = hs.cmap; cinv_ = hs.cinv; lmap_ = hs.lmap;
    lm_lo_ = hs.lm_lo; lm_hi_ = hs.lm_hi; tree_shard_ = hs.tree_shard; pad_rank_ = hs.pad_rank;
    n_hubs_ = hs.n_hubs; n_border_tiles_ = hs.n_border_tiles;
    o_orig_h_.assign(hs.o_orig.begin(), hs.o_orig.end());
    n_pairs_ = hs.n_pairs; n_present_ = hs.n_present;
    if (lam_mask_) { hipFree(lam_mask_); lam_mask_ = nullptr; }
    if (tree_shard_) {
        HIP_TRY(dev_alloc(&lam_mask_, hs.lam_mask.size()));
        HIP_TRY(hipMemcpy(lam_mask_, hs.lam_mask.data(), hs.lam_mask.size(), hipMemcpyHostToDevice));
    }
    // ---- uploads of everything that does not depend on the tile plan, on a thread of their own: the observation lists
    // (1.7 GB on final-13682), the camera staging lists, the masks and the work arrays go to the device while this thread
    // builds the tile plan and the pair list (round 5: 0.09 s of copies under 0.27 s of host work) -------------------------
    const auto t_up = std::chrono::steady_clock::now();
    double up_seconds = 0.0;
    const auto &o_cam = hs.o_cam, &o_pt = hs.o_pt, &co_pt = hs.co_pt;
    const auto &o_uv = hs.o_uv, &co_uv = hs.co_uv;
    const auto &pt_ptr = hs.pt_ptr, &cam_ptr = hs.cam_ptr;
    const auto &cam_obs = hs.cam_obs, &co_rank = hs.co_rank;
    auto up = [&](auto** dptr, const auto& hv) -> hipError_t {
        using T = typename std::remove_reference<decltype(hv)>::type::value_type;
        if (*dptr) { hipFree(*dptr); *dptr = nullptr; }
        hipError_t e = dev_alloc(reinterpret_cast<T**>(dptr), hv.size());
        if (e != hipSuccess) return e;
        if (hv.empty()) return hipSuccess;
        return hipMemcpy(*dptr, hv.data(), hv.size() * sizeof(T), hipMemcpyHostToDevice);
    };
    auto alloc = [&](double** p, size_t n) -> hipError_t {
        if (*p) { hipFree(*p); *p = nullptr; }
        hipError_t e = dev_alloc(p, n);
        if (e != hipSuccess) return e;
        return hipMemset(*p, 0, std::max<size_t>(n, 1) * sizeof(double));
    };
    // (the uploader thread keeps its own error text: err_ belongs to the calling thread)
    std::string up_err;
#define UP_TRY(expr) do { const hipError_t _e = (expr); if (_e != hipSuccess) { up_err = std::string("HIP error in " #expr ": ") + hipGetErrorString(_e); return (int)kDeviceError; } } while (0)
    auto upload_lists = [&]() -> int {
        const auto t0 = std::chrono::steady_clock::now();
        UP_TRY(hipSetDevice(device_));
        {   // camera staging lists of the landmark-major kernels (ba_kernels.h, BAView::o_slot)
            const int64_t n_wg = (n_pt_ + kLmWg - 1) / kLmWg;
            raw_vector<uint8_t> slot(o_cam.size());
            std::vector<uint8_t> wn((size_t)std::max<int64_t>(n_wg, 1), 0);
            raw_vector<uint32_t> wlist((size_t)std::max<int64_t>(n_wg, 1) * kCamStageCap);
            parallel_ranges(n_wg, 64, [&](int64_t wb, int64_t we) {
                std::vector<int> where(n_cam_, -1), stamp(n_cam_, -1);
                for (int64_t w = wb; w < we; ++w) {
                    const int64_t l0 = w * kLmWg, l1 = std::min<int64_t>(n_pt_, l0 + kLmWg);
                    int n = 0;
                    uint32_t* list = wlist.data() + (size_t)w * kCamStageCap;
                    for (int64_t i = pt_ptr[l0]; i < pt_ptr[l1]; ++i) {
                        const uint32_t c = o_cam[i];
                        if (stamp[c] != (int)w) { stamp[c] = (int)w; where[c] = n < kCamStageCap ? n : 255; if (n < kCamStageCap) list[n++] = c; }
                        slot[i] = (uint8_t)where[c];
                    }
                    for (int k = n; k < kCamStageCap; ++k) list[k] = 0;
                    wn[w] = (uint8_t)n;
                }
            });
            static_assert(kCamStageCap <= 254, "slot 255 means not staged");
            UP_TRY(up(&o_slot_, slot));
            UP_TRY(up(&wg_cam_n_, wn));
            UP_TRY(up(&wg_cam_list_, wlist));
        }
        UP_TRY(up(&o_cam_, o_cam));
        UP_TRY(up(&o_pt_, o_pt));
        UP_TRY(up(&o_orig_, o_orig_h_));
        UP_TRY(up(&pt_ptr_, pt_ptr));
        UP_TRY(up(&cam_ptr_, cam_ptr));
        UP_TRY(up(&cam_obs_, cam_obs));
        UP_TRY(up(&co_rank_, co_rank));
        if (so.device_gathers) {
            // the caller's measurements go up as they are (one contiguous copy, no host gather), the three lists that are
            // permutations of what is on the device already are made there
            const size_t n_loc = o_cam.size();
            if (warmer.joinable()) warmer.join();   // (the measurements went up on the device thread, beside the list building)
            double* raw = raw_uv;
            hipError_t ge = hipSuccess;
            if (!raw) {
                UP_TRY(hipMalloc(reinterpret_cast<void**>(&raw_uv), std::max<size_t>(2 * (size_t)n_obs_, 2) * sizeof(double)));
                raw = raw_uv;
                ge = hipMemcpy(raw, obs_uv, 2 * (size_t)n_obs_ * sizeof(double), hipMemcpyHostToDevice);
            }
            // (plain allocations: every element is written by the gathers.  NOT the zero-filling alloc -- its hipMemset runs on the
            // null stream, which stream_ (non-blocking) does not follow: the clear could land AFTER the gather had written the
            // array; seen once in 36 problems of the population test as a step of garbage)
            auto fresh = [](auto** pp, size_t n) -> hipError_t {
                if (*pp) { (void)hipFree(*pp); *pp = nullptr; }
                return hipMalloc(reinterpret_cast<void**>(pp), std::max<size_t>(n, 1) * sizeof(**pp));
            };
            if (ge == hipSuccess) ge = fresh(reinterpret_cast<double**>(&o_uv_), 2 * n_loc);
            if (ge == hipSuccess) ge = fresh(reinterpret_cast<double**>(&co_uv_), 2 * n_loc);
            if (ge == hipSuccess) ge = fresh(&co_pt_, n_loc);
            if (ge == hipSuccess) {
                launch_gather_uv((int64_t)n_loc, o_orig_, raw, reinterpret_cast<double*>(o_uv_), stream_);
                launch_gather_uv((int64_t)n_loc, cam_obs_, reinterpret_cast<const double*>(o_uv_), reinterpret_cast<double*>(co_uv_), stream_);
                launch_gather_u32((int64_t)n_loc, cam_obs_, o_pt_, co_pt_, stream_);
                ge = hipStreamSynchronize(stream_);
            }
            (void)hipFree(raw_uv); raw_uv = nullptr;
            UP_TRY(ge);
        } else {
            UP_TRY(up(reinterpret_cast<double**>(&o_uv_), o_uv));
            UP_TRY(up(&co_pt_, co_pt));
            UP_TRY(up(reinterpret_cast<double**>(&co_uv_), co_uv));
        }
        {
            std::vector<uint8_t> fp(6 * n_cam_, 0), fi(3 * n_cam_, 0), fl(3 * n_pt_, 0);
            for (int64_t c = 0; c < n_cam_; ++c) {
                if (fix_pose) memcpy(fp.data() + 6 * (size_t)cmap_[c], fix_pose + 6 * c, 6);
                if (fix_intr) memcpy(fi.data() + 3 * (size_t)cmap_[c], fix_intr + 3 * c, 3);
            }
            if (fix_pt)
                for (int64_t l = 0; l < n_pt_; ++l) memcpy(fl.data() + 3 * (size_t)lmap_[l], fix_pt + 3 * l, 3);
            UP_TRY(up(&fix_pose_, fp));
            UP_TRY(up(&fix_intr_, fi));
            UP_TRY(up(&fix_pt_, fl));
        }
        for (int w = 0; w < 2; ++w) {
            UP_TRY(alloc(&poses_[w], 7 * n_cam_));
            UP_TRY(alloc(&intr_[w], 3 * n_cam_));
            UP_TRY(alloc(&pts_[w], 3 * n_pt_));
            UP_TRY(alloc(&camp_[w], (size_t)(kCamStride + kCamQStride) * n_cam_));   // [n_cam][16] records | [n_cam][10] compact form
        }
        UP_TRY(alloc(&g_c_, n_c_pad_));
        UP_TRY(alloc(&g_red_, n_c_pad_));
        UP_TRY(alloc(&dcam_, n_c_pad_));
        UP_TRY(alloc(&hinv_, (size_t)kLmStride * n_pt_));  // landmark records: Hll^-1 | g_l | point
        // projection records of the local observations (xn, yn, p_w.z, sqrt(rho')): the record form of the pair kernel
        UP_TRY(alloc(&orec_, 4 * (size_t)o_cam.size()));
        UP_TRY(alloc(&g_l_, 3 * n_pt_));
        UP_TRY(alloc(&dl_, 3 * n_pt_));
        UP_TRY(alloc(&partial_, 3 * (size_t)n_partial_));
        UP_TRY(alloc(&scal_, 32));
        UP_TRY(alloc(&pcg_buf_, 7 * (size_t)n_c_pad_));
        UP_TRY(alloc(&lmu_, (size_t)kLmuStride * n_pt_));
        UP_TRY(alloc(&sd_, (size_t)n_cam_ * dc_ * dc_));
        UP_TRY(alloc(&minv_, (size_t)n_cam_ * dc_ * dc_));
        if (flags_) hipFree(flags_);
        UP_TRY(dev_alloc(&flags_, 4));
        UP_TRY(hipMemset(flags_, 0, 4 * sizeof(int)));
        for (int b = 0; b < 2; ++b) {   // the pinned chunks of upload_staged: mapped here, not in the caller's first set_params
            if (!pin_[b]) UP_TRY(hipHostMalloc(&pin_[b], (size_t)16 << 20, hipHostMallocDefault));
            if (!pin_ev_[b]) UP_TRY(hipEventCreateWithFlags(&pin_ev_[b], hipEventDisableTiming));
        }
        up_seconds = since(t0);
        return kOk;
    };
#undef UP_TRY
    int up_rc = kOk;
    std::thread uploader([&] {   // (nothing may escape a thread: an allocation failure becomes this call's status)
        try { up_rc = upload_lists(); }
        catch (const std::exception& ex) { up_rc = kDeviceError; up_err = std::string("set_structure uploads: ") + ex.what(); }
    });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{uploader};

    // ---- the tile plan: built beside the lists above, or here (distributed plans with tree sharding) -------------
    if (plan_beside_lists) planner.join(); else build_plan();
    if (matrix_free_only_) so.schur_form = -1;
    if (!plan_err.empty()) { uploader.join(); return fail(kInvalidInput, "reduced camera matrix: " + plan_err); }
    hs.seconds[2] = plan_seconds;
    // ---- task lists of the selected form of the Schur reduction (they need the slot map) ------------------------------
    // (tried in round 5: the pair list built from a host-only twin's slot map BEFORE the planner thread is done -- the host pool
    // is shared, the three threads then slow one another down: set-up 0.26-0.28 s against 0.24)
    PairDeviceTables dtab;
    const bool recs_on_device = device_pair_recs_ && so.schur_form == 4 && dc_ == 9;
    hs.build_schur_lists(so, tp_.slot_host(), recs_on_device ? &dtab : nullptr);
    n_ptasks_ = (int)hs.pl.tasks.size();
    n_pair_blocks_ = hs.pl.n_blocks; pair_queued_ = hs.pl.queued;
    n_pair_slots_ = (recs_on_device && hs.pl.queued) ? dtab.n_slots : (int64_t)hs.pl.recs.size();
    const PairLists& pl = hs.pl;
    uploader.join();
    if (up_rc != kOk) return fail(up_rc, up_err);
    hs.release_scratch();
    tr.mark("plan + Schur lists (observation lists uploading beside them)");
    const auto t_up2 = std::chrono::steady_clock::now();
    HIP_TRY(up(&ptasks_, pl.tasks));
    HIP_TRY(up(&pchunks_, pl.chunks));
    HIP_TRY(up(&pblocks_, pl.blocks));
    if (recs_on_device && pl.queued) {
        // the records of the queued layout are written by the device from the observation lists the uploader put there
        // (schur_pairs.h, PairDeviceTables): 17 MB of tables up instead of 1.56 GB of records built and copied
        int *d_rows = nullptr, *d_run_ptr = nullptr, *d_run_piece0 = nullptr;
        uint32_t* d_run_cj = nullptr;
        int2 *d_piece = nullptr, *d_task = nullptr;
        if (precs_) { hipFree(precs_); precs_ = nullptr; }
        hipError_t e = dev_alloc(&precs_, (size_t)dtab.n_slots);
        if (e == hipSuccess) e = up(&d_rows, dtab.rows);
        if (e == hipSuccess) e = up(&d_run_ptr, dtab.run_ptr);
        if (e == hipSuccess) e = up(&d_run_cj, dtab.run_cj);
        if (e == hipSuccess) e = up(&d_run_piece0, dtab.run_piece0);
        if (e == hipSuccess) e = up(&d_piece, dtab.piece);
        if (e == hipSuccess) e = up(&d_task, dtab.task);
        if (e == hipSuccess)
            e = launch_build_pair_recs_q(n_cam_, d_rows, d_run_ptr, d_run_cj, d_run_piece0, d_piece, d_task, cam_ptr_, cam_obs_, o_pt_, pt_ptr_, o_cam_,
                                         precs_, dtab.n_slots, stream_);
        for (void* q : {(void*)d_rows, (void*)d_run_ptr, (void*)d_run_cj, (void*)d_run_piece0, (void*)d_piece, (void*)d_task})
            if (q) (void)hipFree(q);
        HIP_TRY(e);
    } else {
        HIP_TRY(up(&precs_, pl.recs));
    }
    if (pqdesc_) { hipFree(pqdesc_); pqdesc_ = nullptr; }
    if (pl.queued) HIP_TRY(up(&pqdesc_, pl.qdesc));
    up_seconds += since(t_up2);
    (void)t_up;

    HIP_TRY(hipDeviceSynchronize());  // the null-stream memsets above precede any work on stream_
    hs.seconds[4] = up_seconds;   // (copy time; the first part of it ran beside the plan and the pair list)
    hs.seconds[5] = since(t_begin);
    for (int k = 0; k < 6; ++k) setup_s_[k] = hs.seconds[k];
    tr.mark("set_structure body");
    // the host lists (3.7 GB on final-13682) are unmapped off the caller's path: 0.2 s
    if (free_thread_.joinable()) free_thread_.join();
    {
        const char* fm = getenv("APEX_SETUP_FREE");   // experiment switch: "sync" frees on the caller's path, "leak" never
        if (fm && !strcmp(fm, "sync")) hs_owner.reset();
        else if (fm && !strcmp(fm, "leak")) (void)hs_owner.release();
        else free_thread_ = std::thread([p = hs_owner.release()] { delete p; (void)HostBlockCache::get().end_setup(); });   // (what only an older, larger structure used goes back to the system)
    }
    tr.mark("host lists handed to the free thread");

    have_structure_ = true;
    have_params_ = have_step_ = have_trial_ = false;
    cur_ = 0;
    return kOk;
}

int Solver::set_params(const double* poses, const double* intr, const double* points) {
    if (!have_structure_) return fail(kInvalidState, "Block structure not built. Call set_structure() first.");
    HIP_TRY(hipSetDevice(device_));
    std::vector<double> hp(7 * n_cam_), hi(3 * n_cam_);
    for (int64_t c = 0; c < n_cam_; ++c) {
        memcpy(hp.data() + 7 * (size_t)cmap_[c], poses + 7 * c, 7 * sizeof(double));
        memcpy(hi.data() + 3 * (size_t)cmap_[c], intr + 3 * c, 3 * sizeof(double));
    }
    HIP_TRY(hipMemcpyAsync(poses_[cur_], hp.data(), 7 * n_cam_ * sizeof(double), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(intr_[cur_], hi.data(), 3 * n_cam_ * sizeof(double), hipMemcpyHostToDevice, stream_));
    std::vector<double> hpt;
    const double* src_pts = points;
    if (tree_shard_) {  // landmarks are renumbered so that every rank's set is one internal range
        hpt.resize(3 * n_pt_);
        for (int64_t l = 0; l < n_pt_; ++l) memcpy(hpt.data() + 3 * (size_t)lmap_[l], points + 3 * l, 3 * sizeof(double));
        src_pts = hpt.data();
    }
    { const int rc = upload_staged(pts_[cur_], src_pts, 3 * (size_t)n_pt_ * sizeof(double)); if (rc != kOk) return rc; }
    launch_prepare_cams(n_cam_, poses_[cur_], intr_[cur_], camp_[cur_], mode_mask(mode_), stream_);
    HIP_TRY(hipStreamSynchronize(stream_));
    have_params_ = true; have_step_ = have_trial_ = false; orec_fresh_ = false;
    return kOk;
}

int Solver::get_params(double* poses, double* intr, double* points) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    if (comm_ && world_ > 1) {
        // every rank owns a contiguous landmark range: gather the owners' points everywhere
        // (ranges differ in size, so one broadcast per owner)
        std::vector<int64_t> lo(world_ + 1);
        // all ranks compute identical cuts from the replicated structure: re-derive from pt_ptr is not
        // possible without the full lists, so exchange the range starts.
        int64_t mine[2] = {lm_lo_, lm_hi_};
        int64_t* d_rng = reinterpret_cast<int64_t*>(scal_ + 8);
        HIP_TRY(hipMemcpyAsync(d_rng + 2 * rank_, mine, sizeof mine, hipMemcpyHostToDevice, stream_));
        COMM_TRY(comm_->all_gather(d_rng + 2 * rank_, d_rng, 2 * sizeof(int64_t), stream_));
        std::vector<int64_t> rng(2 * world_);
        HIP_TRY(hipMemcpyAsync(rng.data(), d_rng, rng.size() * 8, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        for (int r = 0; r < world_; ++r) {
            const int64_t a = rng[2 * r], b = rng[2 * r + 1];
            if (b > a) COMM_TRY(comm_->broadcast(pts_[cur_] + 3 * a, 3 * (b - a), r, stream_));
        }
    }
    std::vector<double> hp(7 * n_cam_), hi(3 * n_cam_);
    HIP_TRY(hipMemcpyAsync(hp.data(), poses_[cur_], 7 * n_cam_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipMemcpyAsync(hi.data(), intr_[cur_], 3 * n_cam_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    std::vector<double> hpt(tree_shard_ ? 3 * n_pt_ : 0);
    HIP_TRY(hipMemcpyAsync(tree_shard_ ? hpt.data() : points, pts_[cur_], 3 * n_pt_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    if (tree_shard_)
        for (int64_t l = 0; l < n_pt_; ++l) memcpy(points + 3 * l, hpt.data() + 3 * (size_t)lmap_[l], 3 * sizeof(double));
    for (int64_t c = 0; c < n_cam_; ++c) {
        memcpy(poses + 7 * c, hp.data() + 7 * (size_t)cmap_[c], 7 * sizeof(double));
        memcpy(intr + 3 * c, hi.data() + 3 * (size_t)cmap_[c], 3 * sizeof(double));
    }
    return kOk;
}

// ---------------------------------------------------------------------------------------------
// cost (A16)
// ---------------------------------------------------------------------------------------------
int Solver::cost_of(int which, double* out) {
    stage_begin(kStCost);
    launch_cost(view(which), partial_, n_partial_, scal_, stream_);
    if (comm_ && world_ > 1)
        COMM_TRY(comm_->all_reduce_sum(scal_, 1, stream_));
    stage_end(kStCost);
    HIP_TRY(hipGetLastError());
    double ss = 0.0;
    HIP_TRY(hipMemcpyAsync(&ss, scal_, sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    const double nrm = sqrt(ss);  // compute_cost: 0.5 * norm_l2()^2 (optimizer/mod.rs:358-361)
    *out = 0.5 * nrm * nrm;
    return kOk;
}

int Solver::cost(double* out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    return cost_of(cur_, out);
}

// ---------------------------------------------------------------------------------------------
// assembly of S, g_red, Hll^-1, g (A6-A11) at the current parameters
// ---------------------------------------------------------------------------------------------
int Solver::assemble(double lambda, double diag_extra, bool for_factor) {
    if (matrix_free_only_) return fail(kInvalidState, "this handle was built matrix-free only (\"matrix_free_only\"): the explicit S does not exist");
    int rc = assemble_local(lambda, diag_extra, for_factor);
    if (rc != kOk) return rc;
    if (comm_ && world_ > 1) {
        stage_begin(kStAllReduce);
        // A distributed factorisation sums the shared top tiles itself, after the local levels, and a rank's local
        // levels read its own columns only: every column's tiles are reduced to their owner (tile_plan.h).
        const size_t te = (size_t)kNB * kNB;
        COMM_TRY(comm_->group_start());
        if (for_factor && tp_.distributed() && tree_shard_) {
            // tree sharding: a rank's landmarks are exactly those that touch its columns -- its tiles are complete
        } else if (for_factor && tp_.distributed()) {
            for (int o = 0; o < tp_.part_world(); ++o) {
                const std::pair<int64_t, int64_t> rg = tp_.owner_slot_range(o);
                if (rg.second > 0)
                    COMM_TRY(comm_->reduce_sum(tp_.tiles() + (size_t)rg.first * te, (size_t)rg.second * te, o, stream_));
            }
        } else {
            COMM_TRY(comm_->all_reduce_sum(tp_.tiles(), (size_t)tp_.n_touched_slots() * te, stream_));
        }
        COMM_TRY(comm_->all_reduce_sum(g_red_, (size_t)n_c_pad_, stream_));
        COMM_TRY(comm_->all_reduce_sum(g_c_, (size_t)n_c_pad_, stream_));
        COMM_TRY(comm_->all_reduce_max(flags_, 1, stream_));  // a singular landmark block anywhere fails the solve on every rank
        COMM_TRY(comm_->group_end());
        stage_end(kStAllReduce);
    }
    return assemble_finish();
}

// this rank's part of S, g_red, g_c (its landmarks), before any exchange
int Solver::assemble_local(double lambda, double diag_extra, bool for_factor) {
    const BAView v = view(cur_);
    const TileMap tm = tilemap();
    stage_begin(kStAssembleCam);
    // (a tree-sharded rank that assembles for its distributed factorisation adds to its own and the top tiles only; every
    // other use of S -- PCG, exports, the ladder's diagonal -- all-reduces every touched tile and needs them all cleared)
    HIP_TRY(tp_.zero_tiles(tree_shard_ && for_factor, nullptr, for_factor && world_ == 1));   // (the fill tiles stay as they are: tile_plan.h, first_writer_)
    launch_clear3(g_red_, g_c_, n_c_pad_, flags_, 4, stream_);   // (one launch instead of three fills)
    // identity on the padding rows of the last tile (rank 0 only: the all-reduce sums the ranks)
    // (tree sharding: by the owner of the last tile column, whose tiles are never summed -- pad_rank_)
    tp_.add_diag((int)n_c_, 0.0, rank_ == pad_rank_ ? 1.0 : 0.0);
    stage_end(kStAssembleCam);
    stage_begin(kStAssembleLm);
    launch_landmark_reduce(dc_, v, lambda, hinv_, g_l_, flags_, nullptr, stream_, orec_);   // (the pair kernel and the back-substitution read the projection records)
    orec_fresh_ = true;
    stage_end(kStAssembleLm);
    stage_begin(kStAssembleCam);
    launch_cam_reduce(dc_, v, tm, cam_ptr_, cam_obs_, lambda + diag_extra, rank_ == 0 ? 1 : 0, hinv_, g_l_, 1,
                      g_c_, g_red_, stream_);
    stage_end(kStAssembleCam);
    stage_begin(kStScatter);
    launch_schur_pairs(dc_, v, tp_.tiles(), ptasks_, n_ptasks_, pchunks_, pblocks_, precs_, hinv_, stream_, orec_, pqdesc_);
    stage_end(kStScatter);
    return check_hip(hipGetLastError(), "assembly kernels");
}

// after the exchange: the reduced system in the scaled variables when Jacobi scaling is on (linear, so it also
// commutes with the later sum of the top tiles of a distributed factorisation)
int Solver::assemble_finish() {
    if (scaled_) {  // the reduced system in the scaled variables: S := D_c S D_c, g_red := D_c g_red
        stage_begin(kStAssembleCam);
        tp_.scale_sym(cam_scale_);
        launch_vec_mul(n_c_pad_, g_red_, cam_scale_, g_red_, stream_);
        stage_end(kStAssembleCam);
    }
    return kOk;
}

int Solver::cholesky_attempt(int* failed_at) {
    stage_begin(kStFactor);
    const hipError_t fe = tp_.factor(failed_at);
    if (fe != hipSuccess && !comm_err_.empty()) return fail(kDeviceError, comm_err_);
    HIP_TRY(fe);
    stage_end(kStFactor);
    return kOk;
}

// cholesky_attempt on the S that assemble(lambda, reg, true) has just built.  Should the dataflow launch of the top groups
// time out (TilePlan::factor_flow_gave_up: the tiles are half updated and the plan has gone back to the level launches, in a
// distributed plan on every rank alike), S is built again and factorised once more.
int Solver::cholesky_on_fresh_s(double lambda, double reg, int* failed_at) {
    int rc = cholesky_attempt(failed_at);
    if (rc != kOk || !tp_.factor_flow_gave_up()) return rc;
    ++n_factor_flow_timeouts_;
    rc = assemble(lambda, reg, true);
    if (rc != kOk) return rc;
    rc = cholesky_attempt(failed_at);
    if (rc == kOk && tp_.factor_flow_gave_up()) return fail(kDeviceError, "dataflow factorisation timed out twice");
    return rc;
}

int Solver::tri_solve() {
    stage_begin(kStTriSolve);
    const hipError_t se = tp_.solve(g_red_, dcam_, pcg_buf_);
    if (se != hipSuccess && !comm_err_.empty()) return fail(kDeviceError, comm_err_);
    HIP_TRY(se);
    stage_end(kStTriSolve);
    return kOk;
}

// solve_with_cholesky (explicit_schur.rs:539-634) incl. the regularisation ladder
int Solver::factor_and_solve(double lambda) {
    int failed = 0;
    last_reg_ = 0.0;
    int rc = cholesky_on_fresh_s(lambda, 0.0, &failed);
    if (rc != kOk) return rc;
    if (!failed) return tri_solve();
    // the factorisation overwrote S: re-assemble it to read trace and max |diag| (:563-579)
    rc = assemble(lambda, 0.0);
    if (rc != kOk) return rc;
    double* diag = pcg_buf_;
    tp_.diag(diag);
    std::vector<double> hd(n_c_);
    HIP_TRY(hipMemcpyAsync(hd.data(), diag, n_c_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    // the reference's S also holds the 3 n_cam intrinsic rows when the factors do not touch them
    // (BundleAdjustment mode): their diagonal is lambda.
    double trace = 0.0, max_diag = 0.0;
    for (double d : hd) { trace += d; max_diag = std::max(max_diag, fabs(d)); }
    int64_t n_ref = n_c_;
    if (dc_ == 6) { trace += 3.0 * n_cam_ * lambda; max_diag = std::max(max_diag, fabs(lambda)); n_ref = 9 * n_cam_; }
    const double base = std::max(std::max(trace / (double)n_ref, max_diag), 1.0);
    for (int attempt = 0; attempt < 5; ++attempt) {
        const double reg = base * pow(10.0, (double)(attempt - 4));
        rc = assemble(lambda, reg, true);
        if (rc != kOk) return rc;
        rc = cholesky_on_fresh_s(lambda, reg, &failed);
        if (rc != kOk) return rc;
        if (!failed) { last_reg_ = reg; return tri_solve(); }
    }
    return fail(kSingularMatrix, "Schur complement singular after 5 regularization attempts (max reg = " + std::to_string(base) + ")");
}

// solve_with_pcg (explicit_schur.rs:639-756): Jacobi-preconditioned CG on the explicit S.
int Solver::pcg_solve() {
    stage_begin(kStFactor);
    HIP_TRY(tp_.pcg(g_red_, dcam_, pcg_buf_, cg_max_iter_, cg_tol_, &last_pcg_iters_));
    stage_end(kStFactor);
    return kOk;
}

// ---------------------------------------------------------------------------------------------
// A18: IterativeSchurSolver (src/linalg/sparse/implicit_schur.rs) -- S is never formed.
//   assemble_implicit : Hll^-1 / g_l (k_landmark_reduce), g_c and g_red plus the DIAGONAL blocks of S
//                       (k_cam_reduce with its self terms), Schur-Jacobi blocks inverted per variable
//   implicit_pcg_solve: solve_pcg_block (:577-679); every S p is two passes over the observations
//                       (landmark-major, then camera-major), nothing but 64 bytes per landmark in between
// Sharded: g_red, g_c and the diagonal blocks are all-reduced once, every S p once per iteration.
// ---------------------------------------------------------------------------------------------
int Solver::assemble_implicit(double lambda) {
    const BAView v = view(cur_);
    stage_begin(kStAssembleLm);
    HIP_TRY(hipMemsetAsync(flags_, 0, 4 * sizeof(int), stream_));
    launch_landmark_reduce(dc_, v, lambda, hinv_, g_l_, flags_, lmu_, stream_, orec_);
    orec_fresh_ = orec_ != nullptr;
    stage_end(kStAssembleLm);
    stage_begin(kStAssembleCam);
    launch_cam_reduce(dc_, v, tilemap(), cam_ptr_, cam_obs_, lambda, rank_ == 0 ? 1 : 0, hinv_, g_l_, 1, g_c_, g_red_, stream_);
    launch_extract_diag_blocks(dc_, n_cam_, tilemap(), sd_, stream_);
    stage_end(kStAssembleCam);
    if (comm_ && world_ > 1) {
        stage_begin(kStAllReduce);
        COMM_TRY(comm_->group_start());
        COMM_TRY(comm_->all_reduce_sum(sd_, (size_t)n_cam_ * dc_ * dc_, stream_));
        COMM_TRY(comm_->all_reduce_sum(g_red_, (size_t)n_c_pad_, stream_));
        COMM_TRY(comm_->all_reduce_sum(g_c_, (size_t)n_c_pad_, stream_));
        COMM_TRY(comm_->all_reduce_max(flags_, 1, stream_));  // a singular landmark block anywhere fails the solve on every rank
        COMM_TRY(comm_->group_end());
        stage_end(kStAllReduce);
    }
    stage_begin(kStAssembleCam);
    if (scaled_) {
        launch_scale_diag_blocks(dc_, n_cam_, cam_scale_, sd_, stream_);
        launch_vec_mul(n_c_pad_, g_red_, cam_scale_, g_red_, stream_);
    }
    launch_precond_blocks(dc_, n_cam_, sd_, minv_, stream_);
    stage_end(kStAssembleCam);
    return kOk;
}

// y = S x of the matrix-free operator (in the scaled variables when a scaling is set: D_c S0 D_c x, where S0 carries
// lambda / s^2 on its diagonal).  lam_local: lambda on rank 0, 0 elsewhere (the all-reduce sums the ranks' partial products).
int Solver::implicit_matvec(const double* x, double lam_local, double* y, bool reduce) {
    const double* xin = x;
    if (scaled_) {
        double* t = pcg_buf_ + 4 * n_c_pad_;
        launch_vec_mul(n_c_, x, cam_scale_, t, stream_);
        xin = t;
    }
    launch_implicit_matvec(dc_, view(cur_), cam_ptr_, hinv_, lmu_, xin, lam_local, y, stream_, backsub_records());
    if (reduce && comm_ && world_ > 1)
        COMM_TRY(comm_->all_reduce_sum(y, (size_t)n_c_, stream_));
    if (scaled_) launch_vec_mul(n_c_, y, cam_scale_, y, stream_);
    return check_hip(hipGetLastError(), "implicit_matvec");
}

int Solver::implicit_pcg_solve(double lambda, int max_iter, double tol) {
    stage_begin(kStFactor);
    const int n = (int)n_c_;
    double *x = dcam_, *r = pcg_buf_, *z = pcg_buf_ + n_c_pad_, *p = pcg_buf_ + 2 * n_c_pad_, *ap = pcg_buf_ + 3 * n_c_pad_;
    double* sc = scal_ + 16;   // [0] r.r  [1] r.z  [2] p.Ap  [4] rz_old  [5] frozen  [6] beta (chol_kernels.h)
    if (!pcg_host_) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pcg_host_), 16 * sizeof(double), hipHostMallocDefault));
        for (hipEvent_t& ev : pcg_ev_) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    HIP_TRY(hipMemsetAsync(x, 0, n_c_pad_ * sizeof(double), stream_));
    HIP_TRY(hipMemcpyAsync(r, g_red_, n_c_pad_ * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    launch_precond_apply(dc_, n_cam_, minv_, r, z, stream_);
    HIP_TRY(hipMemcpyAsync(p, z, n_c_pad_ * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    launch_dot2(n, r, z, r, r, partial_, n_partial_, sc, stream_);
    launch_pcg_implicit_begin(sc, stream_);   // rz_old := r.z, not frozen
    double* h = pcg_host_;
    HIP_TRY(hipMemcpyAsync(h, sc, 2 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    const double abs_tol = tol * std::max(sqrt(h[1]), 1.0);
    const double lam_local = (rank_ == 0) ? lambda : 0.0;  // the all-reduce sums the ranks' partial S p
    // Every scalar of the iteration stays on the device (alpha, beta, the reference's three termination tests:
    // k_pcg_implicit_close), and the host reads {r.r, r.z, p.Ap, frozen} ONE ITERATION BEHIND: iteration k + 1 is enqueued
    // before the host waits for iteration k, so the device never idles through a host round trip (round 5; two round trips per
    // iteration before).  An iteration enqueued behind a met test changes nothing: x, r, p and the iteration count are those of
    // the loop that waited every time.  (Sharded: every rank reads the same scalars and enqueues the same iterations.)
    auto enqueue_iteration = [&](int slot) -> int {
        const int mrc = implicit_matvec(p, lam_local, ap, true);
        if (mrc != kOk) return mrc;
        launch_dot2(n, p, ap, p, ap, partial_, n_partial_, sc + 2, stream_);
        launch_pcg_update_xr_sc(n, sc, p, ap, x, r, stream_);        // alpha = rz_old / p.Ap; nothing when |p.Ap| < 1e-20 (:610-613)
        launch_precond_apply(dc_, n_cam_, minv_, r, z, stream_);
        launch_dot2(n, r, r, r, z, partial_, n_partial_, sc, stream_);
        launch_pcg_implicit_close(sc, abs_tol, stream_);               // :634-641, :652-654, else beta and rz_old
        launch_pcg_update_p_sc(n, sc, z, p, stream_);
        HIP_TRY(hipMemcpyAsync(pcg_host_ + 8 * slot, sc, 6 * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipEventRecord(pcg_ev_[slot], stream_));
        return kOk;
    };
    int it = 0;
    if (max_iter > 0) { const int rc = enqueue_iteration(0); if (rc != kOk) return rc; }
    for (; it < max_iter; ++it) {
        if (it + 1 < max_iter) { const int rc = enqueue_iteration((it + 1) & 1); if (rc != kOk) return rc; }   // on speculation
        HIP_TRY(hipEventSynchronize(pcg_ev_[it & 1]));
        h = pcg_host_ + 8 * (it & 1);
        if (fabs(h[2]) < 1e-20) break;           // :610-613 (x, r untouched)
        if (h[5] != 0.0) { ++it; break; }        // the device's verdict: |r| < tol (:634-641) or rz_old ~ 0 (:652-654)
    }
    HIP_TRY(hipStreamSynchronize(stream_));      // (the speculative iteration, if any, has drained)
    last_pcg_iters_ = it;
    stage_end(kStFactor);
    return kOk;
}

int Solver::solve_augmented(double lambda, int variant, double* step_out, double* grad_out) {
    if (!have_params_) return fail(kInvalidState, "Block structure not built or parameters not set");
    HIP_TRY(hipSetDevice(device_));
    have_step_ = false;
    have_trial_ = false;   // (the eager step evaluation of this solve overwrites the trial parameter set: an earlier eval_step is void)
    last_pcg_iters_ = 0;   // (the PCG variants set it: apexgpu_info[5] is about THIS solve)
    ++step_serial_;
    last_lambda_ = lambda;
    int pcg_max = cg_max_iter_;
    double pcg_tol = cg_tol_;
    if (auto_fallback_ && variant != 2) {   // set_structure selected the matrix-free variant for this handle (set_auto_variant)
        if (variant == 0) { pcg_max = 500; pcg_tol = 1e-9; }   // IterativeSchurSolver::new (implicit_schur.rs:94-95)
        variant = 2;
    }
    int rc = (variant == 2) ? assemble_implicit(lambda) : assemble(lambda, 0.0, variant == 0);
    if (rc != kOk) return rc;
    // ONE host wait per Cholesky solve (round 5).  Until round 4 the host waited three times inside this call -- for the
    // landmark-inversion flag behind the assembly, for the pivot flag behind the factorisation, for the step at the end -- and
    // the GPU idled through a synchronisation plus a graph launch each time.  On a single rank the factorisation, the sweeps and
    // the back-substitution are now enqueued back to back and the two flags are read at the final wait; a singular landmark
    // block, a failed pivot or a dataflow time-out then takes the old path from the top (re-assembly, ladder), results unchanged.
    const bool one_wait = one_wait_ && variant == 0 && !(comm_ && world_ > 1) && !tp_.distributed();
    bool speculative = false;
    if (one_wait) {
        int failed = 0;
        last_reg_ = 0.0;
        stage_begin(kStFactor);
        const hipError_t fe = tp_.factor(&failed, /*defer_flags=*/true);
        HIP_TRY(fe);
        stage_end(kStFactor);
        rc = tri_solve();
        if (rc != kOk) return rc;
        speculative = true;
    } else {
    int lm_err = 0;
    HIP_TRY(hipMemcpyAsync(&lm_err, flags_, sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    if (lm_err) return fail(kSingularMatrix, "Landmark block is singular");
    rc = (variant == 2) ? implicit_pcg_solve(lambda, pcg_max, pcg_tol) : (variant == 1) ? pcg_solve() : factor_and_solve(lambda);
    if (rc != kOk) return rc;
    }
    for (int attempt = 0;; ++attempt) {
        stage_begin(kStBackSub);
        if (scaled_) launch_vec_mul(n_c_, dcam_, cam_scale_, dcam_, stream_);  // apply_inverse_scaling: dc = D_c y
        // what the LM loop asks next (step statistics, trial cost) rides on this solve's wait (solver.h, eager_eval_); the trial
        // POINTS are written by the back-substitution itself
        const bool eager = eager_eval_ && !(comm_ && world_ > 1);
        trial_pts_written_ = eager && fix_pt_ != nullptr;
        launch_back_substitute(dc_, view(cur_), hinv_, g_l_, dcam_, dl_, stream_, backsub_records(), trial_pts_written_ ? fix_pt_ : nullptr,
                               trial_pts_written_ ? pts_[cur_ ^ 1] : nullptr);
        stage_end(kStBackSub);
        HIP_TRY(hipGetLastError());
        have_step_ = true;
        if (eager) {
            if (!eager_host_) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&eager_host_), 8 * sizeof(double), hipHostMallocDefault));
            rc = enqueue_step_stats();
            if (rc == kOk) rc = enqueue_trial_point(scal_ + 6);
            if (rc != kOk) return rc;
            HIP_TRY(hipMemcpyAsync(eager_host_, scal_, 7 * sizeof(double), hipMemcpyDeviceToHost, stream_));
        }
        rc = export_step(step_out, grad_out);   // (synchronises: the sweeps' error word is on the host now)
        if (rc == kOk && speculative) {   // the flags the old path read before going on
            speculative = false;
            int lm_err = 0, failed = 0;
            HIP_TRY(hipMemcpyAsync(&lm_err, flags_, sizeof(int), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(tp_.read_flags(&failed));   // (synchronises; raises factor_flow_gave_up() on a dataflow time-out)
            if (lm_err) { have_step_ = false; return fail(kSingularMatrix, "Landmark block is singular"); }
            const bool gave_up = tp_.factor_flow_gave_up();
            if (failed || gave_up) {
                // what was enqueued behind the failed factorisation is void: S again, then the old path (ladder included)
                have_step_ = false;
                (void)tp_.sweep_timed_out();   // (clears the word a sweep over a broken factor may have raised)
                if (gave_up) ++n_factor_flow_timeouts_;
                rc = assemble(lambda, 0.0, true);
                if (rc != kOk) return rc;
                rc = factor_and_solve(lambda);
                if (rc != kOk) return rc;
                attempt = -1;   // (the loop's counter is for the sweep time-outs of the solve that follows)
                continue;       // back-substitution and export once more
            }
        }
        if (rc != kOk || variant != 0 || !tp_.sweep_timed_out()) {
            if (rc == kOk && eager) eager_serial_ = step_serial_;   // (the answers of THIS solve: step_stats / eval_step)
            return rc;
        }
        // A dataflow sweep of THIS solve ran into its spin limit (chol_kernels.hip, flow_wait): dcam_ is wrong.  The factor is
        // intact, so the solve is repeated with the level-by-level sweeps -- for this call and for the rest of the plan's
        // life (a device that starved a sweep once will do it again, and every time-out costs ~2 s).  In a distributed plan
        // the word was max-reduced: every rank is here.
        have_step_ = false;
        if (attempt > 0 || !tp_.tri_flow()) return fail(kDeviceError, "triangular sweep timed out");
        tp_.enable_tri_flow(false);
        rc = tri_solve();
        if (rc != kOk) return rc;
    }
}

// the last step / gradient in the reference's global column order (syncs)
int Solver::export_step(double* step_out, double* grad_out) {
    if (step_out || grad_out) {
        std::vector<double> hc(n_c_), hl(3 * n_pt_);
        for (int pass = 0; pass < 2; ++pass) {
            double* out = pass == 0 ? step_out : grad_out;
            if (!out) continue;
            HIP_TRY(hipMemcpyAsync(hc.data(), pass == 0 ? dcam_ : g_c_, n_c_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipMemcpyAsync(hl.data(), pass == 0 ? dl_ : g_l_, 3 * n_pt_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipStreamSynchronize(stream_));
            if (scaled_) {  // the caller's variables are the scaled ones: y = step / s, gradient = s g
                for (int64_t i = 0; i < n_c_; ++i) hc[i] = pass == 0 ? hc[i] / cam_scale_h_[i] : hc[i] * cam_scale_h_[i];
                for (int64_t i = 0; i < 3 * n_pt_; ++i) hl[i] = pass == 0 ? hl[i] / pt_scale_h_[i] : hl[i] * pt_scale_h_[i];
            }
            for (int64_t c = 0; c < n_cam_; ++c) {
                const int64_t ci = cmap_[c];
                for (int a = 0; a < 6; ++a) out[pose_col_[c] + a] = hc[ci * dc_ + a];
                for (int a = 0; a < 3; ++a) out[intr_col_[c] + a] = (dc_ == 9) ? hc[ci * dc_ + 6 + a] : 0.0;
            }
            for (int64_t l = 0; l < n_pt_; ++l)
                for (int a = 0; a < 3; ++a) out[pt_col_[l] + a] = hl[3 * (size_t)lmap_[l] + a];
        }
    } else {
        HIP_TRY(hipStreamSynchronize(stream_));
    }
    return kOk;
}

// ---------------------------------------------------------------------------------------------
// The distributed Cholesky solve cut into its phases, so that a test can drive the `world` instances of a
// sharded problem in lockstep inside ONE process and play the communicator itself (capi: apexgpu_debug_lockstep_solve).
// The production path (solve_augmented with an RCCL communicator) runs exactly these pieces with ncclAllReduce on the
// same buffers in between.  Exchange point p follows phase p:
//   0: each rank's column tiles (reduce to the owner), g_red, g_c   1: the top tile ranges   2: the failure flag (max)
//   3, 4: TilePlan's exchange vector
// ---------------------------------------------------------------------------------------------
int Solver::dist_phase(int phase, double lambda) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    if (!tp_.distributed()) return fail(kInvalidState, "the plan is not distributed (set_shard with world > 1, dist_factor on)");
    if (matrix_free_only_) return fail(kInvalidState, "this handle was built matrix-free only (\"matrix_free_only\"): the explicit S does not exist");
    HIP_TRY(hipSetDevice(device_));
    switch (phase) {
        case 0: have_step_ = false; last_lambda_ = lambda; return assemble_local(lambda, 0.0, true);
        case 1: {
            int rc = assemble_finish();
            if (rc != kOk) return rc;
            stage_begin(kStFactor);
            tp_.factor_phase(0);
            stage_end(kStFactor);
            return kOk;
        }
        case 2:
            stage_begin(kStAllReduce);   // lockstep runs only: the replicated top levels are booked under this stage's
            tp_.factor_phase(1);         // name so that they can be told apart from the local levels
            stage_end(kStAllReduce);
            return kOk;
        case 3: {
            int f[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(&f[0], tp_.flag_dev(), sizeof(int), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipMemcpyAsync(&f[1], flags_, sizeof(int), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipStreamSynchronize(stream_));
            if (f[1]) return fail(kSingularMatrix, "Landmark block is singular");
            if (f[0]) return fail(kFactorizationFailed, "non-positive pivot in tile column " + std::to_string(f[0] - 1));
            stage_begin(kStTriSolve);
            tp_.solve_phase(0, g_red_, dcam_, pcg_buf_);
            stage_end(kStTriSolve);
            return kOk;
        }
        case 4:
            stage_begin(kStTriSolve);
            tp_.solve_phase(1, g_red_, dcam_, pcg_buf_);
            stage_end(kStTriSolve);
            return kOk;
        case 5:
            tp_.solve_phase(2, g_red_, dcam_, pcg_buf_);
            if (scaled_) launch_vec_mul(n_c_, dcam_, cam_scale_, dcam_, stream_);
            launch_back_substitute(dc_, view(cur_), hinv_, g_l_, dcam_, dl_, stream_, backsub_records());
            HIP_TRY(hipStreamSynchronize(stream_));
            have_step_ = true;
            return kOk;
        default: return fail(kInvalidInput, "phase out of range");
    }
}

void Solver::dist_buffers(int point, std::vector<DistBuf>* sums, int** max_flag) {
    const size_t te = (size_t)kNB * kNB;
    sums->clear(); *max_flag = nullptr;
    if (point == 0) {
        for (int o = 0; o < tp_.part_world() && !tree_shard_; ++o) {   // every column's tiles are reduced to their owner
            const std::pair<int64_t, int64_t> rg = tp_.owner_slot_range(o);
            sums->push_back({tp_.tiles() + (size_t)rg.first * te, (size_t)rg.second * te, o});
        }
        sums->push_back({g_red_, (size_t)n_c_pad_, -1});
        sums->push_back({g_c_, (size_t)n_c_pad_, -1});
    } else if (point == 1) {
        std::pair<int64_t, int64_t> rg[2];
        tp_.top_slot_ranges(rg);
        for (int i = 0; i < 2; ++i)
            if (rg[i].second > 0) sums->push_back({tp_.tiles() + (size_t)rg[i].first * te, (size_t)rg[i].second * te, -1});
    } else if (point == 2) {
        *max_flag = tp_.flag_dev();
    } else if (point == 3 || point == 4) {
        sums->push_back({tp_.exch_buffer(), (size_t)tp_.n_pad(), -1});
    }
}

int Solver::assemble_only(double lambda) {
    if (!have_params_) return fail(kInvalidState, "Block structure not built or parameters not set");
    HIP_TRY(hipSetDevice(device_));
    have_step_ = false;
    last_lambda_ = lambda;
    int rc = assemble(lambda, 0.0);
    if (rc != kOk) return rc;
    int lm_err = 0;
    HIP_TRY(hipMemcpyAsync(&lm_err, flags_, sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    if (lm_err) return fail(kSingularMatrix, "Landmark block is singular");
    return kOk;
}

int Solver::enqueue_step_stats() {
    stage_begin(kStStats);
    launch_step_stats(n_c_, g_c_, dcam_, last_lambda_, scaled_ ? cam_scale_ : nullptr, partial_, n_partial_, scal_, stream_);
    launch_step_stats(3 * n_pt_, g_l_, dl_, last_lambda_, scaled_ ? pt_scale_ : nullptr, partial_, n_partial_, scal_ + 3, stream_);
    if (comm_ && world_ > 1)  // landmark part is sharded, camera part replicated
        COMM_TRY(comm_->all_reduce_sum(scal_ + 3, 3, stream_));
    stage_end(kStStats);
    return kOk;
}
static void stats_from_sums(const double h[6], double out3[3]) {
    out3[0] = sqrt(h[0] + h[3]);          // gradient.norm_l2()          (levenberg_marquardt.rs:746)
    out3[1] = sqrt(h[1] + h[4]);          // step.norm_l2()              (:890)
    out3[2] = 0.5 * (h[2] + h[5]);        // compute_predicted_reduction (:721-727)
}
int Solver::step_stats(double out3[3]) {
    if (!have_step_) return fail(kInvalidState, "no step computed");
    if (eager_serial_ == step_serial_ && eager_host_) { stats_from_sums(eager_host_, out3); return kOk; }   // read at the solve's wait
    HIP_TRY(hipSetDevice(device_));
    const int rc = enqueue_step_stats();
    if (rc != kOk) return rc;
    double h[6];
    HIP_TRY(hipMemcpyAsync(h, scal_, sizeof h, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    stats_from_sums(h, out3);
    return kOk;
}
// the trial point x (+) step in the other parameter set and the sum of squared corrected residuals there (device scalar)
int Solver::enqueue_trial_point(double* sumsq_out) {
    const int t = cur_ ^ 1;
    stage_begin(kStRetract);
    launch_retract(dc_, n_cam_, trial_pts_written_ ? 0 : n_pt_, poses_[cur_], intr_[cur_], pts_[cur_], dcam_, dl_, 1.0, fix_pose_, fix_intr_,
                   fix_pt_, poses_[t], intr_[t], pts_[t], stream_);   // (the points: by k_back_substitute when trial_pts_written_)
    trial_pts_written_ = false;
    launch_prepare_cams(n_cam_, poses_[t], intr_[t], camp_[t], mode_mask(mode_), stream_);
    stage_end(kStRetract);
    stage_begin(kStCost);
    launch_cost(view(t), partial_, n_partial_, sumsq_out, stream_);
    if (comm_ && world_ > 1)
        COMM_TRY(comm_->all_reduce_sum(sumsq_out, 1, stream_));
    stage_end(kStCost);
    return kOk;
}
int Solver::eval_step(double* trial_cost) {
    if (!have_step_) return fail(kInvalidState, "no step computed");
    if (eager_serial_ == step_serial_ && eager_host_) {   // the trial point is in place and its cost was read at the solve's wait
        have_trial_ = true;
        const double nrm = sqrt(eager_host_[6]);
        *trial_cost = 0.5 * nrm * nrm;
        return kOk;
    }
    HIP_TRY(hipSetDevice(device_));
    const int t = cur_ ^ 1;
    stage_begin(kStRetract);
    launch_retract(dc_, n_cam_, n_pt_, poses_[cur_], intr_[cur_], pts_[cur_], dcam_, dl_, 1.0, fix_pose_, fix_intr_,
                   fix_pt_, poses_[t], intr_[t], pts_[t], stream_);
    launch_prepare_cams(n_cam_, poses_[t], intr_[t], camp_[t], mode_mask(mode_), stream_);
    stage_end(kStRetract);
    have_trial_ = true;
    return cost_of(t, trial_cost);
}
int Solver::commit_step() {
    if (!have_trial_) return fail(kInvalidState, "no trial point");
    cur_ ^= 1;
    have_trial_ = false; have_step_ = false; orec_fresh_ = false;
    return kOk;
}

// apply_negative_parameter_step (optimizer/mod.rs:343-356): the rejected trial point is moved back
// by the inverse retraction, it is NOT restored from a snapshot.
int Solver::discard_step() {
    if (!have_trial_) return fail(kInvalidState, "no trial point");
    HIP_TRY(hipSetDevice(device_));
    const int t = cur_ ^ 1;
    stage_begin(kStRetract);
    launch_retract(dc_, n_cam_, n_pt_, poses_[t], intr_[t], pts_[t], dcam_, dl_, -1.0, fix_pose_, fix_intr_, fix_pt_,
                   poses_[cur_], intr_[cur_], pts_[cur_], stream_);
    launch_prepare_cams(n_cam_, poses_[cur_], intr_[cur_], camp_[cur_], mode_mask(mode_), stream_);
    stage_end(kStRetract);
    HIP_TRY(hipStreamSynchronize(stream_));
    have_trial_ = false; have_step_ = false; orec_fresh_ = false;
    return kOk;
}

int Solver::parameter_norm(double* out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    launch_sumsq(7 * n_cam_, poses_[cur_], partial_, n_partial_, scal_ + 9, stream_);
    launch_sumsq(3 * n_cam_, intr_[cur_], partial_, n_partial_, scal_ + 10, stream_);
    // points: in a sharded run only the owned range is current on this rank
    launch_sumsq(3 * (lm_hi_ - lm_lo_), pts_[cur_] + 3 * lm_lo_, partial_, n_partial_, scal_ + 11, stream_);
    if (comm_ && world_ > 1)
        COMM_TRY(comm_->all_reduce_sum(scal_ + 11, 1, stream_));
    double h[3];
    HIP_TRY(hipMemcpyAsync(h, scal_ + 9, sizeof h, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    *out = sqrt(h[0] + h[1] + h[2]);
    return kOk;
}

// ---------------------------------------------------------------------------------------------
// Jacobi column scaling (process_jacobian_generic, optimizer/mod.rs:749-763)
// ---------------------------------------------------------------------------------------------
int Solver::ensure_scale_buffers() {
    if (cam_scale_) return kOk;
    HIP_TRY(dev_alloc(&cam_scale_, (size_t)n_c_pad_));
    HIP_TRY(dev_alloc(&pt_scale_, (size_t)std::max<int64_t>(3 * n_pt_, 1)));
    return kOk;
}

// squared column norms of the corrected Jacobian at the current parameters, left in cam_scale_ / pt_scale_
int Solver::column_norms_sq_device() {
    int rc = ensure_scale_buffers();
    if (rc != kOk) return rc;
    const bool was = scaled_;
    scaled_ = false;
    const BAView v = view(cur_);
    scaled_ = was;
    HIP_TRY(hipMemsetAsync(cam_scale_, 0, n_c_pad_ * sizeof(double), stream_));
    HIP_TRY(hipMemsetAsync(pt_scale_, 0, std::max<int64_t>(3 * n_pt_, 1) * sizeof(double), stream_));
    launch_column_norms_sq(dc_, v, cam_scale_, pt_scale_, stream_);
    if (comm_ && world_ > 1)  // every rank sees all cameras but only its own landmarks
        COMM_TRY(comm_->all_reduce_sum(cam_scale_, (size_t)n_c_, stream_));
    return kOk;
}

int Solver::column_norms(double* norms_out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    const bool was = scaled_;
    std::vector<double> keep_c, keep_p;
    if (was) { keep_c = cam_scale_h_; keep_p = pt_scale_h_; }
    int rc = column_norms_sq_device();
    if (rc != kOk) return rc;
    std::vector<double> hc(n_c_), hl(3 * n_pt_);
    HIP_TRY(hipMemcpyAsync(hc.data(), cam_scale_, n_c_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipMemcpyAsync(hl.data(), pt_scale_, 3 * n_pt_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    for (int64_t c = 0; c < n_cam_; ++c) {
        const int64_t ci = cmap_[c];
        for (int a = 0; a < 6; ++a) norms_out[pose_col_[c] + a] = sqrt(hc[ci * dc_ + a]);
        for (int a = 0; a < 3; ++a) norms_out[intr_col_[c] + a] = (dc_ == 9) ? sqrt(hc[ci * dc_ + 6 + a]) : 0.0;
    }
    for (int64_t l = 0; l < n_pt_; ++l)
        for (int a = 0; a < 3; ++a) norms_out[pt_col_[l] + a] = sqrt(hl[3 * (size_t)lmap_[l] + a]);
    if (was) {  // the buffers held the active scaling: put it back
        std::vector<double> pad(n_c_pad_, 1.0);
        std::copy(keep_c.begin(), keep_c.end(), pad.begin());
        HIP_TRY(hipMemcpyAsync(cam_scale_, pad.data(), n_c_pad_ * sizeof(double), hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipMemcpyAsync(pt_scale_, keep_p.data(), 3 * n_pt_ * sizeof(double), hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
    }
    return kOk;
}

int Solver::set_column_scaling(const double* scaling) {
    if (!have_structure_) return fail(kInvalidState, "Block structure not built");
    HIP_TRY(hipSetDevice(device_));
    have_step_ = false;
    if (!scaling) { scaled_ = false; return kOk; }
    int rc = ensure_scale_buffers();
    if (rc != kOk) return rc;
    cam_scale_h_.assign(n_c_, 1.0);
    pt_scale_h_.assign(3 * n_pt_, 1.0);
    for (int64_t c = 0; c < n_cam_; ++c) {
        const int64_t ci = cmap_[c];
        for (int a = 0; a < 6; ++a) cam_scale_h_[ci * dc_ + a] = scaling[pose_col_[c] + a];
        if (dc_ == 9)
            for (int a = 0; a < 3; ++a) cam_scale_h_[ci * dc_ + 6 + a] = scaling[intr_col_[c] + a];
    }
    for (int64_t l = 0; l < n_pt_; ++l)
        for (int a = 0; a < 3; ++a) pt_scale_h_[3 * (size_t)lmap_[l] + a] = scaling[pt_col_[l] + a];
    for (double v : cam_scale_h_) if (!(v > 0.0) || !std::isfinite(v)) return fail(kInvalidInput, "column scaling must be positive and finite");
    for (double v : pt_scale_h_) if (!(v > 0.0) || !std::isfinite(v)) return fail(kInvalidInput, "column scaling must be positive and finite");
    std::vector<double> pad(n_c_pad_, 1.0);
    std::copy(cam_scale_h_.begin(), cam_scale_h_.end(), pad.begin());
    HIP_TRY(hipMemcpyAsync(cam_scale_, pad.data(), n_c_pad_ * sizeof(double), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(pt_scale_, pt_scale_h_.data(), 3 * n_pt_ * sizeof(double), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    scaled_ = true;
    return kOk;
}

// iteration 0 of the reference's loop: norms of the current Jacobian -> s = 1 / (1 + norm), kept for the whole optimize
int Solver::set_jacobi_scaling(bool on) {
    if (!on) { scaled_ = false; have_step_ = false; return kOk; }
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    int rc = column_norms_sq_device();
    if (rc != kOk) return rc;
    launch_scaling_from_norms_sq(n_c_pad_, cam_scale_, cam_scale_, stream_);  // padding: n2 = 0 -> 1
    launch_scaling_from_norms_sq(3 * n_pt_, pt_scale_, pt_scale_, stream_);
    cam_scale_h_.resize(n_c_); pt_scale_h_.resize(3 * n_pt_);
    HIP_TRY(hipMemcpyAsync(cam_scale_h_.data(), cam_scale_, n_c_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipMemcpyAsync(pt_scale_h_.data(), pt_scale_, 3 * n_pt_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    scaled_ = true; have_step_ = false;
    return kOk;
}

// ---------------------------------------------------------------------------------------------
// The LM loop (optimize_with_mode, levenberg_marquardt.rs:823-1031) with the state on the device.
// ---------------------------------------------------------------------------------------------
int Solver::lm_optimize(LmConfig* cfg, LmResult* res, LmIterRecord* hist, int hist_cap) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    return run_lm(*this, cfg, res, hist, hist_cap);
}

// ---------------------------------------------------------------------------------------------
// parity / debug exports
// ---------------------------------------------------------------------------------------------
int Solver::get_residual(double* r_out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    double* d = nullptr;
    HIP_TRY(dev_alloc(&d, 2 * n_obs_));
    hipMemsetAsync(d, 0, 2 * n_obs_ * sizeof(double), stream_);
    launch_export_linearization(dc_, view(cur_), o_orig_, d, nullptr, nullptr, stream_);
    hipError_t e = hipMemcpyAsync(r_out, d, 2 * n_obs_ * sizeof(double), hipMemcpyDeviceToHost, stream_);
    hipStreamSynchronize(stream_);
    hipFree(d);
    return check_hip(e, "get_residual");
}

int Solver::get_jacobian_blocks(double* jc_out, double* jl_out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    double *dj = nullptr, *dl = nullptr;
    HIP_TRY(dev_alloc(&dj, 2 * dc_ * n_obs_));
    HIP_TRY(dev_alloc(&dl, 6 * n_obs_));
    hipMemsetAsync(dj, 0, 2 * dc_ * n_obs_ * sizeof(double), stream_);
    hipMemsetAsync(dl, 0, 6 * n_obs_ * sizeof(double), stream_);
    launch_export_linearization(dc_, view(cur_), o_orig_, nullptr, dj, dl, stream_);
    hipError_t e1 = hipMemcpyAsync(jc_out, dj, 2 * dc_ * n_obs_ * sizeof(double), hipMemcpyDeviceToHost, stream_);
    hipError_t e2 = hipMemcpyAsync(jl_out, dl, 6 * n_obs_ * sizeof(double), hipMemcpyDeviceToHost, stream_);
    hipStreamSynchronize(stream_);
    hipFree(dj); hipFree(dl);
    int rc = check_hip(e1, "get_jacobian_blocks");
    return rc != kOk ? rc : check_hip(e2, "get_jacobian_blocks");
}

// Dense S (9 n_cam square, row-major) and g_red in the reference's camera-side column order,
// re-assembled at the current parameters with the last lambda (the factorisation works in place).
int Solver::get_schur(double* S_out, double* gred_out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    int rc = assemble(last_lambda_, 0.0);
    if (rc != kOk) return rc;
    const size_t tile_elems = (size_t)kNB * kNB;
    const int64_t nref = 9 * n_cam_;
    auto ref_row = [&](int64_t i) -> int64_t {
        const int64_t ci = i / dc_; const int a = (int)(i - ci * dc_);
        const int64_t c = cinv_[ci];
        return a < 6 ? pose_col_[c] + a : intr_col_[c] + (a - 6);
    };
    if (gred_out) {
        std::vector<double> h(n_c_);
        HIP_TRY(hipMemcpyAsync(h.data(), g_red_, n_c_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        std::fill(gred_out, gred_out + nref, 0.0);
        for (int64_t i = 0; i < n_c_; ++i) gred_out[ref_row(i)] = h[i];
    }
    if (S_out) {
        std::fill(S_out, S_out + nref * nref, 0.0);
        if (dc_ == 6)  // intrinsic variables exist but no factor touches them: S_ii = lambda
            for (int64_t c = 0; c < n_cam_; ++c)
                for (int a = 0; a < 3; ++a) S_out[(intr_col_[c] + a) * nref + intr_col_[c] + a] = last_lambda_;
        std::vector<double> t(tile_elems);
        for (int I = 0; I < nt_; ++I)
            for (int J = 0; J <= I; ++J) {
                const int s = tp_.slot(I, J);
                if (s < 0) continue;
                HIP_TRY(hipMemcpyAsync(t.data(), tp_.tiles() + (size_t)s * tile_elems, tile_elems * sizeof(double), hipMemcpyDeviceToHost, stream_));
                HIP_TRY(hipStreamSynchronize(stream_));
                for (int r = 0; r < kNB; ++r)
                    for (int c = 0; c < kNB; ++c) {
                        const int64_t gi = (int64_t)I * kNB + r, gj = (int64_t)J * kNB + c;
                        if (gi >= n_c_ || gj >= n_c_ || gj > gi) continue;
                        const double val = t[(size_t)r * kNB + c];
                        const int64_t ri = ref_row(gi), rj = ref_row(gj);
                        S_out[ri * nref + rj] = val;
                        S_out[rj * nref + ri] = val;
                    }
            }
    }
    return kOk;
}

// Parity export: y = S x through BOTH implementations of the reduced camera matrix -- the explicit tiles
// (k_cam_reduce + k_schur_rows, multiplied by the symmetric tile product) and the matrix-free operator of
// the implicit variant -- for the same lambda.  x and the outputs are in the reference's camera-side column
// order (9 n_cam entries).  Two independent code paths that must agree at any problem size.
int Solver::schur_matvec(double lambda, const double* x_in, double* y_explicit, double* y_implicit) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    have_step_ = false;
    const int64_t nref = 9 * n_cam_;
    auto ref_row = [&](int64_t i) -> int64_t {
        const int64_t ci = i / dc_; const int a = (int)(i - ci * dc_);
        const int64_t c = cinv_[ci];
        return a < 6 ? pose_col_[c] + a : intr_col_[c] + (a - 6);
    };
    std::vector<double> h(n_c_pad_, 0.0);
    for (int64_t i = 0; i < n_c_; ++i) h[i] = x_in[ref_row(i)];
    double *xd = pcg_buf_, *yd = pcg_buf_ + n_c_pad_;
    HIP_TRY(hipMemcpyAsync(xd, h.data(), n_c_pad_ * sizeof(double), hipMemcpyHostToDevice, stream_));
    for (int pass = 0; pass < 2; ++pass) {
        double* out = pass == 0 ? y_explicit : y_implicit;
        if (!out) continue;
        int rc = pass == 0 ? assemble(lambda, 0.0) : assemble_implicit(lambda);
        if (rc != kOk) return rc;
        if (pass == 0) tp_.sym_matvec(xd, yd);
        else implicit_matvec(xd, rank_ == 0 ? lambda : 0.0, yd, false);  // a shard's partial
        HIP_TRY(hipMemcpyAsync(h.data(), yd, n_c_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        std::fill(out, out + nref, 0.0);
        for (int64_t i = 0; i < n_c_; ++i) out[ref_row(i)] = h[i];
        if (dc_ == 6 && rank_ == 0)  // intrinsic variables exist but no factor touches them: S_ii = lambda
            for (int64_t c = 0; c < n_cam_; ++c)
                for (int a = 0; a < 3; ++a) out[intr_col_[c] + a] = lambda * x_in[intr_col_[c] + a];
    }
    return kOk;
}

// get_hessian (explicit_schur.rs:1236-1238): H = J^T J, undamped, full symmetric, CSC in the global column order.  The
// device never forms it; this export rebuilds it on the host from the per-factor blocks the device linearises (an
// observer / DogLeg path, not the hot path): structural entries of every factor are kept even when a block is zero
// (a point behind its camera), as the reference's sparse product keeps them.
int Solver::get_hessian_csc(int64_t* nnz_out, int64_t* colptr, int64_t* rowidx, double* values) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    if (world_ > 1) return fail(kInvalidState, "the Hessian export is single-rank");
    if (!nnz_out) return fail(kInvalidInput, "nnz_out is NULL");
    const int64_t total = 9 * n_cam_ + 3 * n_pt_;
    // the caller's factor list, rebuilt from the device's landmark-major lists (single rank: they hold every observation):
    // observation k of the device is the caller's o_orig[k], its camera / landmark the internal ones mapped back
    std::vector<uint32_t> cam_idx_h_(n_obs_), pt_idx_h_(n_obs_);
    {
        HIP_TRY(hipSetDevice(device_));
        if ((int64_t)o_orig_h_.size() != n_obs_) return fail(kInvalidState, "the Hessian export needs every observation on this rank");
        std::vector<uint32_t> oc(n_obs_), op(n_obs_);
        HIP_TRY(hipMemcpyAsync(oc.data(), o_cam_, n_obs_ * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipMemcpyAsync(op.data(), o_pt_, n_obs_ * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        std::vector<int> linv(n_pt_);
        for (int64_t l = 0; l < n_pt_; ++l) linv[lmap_[l]] = (int)l;
        for (int64_t k = 0; k < n_obs_; ++k) {
            const int64_t i = o_orig_h_[k];
            cam_idx_h_[i] = (uint32_t)cinv_[oc[k]];
            pt_idx_h_[i] = (uint32_t)linv[op[k]];
        }
    }
    // unique (camera, landmark) couplings and the variables that carry entries
    std::vector<int64_t> order(n_obs_);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        return pt_idx_h_[a] != pt_idx_h_[b] ? pt_idx_h_[a] < pt_idx_h_[b] : cam_idx_h_[a] < cam_idx_h_[b];
    });
    std::vector<uint8_t> cam_seen(n_cam_, 0), pt_seen(n_pt_, 0);
    int64_t n_cl = 0;
    for (int64_t k = 0; k < n_obs_; ++k) {
        const int64_t i = order[k];
        cam_seen[cam_idx_h_[i]] = 1; pt_seen[pt_idx_h_[i]] = 1;
        if (k == 0 || pt_idx_h_[i] != pt_idx_h_[order[k - 1]] || cam_idx_h_[i] != cam_idx_h_[order[k - 1]]) ++n_cl;
    }
    int64_t nnz = 2 * n_cl * dc_ * 3;
    for (int64_t c = 0; c < n_cam_; ++c) if (cam_seen[c]) nnz += dc_ * dc_;
    for (int64_t l = 0; l < n_pt_; ++l) if (pt_seen[l]) nnz += 9;
    *nnz_out = nnz;
    if (!colptr) return kOk;
    if (!rowidx || !values) return fail(kInvalidInput, "rowidx / values are NULL");
    std::vector<double> jc((size_t)2 * dc_ * n_obs_), jl((size_t)6 * n_obs_);
    int rc = get_jacobian_blocks(jc.data(), jl.data());
    if (rc != kOk) return rc;
    auto cam_col = [&](int64_t c, int a) -> int64_t { return a < 6 ? pose_col_[c] + a : intr_col_[c] + (a - 6); };
    struct Trip { int64_t col, row; double v; };
    std::vector<Trip> t;
    t.reserve((size_t)nnz);
    {   // camera blocks
        std::vector<double> hcc((size_t)n_cam_ * dc_ * dc_, 0.0);
        for (int64_t i = 0; i < n_obs_; ++i) {
            const double* J = jc.data() + (size_t)2 * dc_ * i;
            double* H = hcc.data() + (size_t)cam_idx_h_[i] * dc_ * dc_;
            for (int a = 0; a < dc_; ++a)
                for (int b = 0; b < dc_; ++b) H[a * dc_ + b] += J[a] * J[b] + J[dc_ + a] * J[dc_ + b];
        }
        for (int64_t c = 0; c < n_cam_; ++c)
            if (cam_seen[c])
                for (int a = 0; a < dc_; ++a)
                    for (int b = 0; b < dc_; ++b) t.push_back({cam_col(c, b), cam_col(c, a), hcc[(size_t)c * dc_ * dc_ + a * dc_ + b]});
    }
    {   // landmark blocks
        std::vector<double> hll((size_t)n_pt_ * 9, 0.0);
        for (int64_t i = 0; i < n_obs_; ++i) {
            const double* J = jl.data() + (size_t)6 * i;
            double* H = hll.data() + (size_t)pt_idx_h_[i] * 9;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) H[3 * a + b] += J[a] * J[b] + J[3 + a] * J[3 + b];
        }
        for (int64_t l = 0; l < n_pt_; ++l)
            if (pt_seen[l])
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) t.push_back({pt_col_[l] + b, pt_col_[l] + a, hll[(size_t)l * 9 + 3 * a + b]});
    }
    {   // couplings, duplicated (camera, landmark) factors merged
        std::vector<double> w((size_t)dc_ * 3);
        for (int64_t k = 0; k < n_obs_;) {
            const int64_t i0 = order[k];
            const uint32_t c = cam_idx_h_[i0], l = pt_idx_h_[i0];
            std::fill(w.begin(), w.end(), 0.0);
            for (; k < n_obs_ && cam_idx_h_[order[k]] == c && pt_idx_h_[order[k]] == l; ++k) {
                const double* Jc = jc.data() + (size_t)2 * dc_ * order[k];
                const double* Jl = jl.data() + (size_t)6 * order[k];
                for (int a = 0; a < dc_; ++a)
                    for (int b = 0; b < 3; ++b) w[a * 3 + b] += Jc[a] * Jl[b] + Jc[dc_ + a] * Jl[3 + b];
            }
            for (int a = 0; a < dc_; ++a)
                for (int b = 0; b < 3; ++b) {
                    t.push_back({pt_col_[l] + b, cam_col(c, a), w[a * 3 + b]});
                    t.push_back({cam_col(c, a), pt_col_[l] + b, w[a * 3 + b]});
                }
        }
    }
    if ((int64_t)t.size() != nnz) return fail(kInvalidState, "Hessian export: entry count mismatch");
    std::sort(t.begin(), t.end(), [](const Trip& a, const Trip& b) { return a.col != b.col ? a.col < b.col : a.row < b.row; });
    std::fill(colptr, colptr + total + 1, 0);
    for (const Trip& e : t) colptr[e.col + 1]++;
    for (int64_t j = 0; j < total; ++j) colptr[j + 1] += colptr[j];
    for (int64_t k = 0; k < nnz; ++k) { rowidx[k] = t[k].row; values[k] = t[k].v; }
    return kOk;
}

// tests: the pair records of the default Schur form as they sit on the device (host-built and copied, or written by the device)
int Solver::get_pair_records(uint32_t* recs4_out, int64_t cap_slots) {
    if (!have_structure_) return fail(kInvalidState, "Block structure not built");
    if (!precs_ || cap_slots < n_pair_slots_) return fail(kInvalidInput, "pair records: none on this handle, or the buffer is too small");
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipMemcpy(recs4_out, precs_, (size_t)n_pair_slots_ * sizeof(PairRec), hipMemcpyDeviceToHost));
    return kOk;
}

// mask[l] = 1 for the landmarks this rank assembles and back-substitutes (the caller's landmark numbering)
int Solver::owned_landmarks(uint8_t* mask) const {
    for (int64_t l = 0; l < n_pt_; ++l) mask[l] = (lmap_[l] >= lm_lo_ && lmap_[l] < lm_hi_) ? 1 : 0;
    return kOk;
}

int Solver::get_landmark_blocks(double* hinv_out, double* gl_out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    if (hinv_out)   // the record's first six doubles: Hll^-1 as (00, 01, 02, 11, 12, 22) (ba_kernels.h); expanded below
        HIP_TRY(hipMemcpy2DAsync(hinv_out, 9 * sizeof(double), hinv_, kLmStride * sizeof(double), 6 * sizeof(double), (size_t)n_pt_,
                                 hipMemcpyDeviceToHost, stream_));
    if (gl_out) HIP_TRY(hipMemcpyAsync(gl_out, g_l_, 3 * n_pt_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    if (hinv_out)
        for (int64_t l = 0; l < n_pt_; ++l) {
            double* h = hinv_out + 9 * l;
            const double s[6] = {h[0], h[1], h[2], h[3], h[4], h[5]};
            h[0] = s[0]; h[1] = s[1]; h[2] = s[2]; h[3] = s[1]; h[4] = s[3]; h[5] = s[4]; h[6] = s[2]; h[7] = s[4]; h[8] = s[5];
        }
    if (tree_shard_) {  // back to the caller's landmark order
        std::vector<double> t;
        if (hinv_out) { t.assign(hinv_out, hinv_out + 9 * n_pt_); for (int64_t l = 0; l < n_pt_; ++l) memcpy(hinv_out + 9 * l, t.data() + 9 * (size_t)lmap_[l], 9 * sizeof(double)); }
        if (gl_out) { t.assign(gl_out, gl_out + 3 * n_pt_); for (int64_t l = 0; l < n_pt_; ++l) memcpy(gl_out + 3 * l, t.data() + 3 * (size_t)lmap_[l], 3 * sizeof(double)); }
    }
    return kOk;
}


}  // namespace apex
