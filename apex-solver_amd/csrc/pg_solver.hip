// pg_solver.hip -- see pg_solver.h
#include "pg_solver.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <numeric>

#include "pg_device.hpp"

namespace apex {

#define HIP_TRY(expr)                                      \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return check_hip(_e, #expr); \
    } while (0)

template <typename T>
static hipError_t dev_alloc(T** p, size_t n) {
    return hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
}
template <typename T>
static hipError_t upload(T** dptr, const std::vector<T>& hv) {
    if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
    hipError_t e = dev_alloc(dptr, hv.size());
    if (e != hipSuccess || hv.empty()) return e;
    return hipMemcpy(*dptr, hv.data(), hv.size() * sizeof(T), hipMemcpyHostToDevice);
}
static hipError_t alloc_zero(double** p, size_t n) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    hipError_t e = dev_alloc(p, n);
    if (e != hipSuccess) return e;
    return hipMemset(*p, 0, std::max<size_t>(n, 1) * sizeof(double));
}

PoseGraphSolver::PoseGraphSolver(int64_t n_v, int64_t n_e, int device) : n_v_(n_v), n_e_(n_e), device_(device) {}

PoseGraphSolver::~PoseGraphSolver() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (eager_host_) (void)hipHostFree(eager_host_);
    void* ptrs[] = {poses_[0], poses_[1], posep_[0], posep_[1], e_from_, e_to_, meas_, fix_, g_, rhs_, d_, work_, partial_, scal_, scale_, prior_v_, prior_data_, prior_res_};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (stream_) (void)hipStreamDestroy(stream_);
}

int PoseGraphSolver::check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return kOk;
    return fail(kDeviceError, std::string("HIP error in ") + what + ": " + hipGetErrorString(e));
}

PGView PoseGraphSolver::view(int which) const {
    PGView v;
    v.n_v = n_v_; v.n_e = n_e_;
    v.posep = posep_[which]; v.e_from = e_from_; v.e_to = e_to_; v.meas = meas_;
    v.huber_delta = huber_delta_;
    v.n_prior = n_prior_; v.prior_v = prior_v_; v.prior_data = prior_data_;
    return v;
}

// PriorFactor blocks (prior_factor.rs:96-108); replaces the set.  data7 in to_vector order [t, w, i, j, k].
int PoseGraphSolver::set_priors(int64_t n, const uint32_t* vertex, const double* data7, const double* huber_delta) {
    if (!have_structure_) return fail(kInvalidState, "Block structure not built. Call set_structure() first.");
    if (n < 0 || n > (1 << 24)) return fail(kInvalidInput, "prior count out of range");
    for (int64_t k = 0; k < n; ++k)
        if ((int64_t)vertex[k] >= n_v_) return fail(kInvalidInput, "prior on a vertex that does not exist");
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipStreamSynchronize(stream_));
    if (prior_v_) { (void)hipFree(prior_v_); prior_v_ = nullptr; }
    if (prior_data_) { (void)hipFree(prior_data_); prior_data_ = nullptr; }
    if (prior_res_) { (void)hipFree(prior_res_); prior_res_ = nullptr; }
    n_prior_ = (int)n;
    have_step_ = have_trial_ = false;
    if (n == 0) return kOk;
    std::vector<uint32_t> hv((size_t)n);
    std::vector<double> hd((size_t)n * kPoseStride, 0.0);
    for (int64_t k = 0; k < n; ++k) {
        hv[k] = (uint32_t)vmap_[vertex[k]];
        memcpy(hd.data() + (size_t)k * kPoseStride, data7 + 7 * k, 7 * sizeof(double));
        hd[(size_t)k * kPoseStride + 7] = huber_delta ? huber_delta[k] : -1.0;
    }
    HIP_TRY(hipMalloc(&prior_v_, hv.size() * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&prior_data_, hd.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(prior_v_, hv.data(), hv.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(prior_data_, hd.data(), hd.size() * sizeof(double), hipMemcpyHostToDevice));
    return kOk;
}

int PoseGraphSolver::get_prior_residual(double* r7_out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    if (n_prior_ == 0) return kOk;
    HIP_TRY(hipSetDevice(device_));
    if (!prior_res_) HIP_TRY(hipMalloc(&prior_res_, (size_t)n_prior_ * 7 * sizeof(double)));   // kept with the priors (set_priors frees it)
    launch_pg_prior_export(view(cur_), prior_res_, stream_);
    HIP_TRY(hipMemcpyAsync(r7_out, prior_res_, (size_t)n_prior_ * 7 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return kOk;
}

int PoseGraphSolver::set_structure(const uint32_t* e_from, const uint32_t* e_to, const double* meas7,
                                   const int64_t* pose_col, const uint8_t* fix6, double huber_delta) {
    if (n_v_ <= 0) return fail(kInvalidInput, "No pose variables found");
    if (n_e_ < 0 || n_e_ > 2000000000LL) return fail(kInvalidInput, "edge count out of range");
    for (int64_t e = 0; e < n_e_; ++e)
        if (e_from[e] >= (uint64_t)n_v_ || e_to[e] >= (uint64_t)n_v_)
            return fail(kInvalidInput, "edge " + std::to_string(e) + " references a missing variable");
    HIP_TRY(hipSetDevice(device_));
    if (!stream_) HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    huber_delta_ = huber_delta;
    pose_col_.assign(pose_col, pose_col + n_v_);
    n_ = 6 * n_v_;
    const int nt = (int)((n_ + kNB - 1) / kNB);
    n_pad_ = (int64_t)nt * kNB;

    // ---- internal vertex order: whole tiles of 24 consecutive vertices, permuted by a nested-dissection
    // ordering of the tile graph (see TilePlan::order) -----------------------------------------------
    std::vector<uint8_t> adjm((size_t)nt * nt, 0);
    for (int64_t e = 0; e < n_e_; ++e) {
        const int a = (int)(e_from[e] / kVertsPerTile), b = (int)(e_to[e] / kVertsPerTile);
        if (a != b) { adjm[(size_t)a * nt + b] = 1; adjm[(size_t)b * nt + a] = 1; }
    }
    const std::vector<int> tperm = TilePlan::order(nt, adjm, use_nd_, nd_leaf_);
    vmap_.resize(n_v_);
    for (int64_t v = 0; v < n_v_; ++v) vmap_[v] = (int)((int64_t)tperm[v / kVertsPerTile] * kVertsPerTile + v % kVertsPerTile);
    std::vector<uint8_t> present((size_t)nt * nt, 0);
    for (int I = 0; I < nt; ++I) present[(size_t)I * nt + I] = 1;
    std::vector<uint32_t> ef(n_e_), et(n_e_);
    for (int64_t e = 0; e < n_e_; ++e) {
        ef[e] = (uint32_t)vmap_[e_from[e]]; et[e] = (uint32_t)vmap_[e_to[e]];
        int a = (int)(ef[e] / kVertsPerTile), b = (int)(et[e] / kVertsPerTile);
        if (a < b) std::swap(a, b);
        present[(size_t)a * nt + b] = 1;
    }
    {
        const std::string err = tp_.build(nt, present, stream_);
        if (!err.empty()) return fail(kInvalidInput, "Hessian tiles: " + err);
    }
    // measurements are constants: normalise once (SE3::from_translation_quaternion, se3.rs:107-113)
    std::vector<double> mp((size_t)n_e_ * kPoseStride, 0.0);
    for (int64_t e = 0; e < n_e_; ++e) pose_normalise(meas7 + 7 * e, mp.data() + kPoseStride * e);
    std::vector<uint8_t> fx((size_t)6 * n_v_, 0);
    if (fix6)
        for (int64_t v = 0; v < n_v_; ++v) memcpy(fx.data() + 6 * (size_t)vmap_[v], fix6 + 6 * v, 6);
    HIP_TRY(upload(&e_from_, ef));
    HIP_TRY(upload(&e_to_, et));
    HIP_TRY(upload(&meas_, mp));
    HIP_TRY(upload(&fix_, fx));
    for (int w = 0; w < 2; ++w) {
        HIP_TRY(alloc_zero(&poses_[w], 7 * (size_t)n_v_));
        HIP_TRY(alloc_zero(&posep_[w], kPoseStride * (size_t)n_v_));
    }
    HIP_TRY(alloc_zero(&g_, n_pad_));
    HIP_TRY(alloc_zero(&rhs_, n_pad_));
    HIP_TRY(alloc_zero(&d_, n_pad_));
    HIP_TRY(alloc_zero(&work_, 6 * (size_t)n_pad_));
    HIP_TRY(alloc_zero(&partial_, 3 * (size_t)n_partial_));
    HIP_TRY(alloc_zero(&scal_, 16));
    HIP_TRY(hipDeviceSynchronize());
    have_structure_ = true;
    have_params_ = have_step_ = have_trial_ = false;
    cur_ = 0;
    return kOk;
}

int PoseGraphSolver::set_params(const double* poses7) {
    if (!have_structure_) return fail(kInvalidState, "Block structure not built. Call set_structure() first.");
    HIP_TRY(hipSetDevice(device_));
    std::vector<double> hp(7 * (size_t)n_v_);
    for (int64_t v = 0; v < n_v_; ++v) memcpy(hp.data() + 7 * (size_t)vmap_[v], poses7 + 7 * v, 7 * sizeof(double));
    HIP_TRY(hipMemcpyAsync(poses_[cur_], hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
    launch_pg_prepare(n_v_, poses_[cur_], posep_[cur_], stream_);
    HIP_TRY(hipStreamSynchronize(stream_));
    have_params_ = true; have_step_ = have_trial_ = false;
    return kOk;
}

int PoseGraphSolver::get_params(double* poses7) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    std::vector<double> hp(7 * (size_t)n_v_);
    HIP_TRY(hipMemcpyAsync(hp.data(), poses_[cur_], hp.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    for (int64_t v = 0; v < n_v_; ++v) memcpy(poses7 + 7 * v, hp.data() + 7 * (size_t)vmap_[v], 7 * sizeof(double));
    return kOk;
}

int PoseGraphSolver::cost_of(int which, double* out) {
    timer_.begin(kPgCost, stream_);
    launch_pg_cost(view(which), partial_, n_partial_, scal_, stream_);
    timer_.end(kPgCost, stream_);
    double ss = 0.0;
    HIP_TRY(hipMemcpyAsync(&ss, scal_, sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    const double nrm = sqrt(ss);  // compute_cost: 0.5 * norm_l2()^2 (optimizer/mod.rs:358-361)
    *out = 0.5 * nrm * nrm;
    return kOk;
}

int PoseGraphSolver::cost(double* out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    return cost_of(cur_, out);
}

// H + lambda I (tiles) and g = J^T r at the current parameters
int PoseGraphSolver::assemble(double lambda) {
    timer_.begin(kPgAssemble, stream_);
    HIP_TRY(tp_.zero_tiles());
    HIP_TRY(hipMemsetAsync(g_, 0, n_pad_ * sizeof(double), stream_));
    tp_.add_diag((int)n_, scaled_ ? 0.0 : lambda, 1.0);  // lambda on the real rows, identity on the padding rows
    launch_pg_edges(view(cur_), tp_.tilemap(), g_, stream_);
    launch_pg_priors(view(cur_), tp_.tilemap(), g_, stream_);
    if (scaled_) {  // Jacobi scaling: H := D H D, then the damping of the scaled system
        tp_.scale_sym(scale_);
        tp_.add_diag((int)n_, lambda, 1.0);
    }
    timer_.end(kPgAssemble, stream_);
    return kOk;
}

// SparseCholeskySolver::solve_augmented_equation (cholesky.rs:159-230): (J^T J + lambda I) dx = -J^T r
int PoseGraphSolver::solve_augmented(double lambda, int variant, double* step_out, double* grad_out) {
    if (!have_params_) return fail(kInvalidState, "Block structure not built or parameters not set");
    if (variant != 0) return fail(kInvalidInput, "the pose-graph backend has the sparse Cholesky solver only");
    HIP_TRY(hipSetDevice(device_));
    have_step_ = false;
    have_trial_ = false;   // (this solve's eager step evaluation overwrites the trial poses)
    ++step_serial_;
    last_lambda_ = lambda;
    int rc = assemble(lambda);
    if (rc != kOk) return rc;
    launch_pg_negate(n_pad_, g_, rhs_, stream_);
    if (scaled_) launch_vec_mul(n_pad_, rhs_, scale_, rhs_, stream_);  // -D g
    timer_.begin(kPgFactor, stream_);
    int failed = 0;
    // (one_wait_: the pivot flags and the dataflow launch's time-out word are read at the final wait below; the sweeps over a
    // failed factor are then void and the old path runs from the assembly on)
    bool speculative = one_wait_;
    HIP_TRY(tp_.factor(&failed, /*defer_flags=*/speculative));
    auto after_time_out = [&]() -> int {
        // the dataflow launch of the top groups timed out (the plan is back on the level launches): H is half updated
        ++n_factor_flow_timeouts_;
        int r = assemble(lambda);
        if (r != kOk) return r;
        launch_pg_negate(n_pad_, g_, rhs_, stream_);
        if (scaled_) launch_vec_mul(n_pad_, rhs_, scale_, rhs_, stream_);
        timer_.begin(kPgFactor, stream_);
        HIP_TRY(tp_.factor(&failed));
        timer_.end(kPgFactor, stream_);
        if (tp_.factor_flow_gave_up()) return fail(kDeviceError, "dataflow factorisation timed out twice");
        return kOk;
    };
    timer_.end(kPgFactor, stream_);
    if (!speculative) {
        if (tp_.factor_flow_gave_up()) { rc = after_time_out(); if (rc != kOk) return rc; }
        if (failed) return fail(kSingularMatrix, "Cholesky factorization failed (matrix may be singular)");
    }
    for (int attempt = 0;; ++attempt) {
        timer_.begin(kPgTriSolve, stream_);
        HIP_TRY(tp_.solve(rhs_, d_, work_));
        if (scaled_) launch_vec_mul(n_pad_, d_, scale_, d_, stream_);  // apply_inverse_scaling: step = D y
        timer_.end(kPgTriSolve, stream_);
        have_step_ = true;
        if (eager_eval_) {   // what the LM loop asks next rides on this solve's wait (step_stats, eval_step)
            if (!eager_host_) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&eager_host_), 8 * sizeof(double), hipHostMallocDefault));
            enqueue_step_stats();
            enqueue_trial_point(scal_ + 4);
            HIP_TRY(hipMemcpyAsync(eager_host_, scal_ + 1, 4 * sizeof(double), hipMemcpyDeviceToHost, stream_));
        }
        if (step_out || grad_out) {
            std::vector<double> h(n_);
            for (int pass = 0; pass < 2; ++pass) {
                double* out = pass == 0 ? step_out : grad_out;
                if (!out) continue;
                HIP_TRY(hipMemcpyAsync(h.data(), pass == 0 ? d_ : g_, n_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
                HIP_TRY(hipStreamSynchronize(stream_));
                if (scaled_)  // the caller's variables are the scaled ones: y = step / s, gradient = s g
                    for (int64_t i = 0; i < n_; ++i) h[i] = pass == 0 ? h[i] / scale_h_[i] : h[i] * scale_h_[i];
                for (int64_t v = 0; v < n_v_; ++v)
                    for (int a = 0; a < 6; ++a) out[pose_col_[v] + a] = h[6 * (size_t)vmap_[v] + a];
            }
        } else {
            HIP_TRY(hipStreamSynchronize(stream_));
        }
        if (speculative) {   // the flags the old path read behind the factorisation
            speculative = false;
            HIP_TRY(tp_.read_flags(&failed));
            if (tp_.factor_flow_gave_up()) {
                have_step_ = false;
                (void)tp_.sweep_timed_out();   // (clears the word a sweep over a broken factor may have raised)
                rc = after_time_out();
                if (rc != kOk) return rc;
                if (failed) return fail(kSingularMatrix, "Cholesky factorization failed (matrix may be singular)");
                attempt = -1;
                continue;   // the sweeps once more, over the good factor
            }
            if (failed) { have_step_ = false; (void)tp_.sweep_timed_out(); return fail(kSingularMatrix, "Cholesky factorization failed (matrix may be singular)"); }
        }
        if (!tp_.sweep_timed_out()) {
            if (eager_eval_) eager_serial_ = step_serial_;   // (the answers of THIS solve)
            return kOk;
        }
        // a dataflow sweep of this solve gave up (chol_kernels.hip, flow_wait): repeat it level by level (Solver::solve_augmented)
        have_step_ = false;
        if (attempt > 0 || !tp_.tri_flow()) return fail(kDeviceError, "triangular sweep timed out");
        tp_.enable_tri_flow(false);
    }
}

void PoseGraphSolver::enqueue_step_stats() {
    timer_.begin(kPgStats, stream_);
    launch_step_stats(n_, g_, d_, last_lambda_, scaled_ ? scale_ : nullptr, partial_, n_partial_, scal_ + 1, stream_);
    timer_.end(kPgStats, stream_);
}
void PoseGraphSolver::enqueue_trial_point(double* sumsq_out) {
    const int t = cur_ ^ 1;
    timer_.begin(kPgRetract, stream_);
    launch_pg_retract(n_v_, poses_[cur_], d_, 1.0, fix_, poses_[t], stream_);
    launch_pg_prepare(n_v_, poses_[t], posep_[t], stream_);
    timer_.end(kPgRetract, stream_);
    timer_.begin(kPgCost, stream_);
    launch_pg_cost(view(t), partial_, n_partial_, sumsq_out, stream_);
    timer_.end(kPgCost, stream_);
}

int PoseGraphSolver::step_stats(double out3[3]) {
    if (!have_step_) return fail(kInvalidState, "no step computed");
    if (eager_serial_ == step_serial_ && eager_host_) {   // read at the solve's wait
        out3[0] = sqrt(eager_host_[0]); out3[1] = sqrt(eager_host_[1]); out3[2] = 0.5 * eager_host_[2];
        return kOk;
    }
    HIP_TRY(hipSetDevice(device_));
    enqueue_step_stats();
    double h[3];
    HIP_TRY(hipMemcpyAsync(h, scal_ + 1, sizeof h, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    out3[0] = sqrt(h[0]);   // gradient.norm_l2()          (levenberg_marquardt.rs:746)
    out3[1] = sqrt(h[1]);   // step.norm_l2()              (:890)
    out3[2] = 0.5 * h[2];   // compute_predicted_reduction (:721-727)
    return kOk;
}

int PoseGraphSolver::eval_step(double* trial_cost) {
    if (!have_step_) return fail(kInvalidState, "no step computed");
    if (eager_serial_ == step_serial_ && eager_host_) {   // the trial point is in place, its cost was read at the solve's wait
        have_trial_ = true;
        const double nrm = sqrt(eager_host_[3]);
        *trial_cost = 0.5 * nrm * nrm;
        return kOk;
    }
    HIP_TRY(hipSetDevice(device_));
    const int t = cur_ ^ 1;
    timer_.begin(kPgRetract, stream_);
    launch_pg_retract(n_v_, poses_[cur_], d_, 1.0, fix_, poses_[t], stream_);
    launch_pg_prepare(n_v_, poses_[t], posep_[t], stream_);
    timer_.end(kPgRetract, stream_);
    have_trial_ = true;
    return cost_of(t, trial_cost);
}

int PoseGraphSolver::commit_step() {
    if (!have_trial_) return fail(kInvalidState, "no trial point");
    cur_ ^= 1;
    have_trial_ = false; have_step_ = false;
    return kOk;
}

// apply_negative_parameter_step (optimizer/mod.rs:343-356): inverse retraction of the trial point
int PoseGraphSolver::discard_step() {
    if (!have_trial_) return fail(kInvalidState, "no trial point");
    HIP_TRY(hipSetDevice(device_));
    const int t = cur_ ^ 1;
    timer_.begin(kPgRetract, stream_);
    launch_pg_retract(n_v_, poses_[t], d_, -1.0, fix_, poses_[cur_], stream_);
    launch_pg_prepare(n_v_, poses_[cur_], posep_[cur_], stream_);
    timer_.end(kPgRetract, stream_);
    HIP_TRY(hipStreamSynchronize(stream_));
    have_trial_ = false; have_step_ = false;
    return kOk;
}

int PoseGraphSolver::parameter_norm(double* out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    launch_sumsq(7 * n_v_, poses_[cur_], partial_, n_partial_, scal_ + 4, stream_);
    double h = 0.0;
    HIP_TRY(hipMemcpyAsync(&h, scal_ + 4, sizeof h, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    *out = sqrt(h);
    return kOk;
}

// ---- Jacobi column scaling (process_jacobian_generic, optimizer/mod.rs:749-763) -------------------
int PoseGraphSolver::ensure_scale_buffer() {
    if (scale_) return kOk;
    HIP_TRY(dev_alloc(&scale_, (size_t)n_pad_));
    std::vector<double> ones(n_pad_, 1.0);
    HIP_TRY(hipMemcpy(scale_, ones.data(), n_pad_ * sizeof(double), hipMemcpyHostToDevice));
    return kOk;
}

// compute_column_norms (linearizer/mod.rs:229-239): the squared column norms of the corrected Jacobian are the
// diagonal of J^T J, which the edge kernel already assembles.
int PoseGraphSolver::column_norms(double* norms_out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    const bool was = scaled_;
    scaled_ = false;
    int rc = assemble(0.0);
    scaled_ = was;
    if (rc != kOk) return rc;
    have_step_ = false;
    tp_.diag(work_);
    std::vector<double> h(n_);
    HIP_TRY(hipMemcpyAsync(h.data(), work_, n_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    for (int64_t v = 0; v < n_v_; ++v)
        for (int a = 0; a < 6; ++a) norms_out[pose_col_[v] + a] = sqrt(h[6 * (size_t)vmap_[v] + a]);
    return kOk;
}

int PoseGraphSolver::set_column_scaling(const double* scaling) {
    if (!have_structure_) return fail(kInvalidState, "Block structure not built");
    HIP_TRY(hipSetDevice(device_));
    have_step_ = false;
    if (!scaling) { scaled_ = false; return kOk; }
    int rc = ensure_scale_buffer();
    if (rc != kOk) return rc;
    scale_h_.assign(n_, 1.0);
    for (int64_t v = 0; v < n_v_; ++v)
        for (int a = 0; a < 6; ++a) scale_h_[6 * (size_t)vmap_[v] + a] = scaling[pose_col_[v] + a];
    for (double v : scale_h_) if (!(v > 0.0) || !std::isfinite(v)) return fail(kInvalidInput, "column scaling must be positive and finite");
    HIP_TRY(hipMemcpyAsync(scale_, scale_h_.data(), n_ * sizeof(double), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    scaled_ = true;
    return kOk;
}

int PoseGraphSolver::set_jacobi_scaling(bool on) {
    if (!on) { scaled_ = false; have_step_ = false; return kOk; }
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    int rc = ensure_scale_buffer();
    if (rc != kOk) return rc;
    scaled_ = false;
    rc = assemble(0.0);
    if (rc != kOk) return rc;
    tp_.diag(work_);
    launch_scaling_from_norms_sq(n_, work_, scale_, stream_);  // the padding keeps its 1
    scale_h_.resize(n_);
    HIP_TRY(hipMemcpyAsync(scale_h_.data(), scale_, n_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    scaled_ = true; have_step_ = false;
    return kOk;
}

int PoseGraphSolver::lm_optimize(LmConfig* cfg, LmResult* res, LmIterRecord* hist, int hist_cap) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    return run_lm(*this, cfg, res, hist, hist_cap);
}

// ---- parity / debug exports ------------------------------------------------------------------
int PoseGraphSolver::get_residual(double* r_out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    double* d = nullptr;
    HIP_TRY(dev_alloc(&d, 6 * (size_t)n_e_));
    launch_pg_export(view(cur_), d, nullptr, stream_);
    hipError_t e = hipMemcpyAsync(r_out, d, 6 * n_e_ * sizeof(double), hipMemcpyDeviceToHost, stream_);
    (void)hipStreamSynchronize(stream_);
    (void)hipFree(d);
    return check_hip(e, "get_residual");
}

int PoseGraphSolver::get_jacobian_blocks(double* j_out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    double* d = nullptr;
    HIP_TRY(dev_alloc(&d, 72 * (size_t)n_e_));
    launch_pg_export(view(cur_), nullptr, d, stream_);
    hipError_t e = hipMemcpyAsync(j_out, d, 72 * n_e_ * sizeof(double), hipMemcpyDeviceToHost, stream_);
    (void)hipStreamSynchronize(stream_);
    (void)hipFree(d);
    return check_hip(e, "get_jacobian_blocks");
}

int PoseGraphSolver::get_hessian(double lambda, double* H_out, double* g_out) {
    if (!have_params_) return fail(kInvalidState, "no parameters set");
    HIP_TRY(hipSetDevice(device_));
    int rc = assemble(lambda);
    if (rc != kOk) return rc;
    have_step_ = false;
    const size_t tile_elems = (size_t)kNB * kNB;
    std::vector<int64_t> col(n_, -1);
    for (int64_t v = 0; v < n_v_; ++v)
        for (int a = 0; a < 6; ++a) col[6 * (size_t)vmap_[v] + a] = pose_col_[v] + a;
    if (g_out) {
        std::vector<double> h(n_);
        HIP_TRY(hipMemcpyAsync(h.data(), g_, n_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        for (int64_t i = 0; i < n_; ++i) g_out[col[i]] = scaled_ ? h[i] * scale_h_[i] : h[i];
    }
    if (H_out) {
        memset(H_out, 0, (size_t)n_ * (size_t)n_ * sizeof(double));
        std::vector<double> t(tile_elems);
        const int nt = tp_.nt();
        for (int I = 0; I < nt; ++I)
            for (int J = 0; J <= I; ++J) {
                const int s = tp_.slot(I, J);
                if (s < 0 || s >= tp_.n_touched_slots()) continue;
                HIP_TRY(hipMemcpyAsync(t.data(), tp_.tiles() + (size_t)s * tile_elems, tile_elems * sizeof(double), hipMemcpyDeviceToHost, stream_));
                HIP_TRY(hipStreamSynchronize(stream_));
                for (int r = 0; r < kNB; ++r)
                    for (int c = 0; c < kNB; ++c) {
                        const int64_t gi = (int64_t)I * kNB + r, gj = (int64_t)J * kNB + c;
                        if (gi >= n_ || gj >= n_ || gj > gi) continue;
                        const double val = t[(size_t)r * kNB + c];
                        H_out[col[gi] * n_ + col[gj]] = val;
                        H_out[col[gj] * n_ + col[gi]] = val;
                    }
            }
    }
    return kOk;
}

}  // namespace apex
