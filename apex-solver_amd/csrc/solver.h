// solver.h -- host side of the MI355X bundle-adjustment backend (one instance per optimize()).
//
// Mirrors the reference's SparseSchurComplementSolver + the LM loop that drives it
// (src/linalg/sparse/explicit_schur.rs:1038-1243, src/optimizer/levenberg_marquardt.rs:702-1031)
// with every per-observation / per-landmark / per-camera stage on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

#include "ba_kernels.h"
#include "chol_kernels.h"
#include "tile_plan.h"

struct ncclComm;  // RCCL communicator (optional)

namespace apex {

// status codes: 0 ok; negative values mirror LinAlgError (src/linalg/mod.rs:76-101)
enum Status : int {
    kOk = 0,
    kFactorizationFailed = -1,
    kSingularMatrix = -2,
    kSparseMatrixCreation = -3,
    kMatrixConversion = -4,
    kInvalidInput = -5,
    kInvalidState = -6,
    kDeviceError = -10,
};

// OptimizationStatus discriminants (src/optimizer/mod.rs:189-216) + one for a failed linear solve
enum LmStatus : int {
    kConverged = 0, kMaxIterationsReached = 1, kCostToleranceReached = 2, kParameterToleranceReached = 3,
    kGradientToleranceReached = 4, kNumericalFailure = 5, kTimeout = 7, kTrustRegionRadiusTooSmall = 8,
    kMinCostThresholdReached = 9, kInvalidNumericalValues = 11, kLinearSolveFailed = 100,
};

struct LmConfig {               // LevenbergMarquardtConfig (levenberg_marquardt.rs:213-358)
    int max_iterations;         // 50 (20 in for_bundle_adjustment)
    double cost_tolerance;      // 1e-6
    double parameter_tolerance; // 1e-8
    double gradient_tolerance;  // 1e-10
    double damping;             // 1e-3
    double damping_min;         // 1e-12
    double damping_max;         // 1e12
    double damping_nu;          // 2.0
    double trust_region_radius;     // 1e4
    double min_trust_region_radius; // 1e-32
    double min_cost_threshold;      // < 0: None
    double timeout_s;               // <= 0: None
    int variant;                    // 0 Sparse (Cholesky), 1 Iterative (Jacobi-PCG on explicit S)
};

struct LmIterRecord {  // one row of the per-iteration history
    double cost, damping, rho, accepted, gradient_norm, step_norm, predicted_reduction, trial_cost;
};

struct LmResult {
    int status;
    int iterations;
    double initial_cost, final_cost;
    double final_gradient_norm, final_step_norm;
    double elapsed_s;
    int cost_evaluations, jacobian_evaluations, successful_steps, unsuccessful_steps;
};

enum Stage { kStAssembleCam = 0, kStAssembleLm, kStScatter, kStAllReduce, kStFactor, kStTriSolve, kStBackSub, kStStats,
             kStRetract, kStCost, kNumStages };

void shard_range(int64_t n_pt, const int64_t* ptr, int rank, int world, int64_t* lo, int64_t* hi);

class Solver {
   public:
    Solver(int64_t n_cam, int64_t n_pt, int64_t n_obs, int mode, int device);
    ~Solver();

    int set_structure(const uint32_t* cam_idx, const uint32_t* pt_idx, const double* obs_uv, const int64_t* intr_col,
                      const int64_t* pose_col, const int64_t* pt_col, const uint8_t* fix_pose, const uint8_t* fix_intr,
                      const uint8_t* fix_pt, double huber_delta);
    int set_params(const double* poses, const double* intr, const double* points);
    int get_params(double* poses, double* intr, double* points);
    void set_cg_params(int max_iter, double tol) { cg_max_iter_ = max_iter; cg_tol_ = tol; }

    // hot path
    int cost(double* out);                                  // A16 on the current parameters
    int solve_augmented(double lambda, int variant, double* step_out, double* grad_out);
    int assemble_only(double lambda);
    int step_stats(double out3[3]);                         // |g|, |step|, predicted reduction
    int eval_step(double sign_unused, double* trial_cost);  // x (+) step into the trial set, A16 there
    int commit_step();
    int discard_step();                                     // reference semantics: trial (+) (-step)
    int parameter_norm(double* out);
    int lm_optimize(LmConfig* cfg, LmResult* res, LmIterRecord* hist, int hist_cap);

    // parity / debug exports
    int get_residual(double* r_out);
    int get_jacobian_blocks(double* jc_out, double* jl_out);
    int get_schur(double* S_out, double* gred_out);  // reference camera-side order, dense
    int get_landmark_blocks(double* hinv_out, double* gl_out);
    int get_step_internal(double* dc_out, double* dl_out);
    int64_t tile_count() const { return tp_.n_slots(); }
    int n_tile_rows() const { return nt_; }
    double last_reg() const { return last_reg_; }
    int last_pcg_iters() const { return last_pcg_iters_; }
    int stage_times(double* ms, int64_t* launches);  // averaged HIP-event time per stage since reset
    void reset_stage_times();
    void enable_stage_timing(bool on) { timing_ = on; }
    void enable_graphs(bool on) { use_graphs_ = on; tp_.enable_graphs(on); }
    void use_row_schur(bool on) { use_rows_ = on; }
    void set_rows_debug(int v) { rows_dbg_ = v; }
    void set_nd(bool on, int leaf) { use_nd_ = on; if (leaf > 0) nd_leaf_ = leaf; }
    int n_levels() const { return tp_.n_levels(); }
    double schur_scatter_pairs() const { return (double)n_pairs_; }
    double touched_tiles() const { return (double)n_present_; }
    double local_obs() const { return (double)o_orig_h_.size(); }

    // multi-GPU (one process per GPU; landmarks sharded, S and g_red all-reduced over RCCL)
    int comm_init(int world, int rank, const void* unique_id128);
    int set_shard(int rank, int world);  // without RCCL: assemble only this rank's landmark range

    const char* last_error() const { return err_.c_str(); }
    int dc() const { return dc_; }
    int64_t cam_dof_internal() const { return n_c_; }

   private:
    int fail(int code, const std::string& msg);
    int check_hip(hipError_t e, const char* what);
    BAView view(int which) const;
    TileMap tilemap() const;
    int assemble(double lambda, double diag_extra);
    int factor_and_solve(double lambda);
    int cholesky_attempt(int* failed_at);
    int tri_solve();
    int pcg_solve();
    int cost_of(int which, double* out);
    void stage_begin(int st);
    void stage_end(int st);

    // sizes
    int64_t n_cam_, n_pt_, n_obs_;
    int mode_, dc_, device_;
    int64_t n_c_ = 0, n_c_pad_ = 0;
    int nt_ = 0;
    double huber_delta_ = 1.0;
    bool have_structure_ = false, have_params_ = false, have_step_ = false, have_trial_ = false;
    int cur_ = 0;  // index of the current parameter set (0/1); the other one is the trial set
    double last_lambda_ = 0.0, last_reg_ = 0.0;
    int last_pcg_iters_ = 0;
    int cg_max_iter_ = 200;   // SparseSchurComplementSolver::new (explicit_schur.rs:211-212)
    double cg_tol_ = 1e-6;
    int64_t n_pairs_ = 0, n_present_ = 0;
    // shard
    int rank_ = 0, world_ = 1;
    int64_t lm_lo_ = 0, lm_hi_ = 0;  // landmark range owned by this rank
    ncclComm* comm_ = nullptr;

    // host copies
    std::vector<int64_t> intr_col_, pose_col_, pt_col_;
    std::vector<int> o_orig_h_;

    // device
    hipStream_t stream_ = nullptr;
    double *poses_[2] = {nullptr, nullptr}, *intr_[2] = {nullptr, nullptr}, *pts_[2] = {nullptr, nullptr};
    double* camp_[2] = {nullptr, nullptr};  // prepared cameras of the two parameter sets
    RowTask* rtasks_ = nullptr;
    RowBatch* rbatches_ = nullptr;
    uint16_t* cam_obs_off_ = nullptr;
    int* nbr_ = nullptr;
    int n_rtasks_ = 0;
    int rows_dbg_ = 0;      // timing-only ablation switches of k_schur_rows (results are wrong when != 0)
    bool use_rows_ = true;  // Schur reduction: LDS row form (default) or the global-atomics form
    uint32_t *o_cam_ = nullptr, *o_pt_ = nullptr;
    double2* o_uv_ = nullptr;
    int *o_orig_ = nullptr, *pt_ptr_ = nullptr, *cam_ptr_ = nullptr, *cam_obs_ = nullptr;
    uint8_t *fix_pose_ = nullptr, *fix_intr_ = nullptr, *fix_pt_ = nullptr;
    TilePlan tp_;  // tiles of S, their factorisation and solves
    double *g_c_ = nullptr, *g_red_ = nullptr, *dcam_ = nullptr, *hinv_ = nullptr, *g_l_ = nullptr, *dl_ = nullptr;
    double *partial_ = nullptr, *scal_ = nullptr;  // reduction scratch, scalar outputs
    int* flags_ = nullptr;                          // [0] landmark inversion error
    ScatterTask* tasks_ = nullptr;
    int n_tasks_ = 0;
    std::vector<int> cmap_, cinv_;   // external camera -> internal camera and back
    bool use_nd_ = true;
    int nd_leaf_ = 16;
    double* pcg_buf_ = nullptr;                    // 7 vectors of n_c_pad
    int n_partial_ = 1024;

    bool use_graphs_ = true;

    // timing
    bool timing_ = false;
    std::vector<hipEvent_t> ev_pool_;
    std::pair<hipEvent_t, hipEvent_t> ev_open_[kNumStages] = {};
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> ev_pending_;
    void resolve_stage_events();
    double stage_ms_[kNumStages] = {0};
    int64_t stage_n_[kNumStages] = {0};

    std::string err_;
};

}  // namespace apex
