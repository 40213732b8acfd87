// lm_loop.h -- the Levenberg-Marquardt loop shared by the device backends.
//
// optimize_with_mode (src/optimizer/levenberg_marquardt.rs:823-1031) drives any backend that can
// linearise + solve the damped system at the current point, report the step statistics, move to a
// trial point, evaluate its cost and then keep or undo the move.  The bundle-adjustment backend
// (solver.h) and the pose-graph backend (pg_solver.h) both implement LmBackend.
#pragma once
#include <stdint.h>

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
    int variant;                    // 0 Sparse (Cholesky), 1 Iterative (Jacobi-PCG on explicit S), 2 matrix-free PCG
    int use_jacobi_scaling;         // false (:352): s = 1/(1 + column norm) from iteration 0 (optimizer/mod.rs:749-763)
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

class LmBackend {
   public:
    virtual ~LmBackend() = default;
    virtual int cost(double* out) = 0;                                                // compute_cost at the current point
    virtual int solve_augmented(double lambda, int variant, double* step_out, double* grad_out) = 0;
    virtual int step_stats(double out3[3]) = 0;                                       // |g|, |step|, predicted reduction
    virtual int eval_step(double* trial_cost) = 0;                                    // x (+) step -> trial point, its cost
    virtual int commit_step() = 0;
    virtual int discard_step() = 0;                                                   // trial (+) (-step)
    virtual int parameter_norm(double* out) = 0;
    virtual int set_jacobi_scaling(bool on) = 0;  // on: column scaling from the Jacobian at the current point; off: none
    virtual const char* last_error() const = 0;
};

// Runs the loop; history rows as LmIterRecord.  Returns a Status (kOk unless a backend call failed hard).
int run_lm(LmBackend& b, LmConfig* cfg, LmResult* res, LmIterRecord* hist, int hist_cap);

}  // namespace apex
