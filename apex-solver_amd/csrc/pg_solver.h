// pg_solver.h -- host side of the MI355X SE3 pose-graph backend (BASELINE.json configs[1]).
//
// Mirrors SparseCholeskySolver (src/linalg/sparse/cholesky.rs:159-230) driven by the LM loop
// (src/optimizer/levenberg_marquardt.rs:823-1031) on a problem of BetweenFactor<SE3> blocks
// (src/factors/between_factor.rs:268-322) as bin/pose_graph_g2o.rs:748-830 builds it:
// H = J^T J is assembled block-sparse (6x6 blocks) straight from the edges into 144x144 tiles,
// H + lambda I is factorised by the level-scheduled tile Cholesky of TilePlan -- no Schur complement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "lm_loop.h"
#include "pg_kernels.h"
#include "stage_timer.h"
#include "tile_plan.h"

namespace apex {

enum PgStage { kPgAssemble = 0, kPgFactor, kPgTriSolve, kPgStats, kPgRetract, kPgCost, kPgNumStages };

class PoseGraphSolver : public LmBackend {
   public:
    PoseGraphSolver(int64_t n_v, int64_t n_e, int device);
    ~PoseGraphSolver() override;

    int set_structure(const uint32_t* e_from, const uint32_t* e_to, const double* meas7, const int64_t* pose_col,
                      const uint8_t* fix6, double huber_delta);
    int set_params(const double* poses7);
    int set_priors(int64_t n, const uint32_t* vertex, const double* data7, const double* huber_delta);   // PriorFactor blocks
    int get_prior_residual(double* r7_out);
    int get_params(double* poses7);

    int cost(double* out) override;
    int solve_augmented(double lambda, int variant, double* step_out, double* grad_out) override;
    int step_stats(double out3[3]) override;
    int eval_step(double* trial_cost) override;
    void enqueue_step_stats();                       // (the kernels of the two calls above, without the read-back)
    void enqueue_trial_point(double* sumsq_out);
    int commit_step() override;
    int discard_step() override;
    int parameter_norm(double* out) override;
    int lm_optimize(LmConfig* cfg, LmResult* res, LmIterRecord* hist, int hist_cap);
    // Jacobi column scaling (optimizer/mod.rs:749-763), same contract as Solver's
    int column_norms(double* norms_out);
    int set_column_scaling(const double* scaling);
    int set_jacobi_scaling(bool on) override;

    // parity / debug exports (caller's edge and column order)
    int get_residual(double* r_out);
    int get_jacobian_blocks(double* j_out);
    int get_hessian(double lambda, double* H_out, double* g_out);  // dense J^T J + lambda I, J^T r

    void enable_graphs(bool on) { tp_.enable_graphs(on); }
    void enable_overlap(bool on) { tp_.enable_overlap(on); }
    void enable_tri_flow(bool on) { tp_.enable_tri_flow(on); }
    int sweep_timeouts() const { return tp_.sweep_timeouts(); }
    void debug_poison_next_solve(int which) { tp_.debug_poison_next_solve(which); }
    void set_split_u1(int min_tasks) { tp_.set_split_u1(min_tasks); }
    void set_one_wait(bool on) { one_wait_ = on; }
    void set_eager_step_eval(bool on) { eager_eval_ = on; }
    void set_overlap_min(int n) { tp_.set_overlap_min(n); }
    void set_gate_min(int n) { tp_.set_gate_min(n); }
    void set_two_side(int mode) { tp_.set_two_side(mode); }
    void set_factor_flow(int max_cols, int max_rows) { tp_.set_factor_flow(max_cols, max_rows); }
    int factor_flow_timeouts() const { return n_factor_flow_timeouts_; }
    void debug_poison_next_factor() { tp_.debug_poison_next_factor(); }
    void set_nd(bool on, int leaf) { use_nd_ = on; if (leaf > 0) nd_leaf_ = leaf; }
    void enable_stage_timing(bool on) { timer_.enable(on); }
    void enable_stage_timing_only(uint32_t stage_mask) { timer_.enable_only(stage_mask); }
    void reset_stage_times() { timer_.reset(); }
    int stage_times(double* ms, int64_t* n) { return timer_.times(ms, n); }
    int64_t n_vertices() const { return n_v_; }
    int n_tile_rows() const { return tp_.nt(); }
    int64_t tile_count() const { return tp_.n_slots(); }
    int64_t touched_tiles() const { return tp_.n_touched_slots(); }
    int n_levels() const { return tp_.n_levels(); }
    const TilePlan& plan() const { return tp_; }
    const char* last_error() const override { return err_.c_str(); }

   private:
    int fail(int code, const std::string& msg) { err_ = msg; return code; }
    int check_hip(hipError_t e, const char* what);
    PGView view(int which) const;
    int assemble(double lambda);
    int ensure_scale_buffer();
    int cost_of(int which, double* out);

    int64_t n_v_, n_e_;
    int device_;
    int64_t n_ = 0, n_pad_ = 0;
    double huber_delta_ = 0.0;
    bool have_structure_ = false, have_params_ = false, have_step_ = false, have_trial_ = false;
    int cur_ = 0;
    double last_lambda_ = 0.0;
    bool use_nd_ = true;
    int nd_leaf_ = 2;
    std::vector<int64_t> pose_col_;
    std::vector<int> vmap_;  // caller's vertex -> internal vertex
    hipStream_t stream_ = nullptr;
    TilePlan tp_;
    double *poses_[2] = {nullptr, nullptr}, *posep_[2] = {nullptr, nullptr};
    uint32_t *e_from_ = nullptr, *e_to_ = nullptr;
    int n_prior_ = 0;
    int n_factor_flow_timeouts_ = 0;
    // one device round trip per LM iteration (round 5, as in Solver): the pivot flags are read at the solve's final wait, and the
    // step statistics and the trial cost the LM loop asks next ride on that wait too ("one_wait", "eager_step_eval")
    bool one_wait_ = true, eager_eval_ = true;
    int64_t step_serial_ = 0, eager_serial_ = -1;
    double* eager_host_ = nullptr;   // pinned: [0..2] step statistics, [3] sum of squares at the trial point
    uint32_t* prior_v_ = nullptr;
    double* prior_data_ = nullptr;
    double* prior_res_ = nullptr;   // staging of get_prior_residual
    double* meas_ = nullptr;
    uint8_t* fix_ = nullptr;
    double *g_ = nullptr, *rhs_ = nullptr, *d_ = nullptr, *work_ = nullptr, *partial_ = nullptr, *scal_ = nullptr;
    double* scale_ = nullptr;        // Jacobi scaling, internal order, [n_pad] with 1 on the padding
    std::vector<double> scale_h_;
    bool scaled_ = false;
    int n_partial_ = 256;
    StageTimer<kPgNumStages> timer_;
    std::string err_;
};

}  // namespace apex
