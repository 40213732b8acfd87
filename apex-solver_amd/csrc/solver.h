// solver.h -- host side of the MI355X bundle-adjustment backend (one instance per optimize()).
//
// Mirrors the reference's SparseSchurComplementSolver + the LM loop that drives it
// (src/linalg/sparse/explicit_schur.rs:1038-1243, src/optimizer/levenberg_marquardt.rs:702-1031)
// with every per-observation / per-landmark / per-camera stage on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "ba_kernels.h"
#include "chol_kernels.h"
#include "lm_loop.h"
#include "schur_pairs.h"
#include "stage_timer.h"
#include "comm.h"
#include "tile_plan.h"


namespace apex {

enum Stage { kStAssembleCam = 0, kStAssembleLm, kStScatter, kStAllReduce, kStFactor, kStTriSolve, kStBackSub, kStStats,
             kStRetract, kStCost, kNumStages };

int mode_mask(int mode);  // 4 POSE + 2 LANDMARK + INTRINSIC of an APEXGPU_MODE_*
void shard_range(int64_t n_pt, const int64_t* ptr, int rank, int world, int64_t* lo, int64_t* hi);

class Solver : public LmBackend {
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
    int cost(double* out) override;                         // A16 on the current parameters
    int solve_augmented(double lambda, int variant, double* step_out, double* grad_out) override;
    int assemble_only(double lambda);
    int step_stats(double out3[3]) override;                // |g|, |step|, predicted reduction
    int eval_step(double* trial_cost) override;             // x (+) step into the trial set, A16 there
    int enqueue_step_stats();                               // (the kernels of the two calls above, without the read-back)
    int enqueue_trial_point(double* sumsq_out);
    int commit_step() override;
    int discard_step() override;                            // reference semantics: trial (+) (-step)
    int parameter_norm(double* out) override;
    int lm_optimize(LmConfig* cfg, LmResult* res, LmIterRecord* hist, int hist_cap);
    // Jacobi column scaling (optimizer/mod.rs:749-763; AssemblyBackend::compute_column_norms / apply_column_scaling,
    // linearizer/mod.rs:229-262).  With a scaling set, solve_augmented returns the SCALED step and gradient (what the
    // reference's solver returns for J diag(s)); the unscaled step stays on the device for eval_step.
    int column_norms(double* norms_out);                    // total_dof, global column order, at the current parameters
    int set_column_scaling(const double* scaling);          // total_dof, global column order; NULL: off
    int set_jacobi_scaling(bool on) override;               // on: s = 1 / (1 + norms) at the current parameters

    // the phases of the distributed Cholesky solve, for the single-process lockstep test (see solver.hip)
    int dist_phase(int phase, double lambda);
    struct DistBuf { double* ptr; size_t n; int root; };   // root >= 0: the sum is needed on that rank only; -1: everywhere
    void dist_buffers(int point, std::vector<DistBuf>* sums, int** max_flag);
    int export_step(double* step_out, double* grad_out);
    double dist_local_fraction() const { return tp_.local_work_fraction(); }
    int dist_top_columns() const { return tp_.n_top_columns(); }

    // parity / debug exports
    int get_residual(double* r_out);
    int get_jacobian_blocks(double* jc_out, double* jl_out);
    int get_schur(double* S_out, double* gred_out);  // reference camera-side order, dense
    int get_landmark_blocks(double* hinv_out, double* gl_out);
    // H = J^T J of the corrected Jacobian at the current parameters as a full symmetric CSC matrix in the global column
    // order (what SparseSchurComplementSolver::get_hessian caches, explicit_schur.rs:1146-1160, 1236-1238).  Two-call
    // pattern: with colptr == NULL only *nnz_out is set.
    int get_hessian_csc(int64_t* nnz_out, int64_t* colptr, int64_t* rowidx, double* values);
    int schur_matvec(double lambda, const double* x_in, double* y_explicit, double* y_implicit);
    int64_t tile_count() const { return tp_.n_slots(); }
    int n_tile_rows() const { return nt_; }
    double last_reg() const { return last_reg_; }
    int last_pcg_iters() const { return last_pcg_iters_; }
    int stage_times(double* ms, int64_t* launches);  // averaged HIP-event time per stage since reset
    void reset_stage_times();
    void enable_stage_timing(bool on) { timer_.enable(on); }
    void enable_stage_timing_only(uint32_t stage_mask) { timer_.enable_only(stage_mask); }
    void enable_graphs(bool on) { use_graphs_ = on; tp_.enable_graphs(on); }
    void enable_overlap(bool on) { tp_.enable_overlap(on); }
    void enable_tri_flow(bool on) { tp_.enable_tri_flow(on); }
    int sweep_timeouts() const { return tp_.sweep_timeouts(); }   // dataflow sweeps that gave up and were repeated level by level
    void debug_poison_next_solve(int which) { tp_.debug_poison_next_solve(which); }
    int debug_occupy_cus(int n_cus, int micros) { return check_hip(tp_.debug_occupy_cus(n_cus, micros), "debug_occupy_cus"); }
    void set_split_u1(int min_tasks) { tp_.set_split_u1(min_tasks); }
    void set_overlap_min(int n) { tp_.set_overlap_min(n); }
    void set_gate_min(int n) { tp_.set_gate_min(n); }
    // before set_structure: the handle will only run the matrix-free variant (2, IterativeSchurSolver).  S is never formed, so
    // neither is its tile structure beyond the diagonal blocks the Schur-Jacobi preconditioner needs, nor the pair list: the
    // set-up and the LM iteration no longer depend on the fill of S (a photo collection whose S is dense: tools/structure_sweep.py)
    void set_matrix_free_only(bool on) { matrix_free_only_opt_ = on; }
    // Automatic variant selection (round 5; the LM dispatch of levenberg_marquardt.rs:1039-1082 never fails on the fill of S,
    // so a drop-in backend may not either): when the tile plan of S is refused at set_structure -- its update list beyond
    // TilePlan's limit, or (single rank) its tiles beyond the free HBM -- the handle is built matrix-free only by itself and
    // variants 0 / 1 are answered by the matrix-free PCG (IterativeSchurSolver semantics, implicit_schur.rs:835-946): variant 0
    // at that solver's own defaults (500 iterations, 1e-9: implicit_schur.rs:94-95), variant 1 at the caller's cg parameters.
    // "auto_variant" 0 restores the refusal.  variant_used() / variant_reason() say what happened (apexgpu_variant_info).
    void set_eager_step_eval(bool on) { eager_eval_ = on; }
    void set_one_wait(bool on) { one_wait_ = on; }
    void set_device_pair_recs(bool on) { device_pair_recs_ = on; }   // before set_structure ("device_pair_list")
    int get_pair_records(uint32_t* recs4_out, int64_t cap_slots);     // tests: the pair records as they sit on the device
    void set_auto_variant(bool on) { auto_variant_ = on; }
    void set_variant_cost_permille(int pct) { variant_cost_permille_ = pct < 0 ? 0 : pct; }   // before set_structure (see build_plan)
    // [0] predicted ms per solve of the direct path (tile Cholesky + sweeps; 0: never evaluated -- "matrix_free_only"), [1] of the
    // matrix-free PCG at IterativeSchurSolver's cap, [2] what set_structure chose: 0 direct, 1 matrix-free by predicted cost,
    // 2 matrix-free because the plan was refused (size / memory), 3 matrix-free by the caller's option, [3] the cap behind [1]
    void variant_costs(double out[4]) const { out[0] = pred_direct_ms_; out[1] = pred_mf_ms_; out[2] = variant_choice_; out[3] = 500.0; }
    void set_max_tile_updates(int64_t n) { tp_.set_max_updates(n); }   // tests: force the refusal on a small problem
    int variant_used(int asked) const { return (auto_fallback_ && asked != 2) ? 2 : asked; }
    bool auto_fallback() const { return auto_fallback_; }
    const std::string& variant_reason() const { return fallback_reason_; }
    void set_two_side(int mode) { tp_.set_two_side(mode); }
    void set_factor_flow(int max_cols, int max_rows) { tp_.set_factor_flow(max_cols, max_rows); }
    int factor_flow_timeouts() const { return n_factor_flow_timeouts_; }
    void debug_poison_next_factor() { tp_.debug_poison_next_factor(); }
    void set_schur_form(int v) { rows_form_ = v == 4 ? 4 : 3; }   // 4 queued layout (default; nine-column cameras), 3 one running block per wave
    bool has_structure() const { return have_structure_; }
    void set_nd(bool on, int leaf) { use_nd_ = on; if (leaf > 0) nd_leaf_ = leaf; }
    void set_hubs_last(bool on) { hubs_last_ = on; }
    int n_hubs() const { return n_hubs_; }
    void set_dist_factor(bool on) { dist_factor_ = on; }   // before set_structure
    void set_tree_sharding(bool on) { tree_sharding_ = on; }  // before set_structure
    void set_dist_selftest(int world) { dist_selftest_ = world; }  // before set_structure; single rank only
    int owned_landmarks(uint8_t* mask) const;
    bool tree_sharded() const { return tree_shard_; }
    int n_levels() const { return tp_.n_levels(); }
    const TilePlan& plan() const { return tp_; }
    double schur_scatter_pairs() const { return (double)n_pairs_; }
    double pair_blocks() const { return (double)n_pair_blocks_; }
    double pair_slots() const { return (double)n_pair_slots_; }
    // the form that RUNS (4 = the queued layout: nine-column cameras)
    int schur_form() const { return (rows_form_ == 4 && !pair_queued_) ? 3 : rows_form_; }
    const double* setup_seconds() const { return setup_s_; }
    double touched_tiles() const { return (double)n_present_; }
    double local_obs() const { return (double)o_orig_h_.size(); }

    // multi-GPU (one process per GPU; landmarks sharded, S and g_red all-reduced over RCCL)
    int comm_init(int world, int rank, const void* unique_id128);          // RCCL over xGMI (production)
    int comm_init_shm(int world, int rank, const char* name);              // host shared memory (bring-up / tests, comm.h)
    int set_shard(int rank, int world);  // without RCCL: assemble only this rank's landmark range

    const char* last_error() const override { return err_.c_str(); }
    int dc() const { return dc_; }
    int64_t cam_dof_internal() const { return n_c_; }

   private:
    int fail(int code, const std::string& msg);
    int check_hip(hipError_t e, const char* what);
    BAView view(int which) const;
    TileMap tilemap() const;
    // for_factor: the result feeds tp_.factor() (a distributed plan then leaves the top tiles to the factorisation's
    // own exchange); otherwise S is complete on every rank (PCG, exports, the ladder's diagonal read)
    int assemble(double lambda, double diag_extra, bool for_factor = false);
    int assemble_local(double lambda, double diag_extra, bool for_factor = false);
    int assemble_finish();
    int assemble_implicit(double lambda);
    int implicit_pcg_solve(double lambda, int max_iter, double tol);
    int implicit_matvec(const double* x, double lam_local, double* y, bool reduce);
    int ensure_scale_buffers();
    int column_norms_sq_device();   // -> n2 in cam_scale_ / pt_scale_ (camera part all-reduced over the shards)
    int factor_and_solve(double lambda);
    int cholesky_attempt(int* failed_at);
    int cholesky_on_fresh_s(double lambda, double reg, int* failed_at);
    int n_factor_flow_timeouts_ = 0;
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
    int pad_rank_ = 0;          // rank that writes the identity on the padding rows of the last tile (assemble_local)
    std::string comm_err_;      // text of the last failed collective inside TilePlan's hooks
    int64_t lm_lo_ = 0, lm_hi_ = 0;  // landmark range owned by this rank
    std::unique_ptr<Communicator> comm_;
    int adopt_comm(std::unique_ptr<Communicator> c, const std::string& err);

    // host copies
    std::vector<int64_t> intr_col_, pose_col_, pt_col_;
    std::vector<int> o_orig_h_;
    std::thread free_thread_;   // unmaps the set-up's host lists off the caller's path

    // device
    hipStream_t stream_ = nullptr;
    // host -> device copies of caller (pageable) memory through two pinned chunks: the DMA of one overlaps the memcpy into the other
    int upload_staged(void* dst_dev, const void* src_host, size_t bytes);
    void* pin_[2] = {nullptr, nullptr};
    hipEvent_t pin_ev_[2] = {nullptr, nullptr};
    bool pin_busy_[2] = {false, false};   // pin_ev_[b] has been recorded behind a DMA out of pin_[b] and not waited for yet
    double *poses_[2] = {nullptr, nullptr}, *intr_[2] = {nullptr, nullptr}, *pts_[2] = {nullptr, nullptr};
    double* camp_[2] = {nullptr, nullptr};  // prepared cameras of the two parameter sets
    PairTask* ptasks_ = nullptr;     // rows_form_ 3: the sorted camera-pair list (schur_pairs.h)
    PairChunk* pchunks_ = nullptr;
    PairBlock* pblocks_ = nullptr;
    PairRec* precs_ = nullptr;
    PairQDesc* pqdesc_ = nullptr;    // rows_form_ 4 (and d_c = 9): the queued layout's descriptors, else null
    uint8_t *o_slot_ = nullptr, *wg_cam_n_ = nullptr;   // camera staging lists of the landmark-major kernels (BAView::o_slot)
    uint32_t* wg_cam_list_ = nullptr;
    bool matrix_free_only_opt_ = false;   // the caller's option ("matrix_free_only")
    bool matrix_free_only_ = false;       // the effective state of the structure that is built: the option, or the automatic selection
    bool one_wait_ = true;   // "one_wait": one host wait per Cholesky solve (solve_augmented); 0 = three, as in rounds 1-4
    // "eager_step_eval" (round 5): what the LM loop asks next of every solve -- the step statistics (apexgpu_step_stats) and the
    // trial point with its cost (apexgpu_eval_step) -- is enqueued behind the back-substitution and read at the solve's own wait:
    // the two calls then answer from the host, without a launch or a wait of their own (three device round trips per LM
    // iteration become one).  The answers are those of this very solve (step_serial_); single rank.
    bool device_gathers_ = true;   // "device_gathers": set-up, single rank: o_uv / co_uv / co_pt are permuted on the device (ba_structure.h)
    bool eager_eval_ = true;
    bool trial_pts_written_ = false;   // the back-substitution of this solve has written the trial points (enqueue_trial_point skips them)
    int64_t step_serial_ = 0, eager_serial_ = -1;
    double* eager_host_ = nullptr;               // pinned: [0..5] step statistics, [6] sum of squares at the trial point
    double* pcg_host_ = nullptr;                 // pinned: two slots of the matrix-free PCG's scalars (read one iteration behind)
    hipEvent_t pcg_ev_[2] = {nullptr, nullptr};
    bool device_pair_recs_ = true;   // queued layout: the pair records are written by the device (k_build_pair_recs_q), not built on the host and copied
    bool auto_variant_ = true, auto_fallback_ = false;   // see set_auto_variant
    int variant_cost_permille_ = 1000, variant_choice_ = 0;
    double pred_direct_ms_ = 0.0, pred_mf_ms_ = 0.0;
    std::string fallback_reason_;
    bool orec_fresh_ = false;   // orec_ holds the records of the current parameters' last linearisation
    const double* backsub_records() const { return orec_fresh_ ? orec_ : nullptr; }   // the projection records of THIS linearisation
    double* orec_ = nullptr;   // [local observations][4] projection records written by k_landmark_reduce (pair kernel, record form)
    int n_ptasks_ = 0;
    int64_t n_pair_blocks_ = 0, n_pair_slots_ = 0;
    bool pair_queued_ = false;       // what build_pair_lists arrived at (PairLists::queued)
    int rows_form_ = 4;              // 4 (default): the sorted pair list in the QUEUED layout (every lane group owns a block, no fold:
                                     // schur_pairs.h); 3: the same list reduced over the lanes of a wave (k_schur_pairs_r: every block
                                     // S(ci, cj) stored once by one wave, no atomics, no LDS accumulators: what six-column cameras
                                     // run).  Select before set_structure.  (Rounds 1-3 also carried a global-atomics form, 135 ms,
                                     // and two LDS row forms, 9.7 / 6.0 ms: deleted in rounds 4 and 6.)
    uint32_t *o_cam_ = nullptr, *o_pt_ = nullptr, *co_pt_ = nullptr;
    double2* co_uv_ = nullptr;
    int* co_rank_ = nullptr;
    double2* o_uv_ = nullptr;
    int *o_orig_ = nullptr, *pt_ptr_ = nullptr, *cam_ptr_ = nullptr, *cam_obs_ = nullptr;
    uint8_t *fix_pose_ = nullptr, *fix_intr_ = nullptr, *fix_pt_ = nullptr;
    TilePlan tp_;  // tiles of S, their factorisation and solves
    double *g_c_ = nullptr, *g_red_ = nullptr, *dcam_ = nullptr, *hinv_ = nullptr, *g_l_ = nullptr, *dl_ = nullptr;
    double *partial_ = nullptr, *scal_ = nullptr;  // reduction scratch, scalar outputs
    int* flags_ = nullptr;                          // [0] landmark inversion error
    std::vector<int> cmap_, cinv_;   // external camera -> internal camera and back
    bool use_nd_ = true;
    bool hubs_last_ = true;     // order cameras covisible with > max(16, 10 sqrt(n_cam)) others last (ba_structure.h)
    int n_hubs_ = 0, n_border_tiles_ = 1;
    int nd_leaf_ = 16;
    bool dist_factor_ = true;   // world > 1: factorise the elimination tree's subtrees on their owner ranks (tile_plan.h)
    bool tree_sharding_ = true; // ... and give every landmark to the rank whose columns it touches (set_structure)
    bool tree_shard_ = false;   // what set_structure arrived at
    int dist_selftest_ = 0;
    std::vector<int> lmap_;     // external landmark -> internal landmark (identity unless tree sharded)
    uint8_t* lam_mask_ = nullptr;  // tree sharding: cameras whose diagonal block gets lambda on this rank
    double* pcg_buf_ = nullptr;                    // 7 vectors of n_c_pad
    double *lmu_ = nullptr, *sd_ = nullptr, *minv_ = nullptr;  // matrix-free variant: {pt, u_l} records, diag blocks of S, their inverses
    double *cam_scale_ = nullptr, *pt_scale_ = nullptr;   // Jacobi scaling, internal order ([n_c_pad] with 1 on the padding, [3 n_pt])
    std::vector<double> cam_scale_h_, pt_scale_h_;
    bool scaled_ = false;
    int n_partial_ = 1024;

    bool use_graphs_ = true;
    double setup_s_[6] = {0, 0, 0, 0, 0, 0};  // set_structure by phase: order + tile structure, landmark / camera lists, tile plan, Schur lists, uploads, total

    StageTimer<kNumStages> timer_;

    std::string err_;
};

}  // namespace apex
