/*
 * apexgpu.h -- C ABI of the MI355X-native bundle-adjustment backend for apex-solver.
 *
 * This is the drop-in boundary: what a thin Rust `extern "C"` shim that
 * implements apex-solver's `LinearSolver<M>` / `StructureAware` / `AssemblyBackend` traits binds to
 * (the shim itself is in INTEGRATION.md), and what the bundled C++ Levenberg-Marquardt twin and the
 * Python host layer call.  Plain pointers and sizes only; all host buffers are caller-owned and are
 * copied in/out during the call; one handle per `optimize()`; a handle is not re-entrant; no
 * callbacks into the caller.
 *
 * Every function returns an int status: 0 = ok, negative = error class mirroring
 * `LinAlgError` (src/linalg/mod.rs:76-101); the message is available from apexgpu_last_error().
 *
 * Conventions (the reference's): pose = [tx,ty,tz,qw,qx,qy,qz] (SE3 as
 * VariableEnum::to_vector stores it, src/core/problem.rs:161-173), intrinsics = [f,k1,k2]
 * (bal_pinhole.rs:160-181), point = [x,y,z]; tangent step of a pose = [rho(3); theta(3)]
 * applied as the right-plus retraction T o Exp(delta) (apex-manifolds/src/lib.rs:269-283);
 * vectors named *_out of length total_dof are in the reference's GLOBAL column order, i.e. the
 * lexicographic order of the variable names (src/optimizer/mod.rs:530-536) that the caller passes
 * as column offsets to apexgpu_set_structure.
 */
#ifndef APEXGPU_H
#define APEXGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct apexgpu_solver apexgpu_solver;

/* status codes = LinAlgError variants (src/linalg/mod.rs:76-101) */
#define APEXGPU_OK 0
#define APEXGPU_ERR_FACTORIZATION_FAILED (-1)
#define APEXGPU_ERR_SINGULAR_MATRIX (-2)
#define APEXGPU_ERR_SPARSE_MATRIX_CREATION (-3)
#define APEXGPU_ERR_MATRIX_CONVERSION (-4)
#define APEXGPU_ERR_INVALID_INPUT (-5)
#define APEXGPU_ERR_INVALID_STATE (-6)
#define APEXGPU_ERR_DEVICE (-10)

/* OptimizationType of the projection factors (src/factors/mod.rs:82-101,
 * bin/bundle_adjustment.rs:268-284): which variables each factor is keyed on. */
#define APEXGPU_MODE_BUNDLE_ADJUSTMENT 0 /* keys [pose, pt]        -> 6 camera DOF per block */
#define APEXGPU_MODE_SELF_CALIBRATION 1  /* keys [pose, pt, intr]  -> 9 camera DOF per block */
/* The other OptimizeParams<POSE, LANDMARK, INTRINSIC> configurations (src/factors/mod.rs:88-101): a factor keyed on the
 * optimised blocks only, the rest constants of the factor (with_fixed_pose / with_fixed_landmarks / its camera model).
 * The variable set and the global column order are unchanged (every pose_*, intr_*, pt_* variable exists, as in the
 * reference's bin, bundle_adjustment.rs:232-257); the blocks that are not optimised have no Jacobian columns, so the
 * damped system gives them a zero step.  Same kernels with the Jacobian's column groups masked. */
#define APEXGPU_MODE_ONLY_POSE 2                /* <true,  false, false> */
#define APEXGPU_MODE_ONLY_LANDMARKS 3           /* <false, true,  false> */
#define APEXGPU_MODE_ONLY_INTRINSICS 4          /* <false, false, true > */
#define APEXGPU_MODE_POSE_AND_INTRINSICS 5      /* <true,  false, true > */
#define APEXGPU_MODE_LANDMARKS_AND_INTRINSICS 6 /* <false, true,  true > */

/* SchurVariant (src/linalg/sparse/explicit_schur.rs:58-65) */
#define APEXGPU_VARIANT_SPARSE 0    /* explicit S + Cholesky (solve_with_cholesky, :539-634) */
#define APEXGPU_VARIANT_ITERATIVE 1 /* explicit S + Jacobi-PCG (solve_with_pcg, :639-756)    */
/* IterativeSchurSolver (src/linalg/sparse/implicit_schur.rs:163-251, 456-679, 835-946): S is never formed; PCG on the
 * matrix-free operator S x = H_cc x - H_cp (H_pp^-1 (H_cp^T x)) with the Schur-Jacobi preconditioner (inverse diagonal
 * blocks of S per pose / intrinsics variable); tolerance tol*max(|b|,1) with apexgpu_set_cg_params (reference default of
 * IterativeSchurSolver::new: 500 iterations, 1e-9).  grad_out stays +J^T r as for the other variants (the reference's
 * IterativeSchurSolver::get_gradient returns -J^T r, :1053-1057, which its LM loop never consumes). */
#define APEXGPU_VARIANT_IMPLICIT 2

/* ---- lifetime ------------------------------------------------------------------------------
 * Replaces SparseSchurComplementSolver::new() (explicit_schur.rs:205-217) + the per-optimize
 * state of optimize_with_mode (levenberg_marquardt.rs:823-848). */
int apexgpu_create(int64_t n_cam, int64_t n_pt, int64_t n_obs, int mode, int device, apexgpu_solver** out);
void apexgpu_destroy(apexgpu_solver* h);
const char* apexgpu_last_error(const apexgpu_solver* h);
const char* apexgpu_version(void);
/* The structure set-up builds its lists (3.7 GB on BAL final-13682) in host blocks that are cached WHILE A HANDLE IS ALIVE and
 * serve the next apexgpu_set_structure: handing them back to the system costs 0.2 s of page zapping that stalls the caller's
 * next GPU calls even from a background thread (tools/setup_probe.py).  The cache holds the blocks of the last set-up only (a
 * smaller structure after a larger one releases the surplus), at most 8 GB, and the apexgpu_destroy of the last live handle
 * returns everything to the system.  apexgpu_trim_host_cache does that at once (*released_bytes, may be NULL),
 * apexgpu_host_cache_bytes reports what is held, the environment variable APEX_HOST_CACHE=0 switches the cache off (blocks are
 * then freed where they are released).  No counterpart in the reference: its symbolic structures live and die with the solver
 * (src/linearizer/cpu/sparse.rs:54-105). */
int apexgpu_trim_host_cache(int64_t* released_bytes);
int64_t apexgpu_host_cache_bytes(void);

/* ---- structure -------------------------------------------------------------------------------
 * Replaces StructureAware::initialize_structure (src/linalg/mod.rs:116-123; explicit_schur.rs:
 * 1038-1062, 244-323), build_symbolic_structure (src/linearizer/cpu/sparse.rs:54-105) and the
 * problem description the reference keeps in Problem's residual blocks
 * (bin/bundle_adjustment.rs:391-441): one BAL projection factor per observation.
 *   cam_idx/pt_idx[n_obs]   variable indices of each factor, in residual-block insertion order
 *   obs_uv[2*n_obs]         observed pixel (u,v) per factor
 *   intr_col/pose_col[n_cam], pt_col[n_pt]  first global column of intr_i / pose_i / pt_j
 *   fix_pose[6*n_cam], fix_intr[3*n_cam], fix_pt[3*n_pt]  per-DOF fixed masks (may be NULL);
 *                           fixed DOF are zeroed in the step at apply time only
 *                           (src/core/problem.rs:185-197), they stay in the linear system
 *   huber_delta             HuberLoss scale on every factor (<= 0: no loss function)          */
int apexgpu_set_structure(apexgpu_solver* h, const uint32_t* cam_idx, const uint32_t* pt_idx, const double* obs_uv,
                          const int64_t* intr_col, const int64_t* pose_col, const int64_t* pt_col,
                          const uint8_t* fix_pose, const uint8_t* fix_intr, const uint8_t* fix_pt, double huber_delta);

/* CG limits of the Iterative variant (with_cg_params, explicit_schur.rs:234-238; defaults 200, 1e-6) */
int apexgpu_set_cg_params(apexgpu_solver* h, int max_iterations, double tolerance);

/* ---- parameters ------------------------------------------------------------------------------
 * Replaces Problem::initialize_variables (src/core/problem.rs:686-808) / reading
 * SolverResult.parameters back (src/optimizer/mod.rs:250-273). */
int apexgpu_set_params(apexgpu_solver* h, const double* poses, const double* intr, const double* points);
int apexgpu_get_params(apexgpu_solver* h, double* poses, double* intr, double* points);

/* ---- hot path --------------------------------------------------------------------------------*/
/* cost = 1/2 |r~|^2 of the Huber-corrected residuals at the current parameters:
 * Problem::compute_residual_sparse + compute_cost (src/core/problem.rs:864-899,
 * src/optimizer/mod.rs:358-361). */
int apexgpu_cost(apexgpu_solver* h, double* cost);

/* AssemblyBackend::assemble (src/linearizer/mod.rs:191-213, cpu/sparse.rs:119-184) fused with
 * LinearSolver::solve_augmented_equation (src/linalg/mod.rs:143-180; explicit_schur.rs:1129-1234):
 * linearises every factor at the current parameters, forms H_cc, H_ll^-1, S, g_red on the device,
 * solves S dc = g_red and back-substitutes.  The Jacobian is never materialised.
 *   step_out  (total_dof, may be NULL): the step, global column order
 *   grad_out  (total_dof, may be NULL): +J^T r, what get_gradient() returns (:1240-1242)
 * The step also stays on the device for apexgpu_eval_step. */
int apexgpu_solve_augmented(apexgpu_solver* h, double lambda, int variant, double* step_out, double* grad_out);

/* The assembly half alone (A1-A11: H_cc, H_ll^-1, S, g_red, g on the device at the current
 * parameters, no solve) -- AssemblyBackend::assemble's slot (src/linearizer/mod.rs:191-213).  Used by
 * tests and by callers that want S / g_red (apexgpu_get_schur) without a step. */
int apexgpu_assemble(apexgpu_solver* h, double lambda);

/* out3 = { gradient.norm_l2(), step.norm_l2(), predicted reduction 1/2 step^T (lambda step - g) }
 * of the last solve (levenberg_marquardt.rs:743-746, 721-727, 890). */
int apexgpu_step_stats(apexgpu_solver* h, double out3[3]);

/* apply_parameter_step + compute_residual_sparse + compute_cost on a trial copy
 * (levenberg_marquardt.rs:776-786; src/optimizer/mod.rs:309-331). */
int apexgpu_eval_step(apexgpu_solver* h, double* trial_cost);
/* accept: the trial copy becomes current (levenberg_marquardt.rs:797-801) */
int apexgpu_commit_step(apexgpu_solver* h);
/* reject: apply_negative_parameter_step (src/optimizer/mod.rs:343-356) -- the inverse retraction of
 * the trial point, not a snapshot restore */
int apexgpu_discard_step(apexgpu_solver* h);
/* compute_parameter_norm (src/optimizer/mod.rs:458-467) */
int apexgpu_parameter_norm(apexgpu_solver* h, double* out);

/* ---- Jacobi column scaling -------------------------------------------------------------------
 * Replaces AssemblyBackend::compute_column_norms / apply_column_scaling / apply_inverse_scaling
 * (src/linearizer/mod.rs:202-209, 229-262) as process_jacobian_generic and compute_step_generic use them
 * (src/optimizer/mod.rs:749-763; levenberg_marquardt.rs:746-760).  The Jacobian is never materialised here, so
 * "scaling J" is a state of the solver:
 *   apexgpu_column_norms        norms_out[total_dof] = l2 norms of the corrected Jacobian's columns at the
 *                               current parameters, global column order (0 for columns no factor touches)
 *   apexgpu_set_column_scaling  scaling[total_dof] > 0 (the reference uses 1 / (1 + norm) of iteration 0), NULL = off.
 * While a scaling is set, apexgpu_solve_augmented solves (D J^T J D + lambda I) y = -D J^T r -- eigenvalue gate,
 * regularisation ladder and PCG tolerances in the scaled variables like the reference -- and returns
 * step_out = y and grad_out = D J^T r, exactly what LinearSolver::solve_augmented_equation / get_gradient return
 * for J D; the caller applies step = D y (apply_inverse_scaling).  On the device the unscaled step D y is kept for
 * apexgpu_eval_step, and apexgpu_step_stats prices it against the scaled gradient as compute_step_generic does.
 * apexgpu_get_schur then returns the scaled S and g_red.  apexgpu_lm_optimize with use_jacobi_scaling does all
 * of this internally (scaling from the Jacobian at the starting point, dropped when the loop returns). */
int apexgpu_column_norms(apexgpu_solver* h, double* norms_out);
int apexgpu_set_column_scaling(apexgpu_solver* h, const double* scaling);

/* ---- the LM loop (C++ twin of optimize_with_mode, levenberg_marquardt.rs:823-1031) ------------*/
typedef struct {
    int max_iterations;             /* 50   (for_bundle_adjustment: 20)            :323, :524 */
    double cost_tolerance;          /* 1e-6                                         :325 */
    double parameter_tolerance;     /* 1e-8                                         :327 */
    double gradient_tolerance;      /* 1e-10                                        :330 */
    double damping;                 /* 1e-3  (in/out: final value)                  :332 */
    double damping_min;             /* 1e-12                                        :333 */
    double damping_max;             /* 1e12                                         :334 */
    double damping_nu;              /* 2.0   (in/out)                               :337 */
    double trust_region_radius;     /* 1e4   constant, never shrinks                :338 */
    double min_trust_region_radius; /* 1e-32                                        :345 */
    double min_cost_threshold;      /* < 0: None                                    :344 */
    double timeout_s;               /* <= 0: None                                   :331 */
    int variant;                    /* APEXGPU_VARIANT_*                            :355 */
    int use_jacobi_scaling;         /* 0     with_jacobi_scaling                    :352, :474 */
} apexgpu_lm_config;

typedef struct {
    double cost, damping, rho, accepted, gradient_norm, step_norm, predicted_reduction, trial_cost;
} apexgpu_lm_iter;

typedef struct {
    int status;     /* OptimizationStatus discriminant (src/optimizer/mod.rs:189-216); 100 = linear solve failed */
    int iterations; /* SolverResult.iterations = last iteration index + 1 */
    double initial_cost, final_cost, final_gradient_norm, final_step_norm, elapsed_s;
    int cost_evaluations, jacobian_evaluations, successful_steps, unsuccessful_steps;
} apexgpu_lm_result;

int apexgpu_lm_optimize(apexgpu_solver* h, apexgpu_lm_config* cfg, apexgpu_lm_result* result,
                        apexgpu_lm_iter* history, int history_capacity);

/* ---- parity / debug exports (observers' set_matrix_data analogue, src/observers/mod.rs:201-260) */
int apexgpu_get_residual(apexgpu_solver* h, double* r_out /* 2*n_obs, caller's factor order */);
/* corrected blocks per factor: jc_out[n_obs][2][d_c] (pose columns first, then intrinsics when
 * d_c = 9), jl_out[n_obs][2][3] */
int apexgpu_get_jacobian_blocks(apexgpu_solver* h, double* jc_out, double* jl_out);
/* dense S ((9 n_cam)^2 row-major, symmetric) and g_red (9 n_cam) in the reference's camera-side
 * column order, for the last lambda; either may be NULL */
int apexgpu_get_schur(apexgpu_solver* h, double* S_out, double* gred_out);
int apexgpu_get_landmark_blocks(apexgpu_solver* h, double* hinv_out /* n_pt*9 */, double* gl_out /* n_pt*3 */);
/* Parity probe of row A9: invert_landmark_blocks_with_lambda(.., 0.0) (explicit_schur.rs:365-442) -- the eigenvalue gate
 * (min_ev < 1e-12 -> + (1e-6 + max_ev 1e-6) I; max_ev / min_ev > 1e10 -> + max_ev 1e-6 I; else plain inverse) exactly as
 * the landmark-reduce kernel applies it, run on GPU `device` over n caller-supplied symmetric 3x3 blocks (row-major).
 * ok_out[i] = 0 where the (regularised) block has a zero determinant (LinAlgError::SingularMatrix). */
int apexgpu_debug_invert_blocks(int device, int64_t n, const double* blocks9, double* inv9_out, int32_t* ok_out);
/* LinearSolver::get_hessian (src/linalg/mod.rs:158; SparseSchurComplementSolver caches it, explicit_schur.rs:1146-1160,
 * 1236-1238; consumed by the observers, src/optimizer/mod.rs:701-720, and by DogLeg): H = J^T J of the corrected
 * Jacobian at the current parameters -- undamped, full symmetric, CSC, global column order, total_dof = 9 n_cam + 3 n_pt
 * columns (the intrinsics' columns are empty in BundleAdjustment mode).  The device never forms H: this export
 * rebuilds it from the per-factor blocks, on demand.  Two calls: with colptr_out == NULL only *nnz_out is written;
 * then colptr_out[total_dof + 1], rowidx_out[nnz], values_out[nnz]. */
int apexgpu_get_hessian_csc(apexgpu_solver* h, int64_t* nnz_out, int64_t* colptr_out, int64_t* rowidx_out, double* values_out);
/* y = S x at the current parameters through both implementations of the reduced camera matrix: the explicit tiles
 * (compute_schur_complement, explicit_schur.rs:771-925) and the matrix-free operator (apply_schur_operator_fast,
 * implicit_schur.rs:163-251).  x_in and the outputs have 9 n_cam entries in the reference's camera-side column
 * order; either output may be NULL.  A size-independent parity property: the two must agree. */
int apexgpu_schur_matvec(apexgpu_solver* h, double lambda, const double* x_in, double* y_explicit, double* y_implicit);

/* Implementation switches (defaults in parentheses).  Round 6 removed every switch whose committed A/B said "loses" or
 * "neutral", with its code (the row form of the Schur reduction, landmark bundles, the queued layout for six-column cameras, the
 * pair kernel's ablations, camera-major records for the matrix-free operator, the vector-unit potrf kernels, dynamic scheduling
 * of the dataflow units, panel lookahead, the forward sweep beside / inside the factorisation, tile clears on side streams, the
 * small-batch limits; the winners "first_writer", "tri_inline", "panel_tri", "factor_flow_tile", "device_gathers", "cam_staging"
 * are what the library does): measurements in profiles/, code in the history (DESIGN_HISTORY.md names the commits).  What is
 * left selects between implementations that are BOTH in use somewhere, shapes the structure, or serves the tests.
 * Switches that shape what apexgpu_set_structure builds or what the captured hipGraphs hold return APEXGPU_ERR_INVALID_STATE
 * once the structure is set ("schur_form", "hubs_last", "nested_dissection", "dist_factor", "tree_sharding", "dist_selftest",
 * "update_overlap", "split_u1", "flood_gate", "two_side", "factor_flow", "factor_flow_rows", "device_pair_list",
 * "matrix_free_only", "auto_variant", "variant_cost_permille", "max_tile_updates").
 *   "schur_form" (4)  layout of the sorted pair list of the Schur reduction: 4 = QUEUED (nine columns per camera; six-column
 *                     cameras always get 3): a task is the blocks of one row padded to nonets of nine slots and cut into seven
 *                     queues, every lane group of the product phase owns a block of its own -- nine fixed steps per chunk, no
 *                     fold over the groups, a block cut between two queues carried in registers and stored once
 *                     (k_schur_pairs_r<.., QL>, csrc/schur_pairs.h); 3 = every camera pair of a landmark in a list sorted by
 *                     the block S(ci, cj) it adds to, the lane groups share ONE running block and fold at its end.  Other
 *                     values (the forms of rounds 1-3) answer APEXGPU_ERR_INVALID_INPUT
 *   "graphs"     (1)  replay the factorisation / triangular solves as captured hipGraphs
 *   "update_overlap" (1)  run the trailing updates the next elimination level does not need on a second
 *                     stream, overlapped with that level's potrf / panel solves; > 1: the smallest batch that moves there
 *   "split_u1" (4)    the updates a level sends into the NEXT level's columns are split: those of its diagonal tiles stay on
 *                     the main stream (the next potrf needs nothing else), the others run on a third stream beside that
 *                     potrf when they are at least this many tasks; 0 = one batch on the main stream
 *   "flood_gate" (256)  the bulk updates of a level of at least this many tasks start when the next level's potrf workgroups
 *                     sit on their CUs (a one-lane gate kernel; 0: off)
 *   "two_side" (1)    the bulk updates on two side streams (targets two / three levels up; four and more): 0 off, 1 by plan
 *                     size, 2 always (tests)
 *   "matrix_free_only" (0)  the handle will only be asked for variant 2 (IterativeSchurSolver semantics): S is never formed,
 *                     so only its diagonal tiles are allocated and no pair list is built -- set-up and LM iteration are then
 *                     independent of the fill of S; variants 0 / 1 and the exports of S answer APEXGPU_ERR_INVALID_STATE
 *   "one_wait" (1)    single rank, Cholesky variant: apexgpu_solve_augmented enqueues factorisation, sweeps and back-substitution
 *                     back to back and waits for the device ONCE (the landmark-inversion and pivot flags are read at that
 *                     wait; a failure repeats the solve on the old path, ladder included); 0: three waits as in rounds 1-4
 *   "eager_step_eval" (1)  single rank: the step statistics and the trial point with its cost -- what the LM loop asks next of
 *                     every solve -- are enqueued behind the back-substitution and read at the solve's own wait; apexgpu_step_stats
 *                     and apexgpu_eval_step then answer from the host (three device round trips per LM iteration become one).
 *                     A caller that never asks (a level-1 binding that only takes the step) sets 0 and saves the kernels
 *   "device_pair_list" (1)  the RECORDS of the queued pair list (1.56 GB on final-13682) are written by the device from the
 *                     observation lists (k_build_pair_recs_q: the host keeps the small, serial part -- blocks, tasks,
 *                     descriptors); 0 = built on the host and copied: the same list, slot for slot (the tests hold one against
 *                     the other), 0.1-0.18 s + 35 ms of upload more per apexgpu_set_structure
 *   "auto_variant" (1)  a structure whose direct factorisation is refused -- more than 8e7 tile products
 *                     per factorisation (S dense at tile granularity: a photo collection), or, on a single rank, tiles beyond
 *                     the free HBM -- does NOT fail apexgpu_set_structure: the handle is built matrix-free only by itself and
 *                     apexgpu_solve_augmented / apexgpu_lm_optimize answer variants 0 and 1 with the matrix-free PCG (variant 2;
 *                     for variant 0 at IterativeSchurSolver's defaults, 500 iterations / 1e-9, for variant 1 at the caller's
 *                     cg parameters).  Round 6: the same happens when the plan is PREDICTED to cost more per solve than the
 *                     matrix-free PCG at its cap (apexgpu_variant_costs) -- between 0.5 s and the 12.7 s of the size limit the direct
 *                     path used to be chosen although the handle owns a 0.5-s way to the same step.  apexgpu_variant_info tells.
 *                     0: a refusal is APEXGPU_ERR_INVALID_INPUT as before, and no plan is refused by cost
 *   "variant_cost_permille" (1000)  scales the matrix-free side of that comparison (tests move the crossover
 *                     onto small problems; 0 switches the cost rule off)
 *   "max_tile_updates" (80000000)  tests: the limit of tile products per factorisation above which the plan is refused
 *   "factor_flow" (-1), "factor_flow_rows" (24)  the TOP of the elimination tree -- the trailing level groups with at most
 *                     that many tile columns each, every column with at most "factor_flow_rows" off-diagonal tiles -- is
 *                     factorised by ONE dataflow launch (-1, the default: where the launch starts is chosen by a cost
 *                     model of both schedules) (k_factor_flow: a workgroup per potrf / per 48-row strip of a
 *                     panel solve or update on the chain, per whole tile off it; per-tile version counters) instead of three
 *                     dependent launches per level; same summation order, bit-identical factor.  0 = level launches
 *                     everywhere.  Its waits are bounded like the sweeps': a launch that gives up is detected in the same
 *                     apexgpu_solve_augmented, S is assembled again and factorised by the level launches, which the handle
 *                     then keeps (apexgpu_counters()[2])
 *   "tri_dataflow" (1)  triangular sweeps of a single-GPU plan as ONE launch each: one workgroup per tile, dependencies
 *                     through per-block flags (k_tri_fwd_flow / k_tri_bwd_flow); 0 = one launch per elimination-tree level.
 *                     The flag waits are bounded (~2 s): a sweep that gives up is detected IN THE SAME apexgpu_solve_augmented
 *                     (error word posted to the host behind the sweeps, max-reduced over the ranks), the solve is repeated
 *                     with the level sweeps and the handle stays on them; apexgpu_counters()[0] counts these events
 *   "debug_poison_factor"  tests only: the next factorisation's dataflow launch ("factor_flow") cannot finish and times out
 *   "debug_poison_sweep", "debug_occupy_cus"  tests only: the next solve's forward (1) / backward (2) dataflow sweep runs
 *                     into its spin limit on purpose; block that many compute units for 40 ms starting now
 *   "nested_dissection" (1)  order camera tiles by nested dissection;
 *                     0 keeps the caller's camera order, a value > 1 sets the leaf size in tiles
 *   "hubs_last" (1)   order a vertex cover of the camera pairs that share landmarks across tiles more than two strong
 *                     hops apart (hub cameras, accidental long-range matches) after all the others: a dense border of S
 *                     instead of fill everywhere (csrc/ba_structure.h)
 *   "dist_factor" (1), "tree_sharding" (1)  multi-GPU only: see the multi-GPU section
 *   "dist_selftest" (0)  single rank, before set_structure: cut the elimination tree as for that many ranks and run
 *                     the distributed schedule (own levels, top levels, phased triangular sweeps) with this rank
 *                     owning every subtree and no-op exchanges -- results must equal the plain schedule's      */
int apexgpu_set_option(apexgpu_solver* h, const char* name, int value);

/* ---- measurement ------------------------------------------------------------------------------*/
#define APEXGPU_NUM_STAGES 10
/* stage order: cam-reduce(+memset), landmark-reduce, schur-scatter, all-reduce, factor (or PCG),
 * triangular solves, back-substitute, step-stats, retract, cost */
/* on: 0 off, 1 every stage, > 1: bit k + 1 set = stage k alone is timed (e.g. 2 << 2: the Schur kernel only -- what bench.py
 * keeps inside its timed region; every stage event costs the stream a few microseconds) */
int apexgpu_enable_stage_timing(apexgpu_solver* h, int on);
int apexgpu_reset_stage_times(apexgpu_solver* h);
int apexgpu_stage_times(apexgpu_solver* h, double ms[APEXGPU_NUM_STAGES], int64_t calls[APEXGPU_NUM_STAGES]);
/* info[0] = S tile rows, [1] = allocated tiles, [2] = camera-pair contributions per Schur reduce,
 * [3] = internal camera DOF, [4] = regularisation used by the last Cholesky, [5] = PCG iterations,
 * [6] = S tiles that receive Schur contributions (before fill), [7] = observations on this rank,
 * [8] = levels of the tile elimination tree (= dependent launch groups of the factorisation),
 * [9..11] = tile operations per factorisation: potrf, panel products, trailing updates (2*144^3 flop each for the last two),
 * [12] = shared top tile columns of a distributed factorisation (0: replicated), [13] = this rank's share of the tile
 * operations below them (1 when not distributed), [14] = 1 when the landmarks are sharded by the elimination tree,
 * [15] = form of the Schur reduction in use ("schur_form") */
int apexgpu_info(apexgpu_solver* h, double info[16]);
/* Which variant a solve asked with `asked_variant` runs on this handle (*used_variant; differs only after the automatic
 * selection of "auto_variant") and why (reason: NUL-terminated, cut to reason_len; empty when nothing was overridden).
 * The reference never needs it: its LM dispatch (src/optimizer/levenberg_marquardt.rs:1039-1082) does not depend on the
 * fill of S, and IterativeSchurSolver (src/linalg/sparse/implicit_schur.rs:835-946) is the solver the fall-back restates. */
int apexgpu_variant_info(apexgpu_solver* h, int asked_variant, int* used_variant, char* reason, int reason_len);
/* What apexgpu_set_structure predicted and chose (round 6): out[0] = milliseconds per solve of the direct path (tile Cholesky
 * + both sweeps; from the plan's operation counts at the rates DESIGN.md section 5 measures; 0 on a handle built with
 * "matrix_free_only"), out[1] = of the matrix-free PCG at IterativeSchurSolver's cap (out[3] = 500 iterations x one S p = two
 * passes over the observations), out[2] = the choice: 0 direct, 1 matrix-free by predicted cost (out[0] > out[1]), 2 matrix-free
 * because the plan was refused (size / memory), 3 matrix-free by the caller's option. */
int apexgpu_variant_costs(apexgpu_solver* h, double out[4]);
/* out[0] = dataflow triangular sweeps that timed out and were repeated level by level (see "tri_dataflow"),
 * out[1] = 1 while the handle still uses the dataflow sweeps, out[2] = dataflow factorisations that timed out and were
 * repeated by the level launches (see "factor_flow"), out[3] = level groups inside the dataflow launches of this plan */
int apexgpu_counters(apexgpu_solver* h, int64_t out[4]);
/* Tests: the records of the sorted pair list as they sit on the device, recs4_out[slots][4] = {i, j, landmark, queue}
 * (i = 0xFFFFFFFF: padding; slots = counts[2] of apexgpu_setup_times). */
int apexgpu_debug_get_pair_records(apexgpu_solver* h, uint32_t* recs4_out, int64_t cap_slots);
/* Wall time of the last apexgpu_set_structure by phase, seconds[6] = {camera order + tile structure, landmark sharding +
 * observation lists, tile plan (symbolic fill, task lists, allocation), lists of the Schur reduction, uploads, total};
 * counts[4] (may be NULL) = {hub cameras ordered last, camera-pair blocks, pair slots incl. padding, Schur form}. */
int apexgpu_setup_times(apexgpu_solver* h, double seconds[6], double counts[4]);

/* ---- multi-GPU: one process per GPU, landmarks sharded, RCCL all-reduce of S and g_red -----------
 * apexgpu_get_unique_id fills 128 bytes on rank 0; broadcast them (any transport) and call
 * apexgpu_comm_init on every rank BEFORE apexgpu_set_structure.  apexgpu_set_shard alone (no
 * communicator) restricts the assembly to this rank's landmark range, for tests that sum the
 * partial S / g_red themselves.
 * With world > 1 the Cholesky of S is distributed as well (option "dist_factor", default 1; set 0 before
 * apexgpu_set_structure for a factorisation replicated on every rank): the nested-dissection elimination tree is cut
 * into `world` groups of subtrees below a shared top; a rank factorises its own subtrees, the updates to the top
 * tiles are summed in one all-reduce (those tiles are left out of the all-reduce of S), the top columns are factorised
 * by every rank, and the triangular sweeps exchange two n-vectors (csrc/tile_plan.h). */
int apexgpu_get_unique_id(void* out128);
int apexgpu_comm_init(apexgpu_solver* h, int world, int rank, const void* unique_id128);
/* The same multi-rank schedule over HOST SHARED MEMORY instead of RCCL (csrc/comm.h): the ranks are processes of one node --
 * they may share one GPU -- and every collective is staged through a POSIX shared-memory segment named after `name`
 * (a string common to the ranks of ONE run and to no other), summed in rank order.  For bring-up and tests: every
 * world > 1 branch of the library runs with the real kernels on a single-GPU box, where RCCL cannot be initialised with
 * two ranks.  Same call order and error convention as apexgpu_comm_init (before apexgpu_set_structure). */
int apexgpu_comm_init_shm(apexgpu_solver* h, int world, int rank, const char* name);
int apexgpu_set_shard(apexgpu_solver* h, int rank, int world);
/* Test entry for the distributed solve without a communicator: hs[0..n) are the ranks of ONE sharded problem
 * (apexgpu_set_shard(r, n), same structure and parameters) living in this process on one GPU; one Cholesky solve of
 * (S, g_red) at `lambda` runs in lockstep, the function itself playing the all-reduces between the phases.  Every
 * handle then holds the step (apexgpu_export_step: camera part complete, landmark part for the rank's own range). */
int apexgpu_debug_lockstep_solve(apexgpu_solver** hs, int n, double lambda);
/* mask[n_pt] (caller's numbering): 1 for the landmarks this rank assembles and back-substitutes.  Contiguous
 * apexgpu_shard_range ranges with a replicated factorisation ("dist_factor" = 0) or with option "tree_sharding" = 0;
 * by default (world > 1) a landmark belongs to the rank whose tile columns its cameras touch below the shared top of
 * the elimination tree, which makes that rank's tiles of S complete without any reduction. */
int apexgpu_owned_landmarks(apexgpu_solver* h, uint8_t* mask);
/* Host arithmetic only: the cut of the tile elimination tree a distributed plan of `world` ranks makes for the
 * lower-triangular tile structure present[nt*nt] (row-major, I >= J).  owner_out[nt] = owning rank of every tile
 * column, -1 for the shared top; returns the number of top columns (0: the plan stays replicated) or a negative
 * status. */
int apexgpu_debug_partition(int nt, const uint8_t* present, int world, int* owner_out);
/* Host-only proof that the factorisation's launch sequence is race free for one tile structure (no device is touched):
 * the plan is built on made-up addresses, TilePlan::enqueue_factor records its launches / event records / stream waits
 * instead of issuing them, and every two launches that touch one tile with a writer among them must be ordered by stream
 * order + event edges; tasks of one launch must not share a written tile.  present: lower-triangular nt x nt structure in
 * the final tile order.  opts = {two_side, update_overlap (>1: minimum batch), split_u1, flood_gate, factor_flow,
 * factor_flow_rows, tests: bring back the round-3 idle-level bug, tests: drop that stream wait of phase 0 (-1: none)}.
 * out = {calls, launches, violations of phase 0 (local level groups / everything), violations of phase 1 (shared top of a
 * distributed plan), dataflow units, level groups inside dataflow launches, stream waits, 1 if a wait was dropped}.
 * Returns the number of level groups (>= 0) or an error; msg receives the first violation.  (round-3 advice: the U2 split
 * had dropped the edge behind a level without side-stream work; the checker finds it on the advisor's pattern.) */
int apexgpu_debug_check_schedule(int nt, const uint8_t* present, int world, int rank, const int opts[8], int64_t out[8], char* msg, int msg_len);
/* Host arithmetic only: the sorted camera-pair lists of the default Schur reduction for an observation list, with the
 * caller's camera order and a dense tile map (slot(I, J) = I (I + 1) / 2 + J).  counts[4] = {slots, chunks, blocks, tasks};
 * outputs may be NULL (size query): recs4 [slots][4] = {i, j, landmark, block local to the chunk} (i = 0xFFFFFFFF:
 * padding), chunks2 [chunks][2] = {K-step mask of block starts, first block}, blocks4 [blocks][4] = {offset of
 * S(ci, cj) in the tile storage, ci, cj, flags}, tasks2 [tasks][2] = {first chunk, chunks}; o_index[n_obs]: caller's index
 * of landmark-major observation k (what i / j count in). */
/* Host arithmetic only: the whole host half of apexgpu_set_structure for rank `rank` of `world` (camera order with hub
 * cameras last, nested dissection, symbolic fill, partition of the elimination tree, landmark sharding, Schur lists).
 * opts[5] = {nested_dissection, hubs_last, dist_factor, tree_sharding, schur_form}; stats_out[16] = {tile rows, hub
 * cameras, border tiles, tiles S touches, tiles after fill, elimination-tree levels, shared top columns, tree sharded,
 * seconds[6] as apexgpu_setup_times (device phases 0), pair contributions, camera-pair blocks}; cmap_out[n_cam],
 * owned_out[n_pt], tile_owner_out[tile rows] may be NULL. */
int apexgpu_debug_host_structure(int64_t n_cam, int64_t n_pt, int64_t n_obs, int mode, const uint32_t* cam_idx,
                                 const uint32_t* pt_idx, int rank, int world, const int opts[5], double stats_out[16],
                                 int32_t* cmap_out, uint8_t* owned_out, int32_t* tile_owner_out);
int apexgpu_debug_pair_lists(int64_t n_cam, int64_t n_pt, int64_t n_obs, int dc, const uint32_t* cam_idx, const uint32_t* pt_idx,
                             int64_t counts[4], uint32_t* recs4_out, int32_t* chunks2_out, int64_t* blocks4_out,
                             int32_t* tasks2_out, int32_t* o_index_out);
/* ... and in the QUEUED layout ("schur_form" 4, nine columns per camera; csrc/schur_pairs.h): qdesc3_out[8 * chunks][3] =
 * {dst, cj, flags} of every (chunk, queue), entry 7 of a chunk carrying the row's camera; chunks2_out[.][0] = flush bits. */
int apexgpu_debug_pair_lists_queued(int64_t n_cam, int64_t n_pt, int64_t n_obs, const uint32_t* cam_idx, const uint32_t* pt_idx,
                                    int64_t counts[4], uint32_t* recs4_out, int32_t* chunks2_out, int64_t* blocks4_out,
                                    int32_t* tasks2_out, int32_t* o_index_out, int64_t* qdesc3_out);
/* The same for either camera width (round 5): dc = 9 as above; dc = 6 (BundleAdjustment mode): sixteen queues of four pairs per
 * chunk -- slot g + 16 t = pair t of queue g --, qdesc3_out[17 * chunks][3], entry 16 of a chunk = the row's camera. */
int apexgpu_debug_pair_lists_queued_dc(int64_t n_cam, int64_t n_pt, int64_t n_obs, int dc, const uint32_t* cam_idx, const uint32_t* pt_idx,
                                       int64_t counts[4], uint32_t* recs4_out, int32_t* chunks2_out, int64_t* blocks4_out,
                                       int32_t* tasks2_out, int32_t* o_index_out, int64_t* qdesc3_out);
int apexgpu_export_step(apexgpu_solver* h, double* step_out, double* grad_out);
/* The landmark range [lo,hi) rank `rank` of `world` owns (contiguous, balanced by observation count).
 * Host arithmetic only -- no device is touched -- so schedulers and tests can call it anywhere. */
int apexgpu_shard_range(int64_t n_pt, int64_t n_obs, const uint32_t* pt_idx, int rank, int world, int64_t* lo,
                        int64_t* hi);

/* ---- input path (host only, no GPU needed): BAL files and the reference's variable order ---------
 * BalLoader::load (crates/apex-io/src/bal.rs:138-202): header "n_cam n_pt n_obs", n_obs lines
 * "cam pt x y", 9 lines per camera (rx ry rz tx ty tz f k1 k2), 3 lines per point; blank lines are
 * skipped; a non-positive or non-finite focal length becomes 500 (:100-114).  Error codes mirror
 * IoError: Io, Parse, MissingFields, InvalidNumber; text via apexgpu_bal_last_error(). */
typedef struct apexgpu_bal apexgpu_bal;
#define APEXGPU_BAL_ERR_IO (-20)
#define APEXGPU_BAL_ERR_PARSE (-21)
#define APEXGPU_BAL_ERR_MISSING_FIELDS (-22)
#define APEXGPU_BAL_ERR_INVALID_NUMBER (-23)
int apexgpu_bal_open(const char* path, apexgpu_bal** out);
void apexgpu_bal_close(apexgpu_bal* b);
const char* apexgpu_bal_last_error(void);
int apexgpu_bal_sizes(const apexgpu_bal* b, int64_t* n_cam, int64_t* n_pt, int64_t* n_obs);
/* raw file contents: cameras9[n_cam][9] = rx ry rz tx ty tz f k1 k2; any pointer may be NULL */
int apexgpu_bal_raw(const apexgpu_bal* b, uint32_t* cam_idx, uint32_t* pt_idx, double* obs_uv, double* cameras9,
                    double* points3);
/* the variables run_bundle_adjustment creates (bin/bundle_adjustment.rs:200-208, 232-257):
 * poses7[n_cam][7] = [t, qw,qx,qy,qz] from the axis-angle, intr3[n_cam][3] = [f,k1,k2] */
int apexgpu_bal_variables(const apexgpu_bal* b, double* poses7, double* intr3);
/* first global column of intr_{i:04} / pose_{i:04} / pt_{j:05} in the sorted-name order of
 * src/optimizer/mod.rs:530-536 (what apexgpu_set_structure expects) */
int apexgpu_reference_columns(int64_t n_cam, int64_t n_pt, int64_t* intr_col, int64_t* pose_col, int64_t* pt_col);

/* =================================================================================================
 * SE3 pose-graph backend (BASELINE.json configs[1]: block-sparse J^T J assembly + Cholesky, no Schur)
 *
 * Replaces, for problems made of BetweenFactor<SE3> residual blocks (bin/pose_graph_g2o.rs:748-830):
 *   BetweenFactor<SE3>::linearize                         src/factors/between_factor.rs:268-322
 *   SparseCholeskySolver::solve_augmented_equation        src/linalg/sparse/cholesky.rs:159-230
 *   the LM loop around them                               src/optimizer/levenberg_marquardt.rs:823-1031
 * Edge information matrices are NOT used (the reference passes only edge.measurement,
 * bin/pose_graph_g2o.rs:805-826).  Vertices are numbered by the caller 0..n_v-1 (the reference sorts
 * the vertex ids); pose_col[v] is the first global column of variable "x{id_v}" in the sorted-name order
 * (apexgpu_pose_graph_columns); tangent order [rho(3); theta(3)], right-plus retraction as above.
 * The same status codes; messages via apexgpu_pg_last_error().
 * ================================================================================================= */
typedef struct apexgpu_pg_solver apexgpu_pg_solver;

/* SparseCholeskySolver::new (cholesky.rs:60-70) + per-optimize state */
int apexgpu_pg_create(int64_t n_vertices, int64_t n_edges, int device, apexgpu_pg_solver** out);
void apexgpu_pg_destroy(apexgpu_pg_solver* h);
const char* apexgpu_pg_last_error(const apexgpu_pg_solver* h);

/* build_symbolic_structure (src/linearizer/cpu/sparse.rs:54-105) + the cached SymbolicLlt
 * (cholesky.rs:190-208): e_from[e] = k0 and e_to[e] = k1 of BetweenFactor e (residual block order),
 * meas7[e] = [t, qw,qx,qy,qz] of the measured k0->k1 transform, fix6[v][a] != 0 fixes tangent DOF a of
 * vertex v (Problem::fix_variable, src/core/problem.rs:185-197), huber_delta <= 0: no loss function,
 * > 0: HuberLoss(delta) on every block. */
int apexgpu_pg_set_structure(apexgpu_pg_solver* h, const uint32_t* e_from, const uint32_t* e_to, const double* meas7,
                             const int64_t* pose_col, const uint8_t* fix6, double huber_delta);
/* Problem::initialize_variables (src/core/problem.rs:686-808): poses7[v] = [t, qw,qx,qy,qz] */
/* PriorFactor blocks on SE3 variables (reference: src/factors/prior_factor.rs:96-108; the gauge of
 * tests/integration_tests.rs:98-118): r = to_vector(x_vertex) - data7 (seven rows, [t, w, i, j, k]); the Jacobian is the 7 x 7
 * identity of which the linearizer keeps the variable's six tangent columns (src/linearizer/cpu/sparse.rs:201-204);
 * huber_delta[k] <= 0 (or huber_delta == NULL): no loss on block k.  Replaces the set; after apexgpu_pg_set_structure.
 * apexgpu_pg_get_prior_residual: the corrected residuals [n][7] at the current parameters. */
int apexgpu_pg_set_priors(apexgpu_pg_solver* h, int64_t n, const uint32_t* vertex, const double* data7, const double* huber_delta);
int apexgpu_pg_get_prior_residual(apexgpu_pg_solver* h, double* r7_out);
int apexgpu_pg_set_params(apexgpu_pg_solver* h, const double* poses7);
int apexgpu_pg_get_params(apexgpu_pg_solver* h, double* poses7);

/* compute_cost on the current parameters (src/optimizer/mod.rs:358-361) */
int apexgpu_pg_cost(apexgpu_pg_solver* h, double* cost);
/* assemble (src/linearizer/mod.rs:191-213) + solve_augmented_equation (cholesky.rs:159-230):
 * (J^T J + lambda I) dx = -J^T r.  step_out / grad_out (= +J^T r, get_gradient) have 6 n_vertices entries in
 * the global column order and may be NULL.  A non-positive pivot returns APEXGPU_ERR_SINGULAR_MATRIX
 * ("Cholesky factorization failed (matrix may be singular)", cholesky.rs:213-219). */
int apexgpu_pg_solve_augmented(apexgpu_pg_solver* h, double lambda, double* step_out, double* grad_out);
/* out3 = { |g|, |step|, predicted reduction } (levenberg_marquardt.rs:721-746) */
int apexgpu_pg_step_stats(apexgpu_pg_solver* h, double out3[3]);
/* apply_parameter_step into a trial set + its cost; commit = accept; discard = apply_negative_parameter_step
 * (src/optimizer/mod.rs:309-356) */
int apexgpu_pg_eval_step(apexgpu_pg_solver* h, double* trial_cost);
int apexgpu_pg_commit_step(apexgpu_pg_solver* h);
int apexgpu_pg_discard_step(apexgpu_pg_solver* h);
int apexgpu_pg_parameter_norm(apexgpu_pg_solver* h, double* out);
/* optimize_with_mode (levenberg_marquardt.rs:823-1031), device-resident; cfg->variant must be 0 */
/* Jacobi column scaling, as apexgpu_column_norms / apexgpu_set_column_scaling */
int apexgpu_pg_column_norms(apexgpu_pg_solver* h, double* norms_out);
int apexgpu_pg_set_column_scaling(apexgpu_pg_solver* h, const double* scaling);
int apexgpu_pg_lm_optimize(apexgpu_pg_solver* h, apexgpu_lm_config* cfg, apexgpu_lm_result* result,
                           apexgpu_lm_iter* history, int history_capacity);

/* parity / debug exports: loss-corrected residuals [n_edges][6], Jacobians [n_edges][6][12] = [dr/dk0 | dr/dk1]
 * in residual-block order; dense H = J^T J + lambda I ([6 n_v]^2 row-major) and g = J^T r in the global column order */
int apexgpu_pg_get_residual(apexgpu_pg_solver* h, double* r_out);
int apexgpu_pg_get_jacobian_blocks(apexgpu_pg_solver* h, double* j_out);
int apexgpu_pg_get_hessian(apexgpu_pg_solver* h, double lambda, double* H_out, double* g_out);

/* name: "graphs" (hipGraph replay of factor / solves), "update_overlap" (second stream for trailing updates),
 * "tri_dataflow" (triangular sweeps as one dataflow launch each), "nested_dissection" (0 off, 1 on, > 1 leaf size; before apexgpu_pg_set_structure) */
int apexgpu_pg_set_option(apexgpu_pg_solver* h, const char* name, int value);
#define APEXGPU_PG_NUM_STAGES 6 /* assemble, factor, tri_solve, step_stats, retract, cost */
int apexgpu_pg_enable_stage_timing(apexgpu_pg_solver* h, int on);
int apexgpu_pg_reset_stage_times(apexgpu_pg_solver* h);
int apexgpu_pg_stage_times(apexgpu_pg_solver* h, double ms[APEXGPU_PG_NUM_STAGES], int64_t calls[APEXGPU_PG_NUM_STAGES]);
/* info[0] tile rows, [1] tiles incl. fill, [2] tiles of H itself, [3] elimination-tree levels, [4] total dof,
 * [5..7] tile operations per factorisation: potrf, panel products, trailing updates */
int apexgpu_pg_info(apexgpu_pg_solver* h, double info[8]);
int apexgpu_pg_counters(apexgpu_pg_solver* h, int64_t out[4]);   /* as apexgpu_counters */

/* ---- input path (host only): G2O files ------------------------------------------------------------
 * G2oLoader::load (crates/apex-io/src/g2o.rs:140-620): VERTEX_SE3:QUAT id x y z qx qy qz qw (norm checked to
 * 1 +- 0.01, then normalised), EDGE_SE3:QUAT from to x y z qx qy qz qw + 21 upper-triangular information
 * values; '#' comments, blank lines and unknown tags are skipped; SE2 lines are validated and counted only.
 * Error codes mirror IoError: Io, Parse, MissingFields, InvalidNumber, DuplicateVertex, InvalidQuaternion. */
typedef struct apexgpu_g2o apexgpu_g2o;
#define APEXGPU_G2O_ERR_IO (-30)
#define APEXGPU_G2O_ERR_PARSE (-31)
#define APEXGPU_G2O_ERR_MISSING_FIELDS (-32)
#define APEXGPU_G2O_ERR_INVALID_NUMBER (-33)
#define APEXGPU_G2O_ERR_DUPLICATE_VERTEX (-34)
#define APEXGPU_G2O_ERR_INVALID_QUATERNION (-35)
int apexgpu_g2o_open(const char* path, apexgpu_g2o** out);
void apexgpu_g2o_close(apexgpu_g2o* g);
const char* apexgpu_g2o_last_error(void);
int apexgpu_g2o_sizes(const apexgpu_g2o* g, int64_t* n_vertices_se3, int64_t* n_edges_se3, int64_t* n_vertices_se2,
                      int64_t* n_edges_se2);
/* file order; poses7 / meas7 = [t, qw,qx,qy,qz]; e_from / e_to are vertex IDS; info36 row-major; any may be NULL */
int apexgpu_g2o_raw(const apexgpu_g2o* g, int64_t* ids, double* poses7, int64_t* e_from, int64_t* e_to, double* meas7,
                    double* info36);
/* the problem bin/pose_graph_g2o.rs:748-830 builds with the LM optimiser: vertices sorted by id, edge endpoints as
 * indices into that order, pose_col from apexgpu_pose_graph_columns, all 6 DOF of the first vertex fixed */
int apexgpu_g2o_problem(const apexgpu_g2o* g, int64_t* sorted_ids, double* poses7, uint32_t* e_from, uint32_t* e_to,
                        double* meas7, int64_t* pose_col, uint8_t* fix6);
/* first global column of variable "x{ids[v]}" in the sorted-name order of src/optimizer/mod.rs:530-536 */
int apexgpu_pose_graph_columns(int64_t n_vertices, const int64_t* ids, int64_t* pose_col);

#ifdef __cplusplus
}
#endif
#endif /* APEXGPU_H */
