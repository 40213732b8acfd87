// ba_kernels.h -- host-visible declarations of the HIP kernels' launchers and argument PODs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace apex {

constexpr int kNB = 144;          // S tile edge: 16*9 = 24*6, multiple of the 16-wide f64 MFMA
constexpr int kScatterCap = 128;  // observations one Schur-scatter workgroup stages in LDS
constexpr int kScatterBlk = 64;   // block size used to split landmarks with more observations

// Read-only view of one parameter set + the landmark-major observation stream.
struct BAView {
    int64_t n_cam, n_pt, n_obs;
    const double* camp;    // [n_cam][16] prepared cameras (R t f k1 k2), see k_prepare_cams
    const double* camq;    // [n_cam][10] the same cameras as (unit quaternion, t, f k1 k2): per-lane gathers (ba_device.hpp)
    const double* pts;     // [n_pt][3]
    const uint32_t* o_cam; // [n_obs] landmark-major
    const uint32_t* o_pt;  // [n_obs]
    const double2* o_uv;   // [n_obs]
    const int* pt_ptr;     // [n_pt+1]
    double huber_delta;
    int mask_code;         // 4 POSE + 2 LANDMARK + INTRINSIC (OptimizeParams); 7 (6 at d_c = 6): no column group is masked
    const uint32_t* co_pt; // [n_obs] camera-major copies (entry k of the camera lists): landmark index
    const double2* co_uv;  // [n_obs]                                                   and measurement
    const int* co_rank;    // [n_obs] position of the observation inside its landmark's list
    // Jacobi column scaling (optimizer/mod.rs:749-763), NULL when off.  The kernels keep working on the
    // UNSCALED blocks: (D J^T J D + lambda I) y = -D J^T r  <=>  (J^T J + lambda D^-2) (D y) = -J^T r, so the
    // damping becomes lambda / s^2 per column and S is rescaled once after the reduction (solver.hip).
    const double* cam_scale;  // [n_cam][d_c]
    const double* pt_scale;   // [n_pt][3]
    // which rank adds lambda to a camera's diagonal block: NULL = rank 0 for every camera (all partial S are summed);
    // tree sharding: the owner of the camera's tile column (its tiles are never summed), rank 0 for the shared top
    const uint8_t* lam_mask;  // [n_cam]
    // Camera staging of the landmark-major kernels (k_landmark_reduce, k_back_substitute; NULL = off): a workgroup handles
    // kLmWg consecutive landmarks, whose observations see a few dozen DISTINCT cameras; the host lists them per workgroup
    // (wg_cam_list[w][0 .. wg_cam_n[w]), first-appearance order) and gives every observation its camera's slot in that list
    // (o_slot[i], 255 = not staged: the list is capped at kCamStageCap).  The workgroup copies those cameras to LDS once
    // -- a few hundred line accesses instead of ten scattered 16-byte loads per observation and lane.
    // Landmark sharding (round 6): a rank's landmarks are ONE internal range, and the landmark-major kernels (k_landmark_reduce,
    // k_back_substitute and its matrix-free form) are launched over the workgroups that cover it -- lm_wgn of them from workgroup
    // lm_wg0 on (0, 0: every landmark).  Until round 5 they ran over all n_pt landmarks on every rank (a record, g_l and a step
    // for 4.4 M landmarks without local observations: 0.39 of 0.09 + 0.30 ms at eight ranks).  The other ranks' entries of hinv,
    // g_l, dl stay zero.
    int lm_wg0 = 0, lm_wgn = 0;
    const uint8_t* o_slot;        // [n_obs]
    const uint8_t* wg_cam_n;      // [workgroups]
    const uint32_t* wg_cam_list;  // [workgroups][kCamStageCap]
};
constexpr int kLmWg = 128;          // landmarks per workgroup of the landmark-major kernels (k_landmark_reduce: 4 lanes per landmark, 512 threads;
                                    // k_back_substitute: 2 lanes, 256 threads) -- round 5; 64 until then
constexpr int kCamStageCap = 128;   // distinct cameras staged per workgroup (slot 255 = gather from memory)

// Per-landmark record written by k_landmark_reduce and read by the camera-major kernels: everything a
// camera-major gather needs from a landmark sits in ONE 128-byte line instead of three arrays.
// In MEMORY (round 4): Hll^-1 -- bitwise symmetric -- as (00, 01, 02, 11, 12, 22) | p_w.x, p_w.y || p_w.z | g_l (3) | pad: the first
// 64-byte line is all the pair kernel needs of a landmark (p_w.z rides in the projection records), the first 96 bytes all
// anybody needs.  In REGISTERS (load_lm_record, ba_kernels.hip) the kernels keep the layout of rounds 1-3:
constexpr int kLmStride = 16;  // doubles: Hll^-1 (9, row-major) | point (3) | g_l (3) | pad
constexpr int kLmPt = 9, kLmG = 12;
constexpr int kLmuStride = 8;  // matrix-free Schur operator: {point(3), -, u_l(3), -} per landmark, 64 bytes

// Lower-triangular tile map of the reduced camera matrix S.
struct TileMap {
    double* tiles;       // slot-major tile storage
    const int* slot;     // [nt*nt], slot[I*nt+J] for I >= J, -1 when the tile is structurally zero
    int nt;
};

void launch_cam_reduce(int dc, const BAView& v, const TileMap& tm, const int* cam_ptr, const int* cam_obs,
                       double lambda, int add_lambda, const double* hinv, const double* g_l, int with_self, double* g_c,
                       double* g_red, hipStream_t s);
// orec (may be NULL): [n_obs][4] projection records (xn, yn, p_w.z, sqrt(rho')), landmark-major, for the record form of
// the Schur pair kernel (schur_pairs.hip) and of the back-substitution
void launch_landmark_reduce(int dc, const BAView& v, double lambda, double* hinv, double* g_l, int* err_flag,
                            double* lmu /* may be NULL */, hipStream_t s, double* orec = nullptr);
// A18 (implicit_schur.rs): y = S x matrix-free, the Schur-Jacobi preconditioner blocks and their application
void launch_implicit_matvec(int dc, const BAView& v, const int* cam_ptr, const double* hinv, double* lmu, const double* x,
                            double lambda, double* y, hipStream_t s, const double* orec = nullptr);
void launch_clear3(double* a, double* b, int64_t n, int* f, int nf, hipStream_t s);   // a[0..n) = b[0..n) = 0, f[0..nf) = 0: one launch
void launch_gather_uv(int64_t n, const int* idx, const double* src, double* dst, hipStream_t s);          // dst[k] = src[idx[k]] (double2)
void launch_gather_u32(int64_t n, const int* idx, const uint32_t* src, uint32_t* dst, hipStream_t s);
void launch_extract_diag_blocks(int dc, int64_t n_cam, const TileMap& tm, double* sd, hipStream_t s);
void launch_precond_blocks(int dc, int64_t n_cam, const double* sd, double* minv, hipStream_t s);
void launch_precond_apply(int dc, int64_t n_cam, const double* minv, const double* r, double* z, hipStream_t s);
// mask_code = 4 POSE + 2 LANDMARK + INTRINSIC: which blocks of the factors' Jacobians exist (OptimizeParams, src/factors/mod.rs:66-101)
void launch_prepare_cams(int64_t n_cam, const double* poses, const double* intr, double* camp, int mask_code, hipStream_t s);
// orec: the projection records of the same linearisation (k_landmark_reduce) or NULL -- the record form of the kernel
void launch_back_substitute(int dc, const BAView& v, const double* hinv, const double* g_l, const double* dcam,
                            double* dl, hipStream_t s, const double* orec = nullptr, const uint8_t* fix_pt = nullptr,
                            double* pts_trial = nullptr /* with fix_pt: the trial points p (+) dl are written too */);
void launch_retract(int dc, int64_t n_cam, int64_t n_pt, const double* poses, const double* intr, const double* pts,
                    const double* dcam, const double* dl, double sign, const uint8_t* fix_pose,
                    const uint8_t* fix_intr, const uint8_t* fix_pt, double* poses_out, double* intr_out,
                    double* pts_out, hipStream_t s);
void launch_cost(const BAView& v, double* partial, int n_partial, double* out_sumsq, hipStream_t s);
// gscale (may be NULL): the gradient the statistics use is g .* gscale (the scaled gradient of Jacobi scaling)
void launch_step_stats(int64_t n, const double* g, const double* d, double lambda, const double* gscale, double* partial,
                       int n_partial, double* out3, hipStream_t s);
// Jacobi scaling: n2_cam[c][d_c], n2_pt[l][3] += squared entries of the corrected Jacobian's columns (atomics; a
// one-off per optimize), and s = 1 / (1 + sqrt(n2)) (process_jacobian_generic, optimizer/mod.rs:754-757)
void launch_column_norms_sq(int dc, const BAView& v, double* n2_cam, double* n2_pt, hipStream_t s);
void launch_scaling_from_norms_sq(int64_t n, const double* n2, double* scale, hipStream_t s);
void launch_vec_mul(int64_t n, const double* a, const double* b, double* out, hipStream_t s);  // out = a .* b (in place ok)
void launch_vec_add(int64_t n, const double* a, const double* b, double* out, hipStream_t s);  // out = a + b (in place ok)
// sd[c][a][b] *= s[c][a] s[c][b]: the Schur-Jacobi diagonal blocks in the scaled variables
void launch_scale_diag_blocks(int dc, int64_t n_cam, const double* scale, double* sd, hipStream_t s);
void launch_sumsq(int64_t n, const double* x, double* partial, int n_partial, double* out, hipStream_t s);
// out[0] = a1.b1, out[1] = a2.b2 in one pass over the vectors (fixed two-level reduction: reproducible)
void launch_dot2(int64_t n, const double* a1, const double* b1, const double* a2, const double* b2, double* partial,
                 int n_partial, double* out, hipStream_t s);
// out[k] = sum_i partial[i*nk + k] in index order (one block: reproducible)
void launch_sum_partials(const double* partial, int n, int nk, double* out, hipStream_t s);
void launch_debug_invert_blocks(int64_t n, const double* in, double* out, int* ok, hipStream_t s);
void launch_export_linearization(int dc, const BAView& v, const int* o_orig, double* r_out, double* jc_out,
                                 double* jl_out, hipStream_t s);

}  // namespace apex
