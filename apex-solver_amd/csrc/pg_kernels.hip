// pg_kernels.hip -- SE3 pose-graph kernels: BetweenFactor linearisation fused with the block-sparse
// J^T J / J^T r assembly, trial cost, retraction.
//
//   k_pg_prepare   vertex-major  normalised poses of a parameter set (SE3::from(DVector))
//   k_pg_edges     edge-major    r, dr/dk0, dr/dk1 per edge in registers (never written to memory),
//                                loss correction, then H_aa += Ja^T Ja, H_bb += Jb^T Jb,
//                                H_(hi,lo) += J_hi^T J_lo, g_a += Ja^T r, g_b += Jb^T r with fp64 atomics
//                                (SparseCholeskySolver's J^T J and J^T r, src/linalg/sparse/cholesky.rs:166-181)
//   k_pg_cost      edge-major    1/2 |r~|^2 at a (trial) parameter set
//   k_pg_retract   vertex-major  x (+) d with the fixed-DOF mask (src/core/problem.rs:185-197)
//
// HBM-bound and tiny next to the factorisation: per edge 2 x 64 B poses + 64 B measurement in,
// 3 x 288 B + 2 x 48 B of atomics out.
#include <hip/hip_runtime.h>

#include "pg_device.hpp"
#include "pg_kernels.h"

namespace apex {

__device__ __forceinline__ double pg_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double pg_block_sum_256(double v, double* scratch) {
    v = pg_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_pg_prepare(int64_t n, const double* __restrict__ poses7, double* __restrict__ posep) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    double p[7], o[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) p[a] = poses7[7 * v + a];
    pose_normalise(p, o);
#pragma unroll
    for (int a = 0; a < 7; ++a) posep[kPoseStride * v + a] = o[a];
    posep[kPoseStride * v + 7] = 0.0;
}

__device__ __forceinline__ void load_pose8(const double* __restrict__ base, int64_t i, double p[7]) {
    const double2* q = reinterpret_cast<const double2*>(base + kPoseStride * i);
    const double2 a = q[0], b = q[1], c = q[2], d = q[3];
    p[0] = a.x; p[1] = a.y; p[2] = b.x; p[3] = b.y; p[4] = c.x; p[5] = c.y; p[6] = d.x;
}

// 6x6 block (row vertex vr, column vertex vc, vr >= vc) of the lower-triangular tile matrix
__device__ __forceinline__ double* h_block_ptr(const TileMap& tm, uint32_t vr, uint32_t vc) {
    const uint32_t I = vr / kVertsPerTile, J = vc / kVertsPerTile;
    const int slot = tm.slot[(size_t)I * tm.nt + J];
    return tm.tiles + (size_t)slot * (kNB * kNB) + (size_t)((vr % kVertsPerTile) * 6) * kNB + (vc % kVertsPerTile) * 6;
}

__global__ __launch_bounds__(256) void k_pg_edges(PGView v, TileMap tm, double* __restrict__ g) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n_e) return;
    const uint32_t a = v.e_from[e], b = v.e_to[e];
    double k0[7], k1[7], m[7], r[6];
    load_pose8(v.posep, a, k0);
    load_pose8(v.posep, b, k1);
    load_pose8(v.meas, e, m);
    Jac6 J0, J1;
    between_linearize(k0, k1, m, r, J0, J1);
    // loss correction: r and J scale by sqrt(rho') (corrector.rs:143-181; rho'' <= 0 for Huber)
    const double sc = pg_huber_scale(v.huber_delta, r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3] + r[4] * r[4] + r[5] * r[5]);
    if (sc != 1.0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] *= sc;
#pragma unroll
        for (int i = 0; i < 9; ++i) { J0.P[i] *= sc; J0.T[i] *= sc; J1.P[i] *= sc; J1.T[i] *= sc; }
    }
    double H[36], gv[6];
    jtj(J0, J0, H);
    {
        double* blk = h_block_ptr(tm, a, a);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) unsafeAtomicAdd(blk + i * kNB + j, H[6 * i + j]);
    }
    jtj(J1, J1, H);
    {
        double* blk = h_block_ptr(tm, b, b);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) unsafeAtomicAdd(blk + i * kNB + j, H[6 * i + j]);
    }
    if (a != b) {
        double* blk;
        if (a > b) { jtj(J0, J1, H); blk = h_block_ptr(tm, a, b); }
        else       { jtj(J1, J0, H); blk = h_block_ptr(tm, b, a); }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) unsafeAtomicAdd(blk + i * kNB + j, H[6 * i + j]);
    } else {  // self-loop: both Jacobians hit the same columns, the cross terms land on the diagonal block
        jtj(J0, J1, H);
        double* blk = h_block_ptr(tm, a, a);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) unsafeAtomicAdd(blk + i * kNB + j, H[6 * i + j] + H[6 * j + i]);
    }
    jtr(J0, r, gv);
#pragma unroll
    for (int i = 0; i < 6; ++i) unsafeAtomicAdd(g + (size_t)a * 6 + i, gv[i]);
    jtr(J1, r, gv);
#pragma unroll
    for (int i = 0; i < 6; ++i) unsafeAtomicAdd(g + (size_t)b * 6 + i, gv[i]);
}

// One prior block: the corrected residual (7 rows) and sqrt(rho') -- the variable as the factor sees it is the prepared
// pose (SE3::from(DVector).to_vector(), unit quaternion).
__device__ __forceinline__ double prior_eval(const PGView& v, int k, double r[7]) {
    double x[7], d[7];
    load_pose8(v.posep, v.prior_v[k], x);
    load_pose8(v.prior_data, k, d);
    const double delta = v.prior_data[kPoseStride * (size_t)k + 7];
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 7; ++a) { r[a] = x[a] - d[a]; s += r[a] * r[a]; }
    const double sc = pg_huber_scale(delta, s);
#pragma unroll
    for (int a = 0; a < 7; ++a) r[a] *= sc;
    return sc;
}
__global__ __launch_bounds__(64) void k_pg_priors(PGView v, TileMap tm, double* __restrict__ g) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= v.n_prior) return;
    double r[7];
    const double sc = prior_eval(v, k, r);
    const uint32_t a = v.prior_v[k];
    double* blk = h_block_ptr(tm, a, a);
#pragma unroll
    for (int i = 0; i < 6; ++i) {   // J~ = sc [I6; 0]: J~^T J~ = sc^2 I6, J~^T r~ = sc r~[0..5]
        unsafeAtomicAdd(blk + i * kNB + i, sc * sc);
        unsafeAtomicAdd(g + (size_t)a * 6 + i, sc * r[i]);
    }
}
__global__ __launch_bounds__(64) void k_pg_prior_export(PGView v, double* __restrict__ r7_out) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= v.n_prior) return;
    double r[7];
    (void)prior_eval(v, k, r);
    for (int a = 0; a < 7; ++a) r7_out[7 * k + a] = r[a];
}

__global__ __launch_bounds__(256) void k_pg_cost_partial(PGView v, double* __restrict__ partial) {
    __shared__ double scratch[4];
    double acc = 0.0;
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < v.n_prior; k += 256) {
            double r[7];
            (void)prior_eval(v, k, r);
#pragma unroll
            for (int a = 0; a < 7; ++a) acc += r[a] * r[a];
        }
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < v.n_e; e += (int64_t)gridDim.x * 256) {
        double k0[7], k1[7], m[7], r[6], tA[3], qA[4], D[9];
        load_pose8(v.posep, v.e_from[e], k0);
        load_pose8(v.posep, v.e_to[e], k1);
        load_pose8(v.meas, e, m);
        between_residual(k0, k1, m, r, tA, qA, D);
        const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3] + r[4] * r[4] + r[5] * r[5];
        const double sc = pg_huber_scale(v.huber_delta, s);
        acc += (sc * sc) * s;
    }
    acc = pg_block_sum_256(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_pg_retract(int64_t n_v, const double* __restrict__ poses,
                                                      const double* __restrict__ d, double sign,
                                                      const uint8_t* __restrict__ fix, double* __restrict__ poses_out) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_v) return;
    double dd[6], p[7], o[7];
#pragma unroll
    for (int a = 0; a < 6; ++a) dd[a] = fix[6 * v + a] ? 0.0 : sign * d[6 * v + a];
#pragma unroll
    for (int a = 0; a < 7; ++a) p[a] = poses[7 * v + a];
    se3_plus(p, dd, o);
#pragma unroll
    for (int a = 0; a < 7; ++a) poses_out[7 * v + a] = o[a];
}

__global__ __launch_bounds__(256) void k_pg_negate(int64_t n, const double* __restrict__ x, double* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = -x[i];
}

__global__ __launch_bounds__(256) void k_pg_export(PGView v, double* __restrict__ r_out, double* __restrict__ j_out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= v.n_e) return;
    double k0[7], k1[7], m[7], r[6];
    load_pose8(v.posep, v.e_from[e], k0);
    load_pose8(v.posep, v.e_to[e], k1);
    load_pose8(v.meas, e, m);
    Jac6 J[2];
    between_linearize(k0, k1, m, r, J[0], J[1]);
    const double sc = pg_huber_scale(v.huber_delta, r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3] + r[4] * r[4] + r[5] * r[5]);
    if (r_out)
        for (int i = 0; i < 6; ++i) r_out[6 * e + i] = sc * r[i];
    if (j_out)
        for (int w = 0; w < 2; ++w)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double* o = j_out + 72 * e + 6 * w;
                    o[12 * i + j] = sc * J[w].P[3 * i + j];
                    o[12 * i + 3 + j] = sc * J[w].T[3 * i + j];
                    o[12 * (i + 3) + j] = 0.0;
                    o[12 * (i + 3) + 3 + j] = sc * J[w].P[3 * i + j];
                }
}

static inline int grid256(int64_t n) { return (int)((n + 255) / 256); }

void launch_pg_prepare(int64_t n, const double* poses7, double* posep, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_pg_prepare, dim3(grid256(n)), dim3(256), 0, s, n, poses7, posep);
}
void launch_pg_edges(const PGView& v, const TileMap& tm, double* g, hipStream_t s) {
    if (v.n_e > 0) hipLaunchKernelGGL(k_pg_edges, dim3(grid256(v.n_e)), dim3(256), 0, s, v, tm, g);
}
void launch_pg_priors(const PGView& v, const TileMap& tm, double* g, hipStream_t s) {
    if (v.n_prior > 0) hipLaunchKernelGGL(k_pg_priors, dim3((v.n_prior + 63) / 64), dim3(64), 0, s, v, tm, g);
}
void launch_pg_prior_export(const PGView& v, double* r7_out, hipStream_t s) {
    if (v.n_prior > 0) hipLaunchKernelGGL(k_pg_prior_export, dim3((v.n_prior + 63) / 64), dim3(64), 0, s, v, r7_out);
}
void launch_pg_cost(const PGView& v, double* partial, int n_partial, double* out_sumsq, hipStream_t s) {
    hipLaunchKernelGGL(k_pg_cost_partial, dim3(n_partial), dim3(256), 0, s, v, partial);
    launch_sum_partials(partial, n_partial, 1, out_sumsq, s);
}
void launch_pg_retract(int64_t n_v, const double* poses, const double* d, double sign, const uint8_t* fix,
                       double* poses_out, hipStream_t s) {
    if (n_v > 0) hipLaunchKernelGGL(k_pg_retract, dim3(grid256(n_v)), dim3(256), 0, s, n_v, poses, d, sign, fix, poses_out);
}
void launch_pg_negate(int64_t n, const double* x, double* y, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_pg_negate, dim3(grid256(n)), dim3(256), 0, s, n, x, y);
}
void launch_pg_export(const PGView& v, double* r_out, double* j_out, hipStream_t s) {
    if (v.n_e > 0) hipLaunchKernelGGL(k_pg_export, dim3(grid256(v.n_e)), dim3(256), 0, s, v, r_out, j_out);
}

}  // namespace apex
