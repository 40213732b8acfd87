// ba_kernels.hip -- CDNA4 (gfx950) kernels of the bundle-adjustment hot path.
//
// Data layout in HBM:
//   observations   landmark-major (sorted by landmark): o_cam[n_obs] u32, o_pt[n_obs] u32,
//                  o_uv[n_obs] double2 (one 16-B load per lane), pt_ptr[n_pt+1] i32
//   cameras        poses[n_cam][7], intr[n_cam][3]   (<= 1.4 MB even for final-13682: L2-resident)
//   landmarks      pts[n_pt][3]
//   camera-major   cam_ptr[n_cam+1], cam_obs[n_obs] (indices into the sorted observation arrays)
//   S              lower-triangular TILES of NB x NB = 144 x 144 doubles, tile (I,J) at
//                  tiles + slot[I*nt+J]*NB*NB, row-major inside a tile.  144 = 16*9 = 24*6, so a
//                  camera's 9 (or 6) rows never straddle a tile and 144 is a multiple of the
//                  16-wide f64 MFMA.
//
// Kernels (one per stage; the per-observation math is recomputed from the 24-byte observation
// record instead of streaming a 216-byte Jacobian row through HBM):
//   k_cam_reduce       camera-major   H_cc diagonal blocks (+lambda), g_c, g_red := -g_c     (A6-A8)
//   k_landmark_reduce  landmark-major H_ll, g_l, eigen-gated 3x3 inverse                     (A6, A8, A9)
//   k_back_substitute  landmark-major dl = Hll^-1 (-g_l - W^T dc)                            (A11)
//   k_retract_*        x (+) d with the fixed-DOF mask                                       (A15)
//   k_cost_partial     1/2 |r~|^2 on a (trial) parameter set                                 (A16)
//   k_step_stats       |g|^2, |d|^2, d.(lambda d - g)                                        (A14)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "ba_device.hpp"
#include "ba_kernels.h"

namespace apex {

// ------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

// sum over the 256 threads of a block; result valid in thread 0.  `scratch` holds 4 doubles.
__device__ __forceinline__ double block_sum_256(double v, double* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    __syncthreads();
    return r;
}

// (memory layout -> register layout, see ba_kernels.h: six 16-byte loads, the expansion is register moves)
__device__ __forceinline__ void load_lm_record(const double* __restrict__ rec, size_t l, double out[kLmStride]) {
    const double2* q = reinterpret_cast<const double2*>(rec + kLmStride * l);
    const double2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4], a5 = q[5];
    out[0] = a0.x; out[1] = a0.y; out[2] = a1.x; out[3] = a0.y; out[4] = a1.y; out[5] = a2.x; out[6] = a1.x; out[7] = a2.x; out[8] = a2.y;
    out[kLmPt] = a3.x; out[kLmPt + 1] = a3.y; out[kLmPt + 2] = a4.x;
    out[kLmG] = a4.y; out[kLmG + 1] = a5.x; out[kLmG + 2] = a5.y; out[15] = 0.0;
}

template <int DC>
__device__ __forceinline__ double* s_block_ptr(const TileMap& tm, uint32_t row_cam, uint32_t col_cam) {
    constexpr int CPT = kNB / DC;  // cameras per tile
    const uint32_t I = row_cam / CPT, J = col_cam / CPT;
    const int slot = tm.slot[(size_t)I * tm.nt + J];
    return tm.tiles + (size_t)slot * (kNB * kNB) + (size_t)((row_cam % CPT) * DC) * kNB + (col_cam % CPT) * DC;
}

// ------------------------------------------------------------------------------------------
// K1: camera-major reduction.  One 256-thread workgroup per camera; every lane walks the
// camera's observation list with stride 256, keeps the upper triangle of Jc^T Jc and Jc^T r in
// registers, then a fixed-order wave/LDS reduction (bitwise reproducible).
// Writes (plain stores, no atomics) the lower triangle of the camera's diagonal block of S
// (+lambda on the diagonal when add_lambda), g_c and g_red := -g_c.
// ------------------------------------------------------------------------------------------
constexpr int kCamThreads = 64;  // one wave per camera: ~30 observations per lane amortise the 63-value reduction
template <int DC, bool MASKED, bool WITH_SELF>
__global__ __launch_bounds__(kCamThreads, 2) void k_cam_reduce(BAView v, TileMap tm, const int* __restrict__ cam_ptr,
                                                      const int* __restrict__ cam_obs, double lambda,
                                                      int add_lambda, const double* __restrict__ hinv,
                                                      const double* __restrict__ g_l,
                                                      double* __restrict__ g_c, double* __restrict__ g_red) {
    // with_self (row form of the Schur reduction): the block also receives the camera's own Schur
    // terms -sum_i Y_i W_i^T and g_red is completed with +sum_i Y_i g_l, so that k_schur_rows only
    // has pairs of DIFFERENT observations left (no same-address LDS atomics).
    constexpr int NH = DC * (DC + 1) / 2;
    const uint32_t c = blockIdx.x;
    constexpr int NW = kCamThreads / 64;
    __shared__ double red[NW][NH + 2 * DC];
    double acc[NH + 2 * DC];
#pragma unroll
    for (int i = 0; i < NH + 2 * DC; ++i) acc[i] = 0.0;
    Cam cam;
    load_cam_prepared(v.camp + kCamStride * (size_t)c, cam);
    const int b = cam_ptr[c], e = cam_ptr[c + 1];
    // the next observation's landmark index and measurement are in flight while this one is reduced: the record gather of an
    // iteration starts at once instead of after a round trip for its address (affordable since the self term goes through the
    // 2 x 2 matrix P: at 256 VGPRs the three extra live registers spilled, 1.14 -> 1.29 ms)
    const int k_first = max(min(b + (int)threadIdx.x, e - 1), 0);
    uint32_t l_next = 0;
    double2 uv_next = make_double2(0.0, 0.0);
    double rec_next[kLmStride];
    if (b < e) {   // (wave-uniform.  A camera without observations on this rank -- seven in eight on a tree-sharded rank of eight --
                   // only writes its zeros / lambda: no dependent index -> record round trips)
        l_next = v.co_pt[k_first];
        uv_next = v.co_uv[k_first];
        load_lm_record(hinv, l_next, rec_next);
        if (k_first + kCamThreads < e) l_next = v.co_pt[k_first + kCamThreads];
    }
    for (int k = b + (int)threadIdx.x; k < e; k += kCamThreads) {
        const double2 uv = uv_next;
        double rec[kLmStride];
#pragma unroll
        for (int q = 0; q < kLmStride; ++q) rec[q] = rec_next[q];
        if (k + kCamThreads < e) {   // the next observation's record and measurement, the index of the one after
            uv_next = v.co_uv[k + kCamThreads];
            load_lm_record(hinv, l_next, rec_next);
            if (k + 2 * kCamThreads < e) l_next = v.co_pt[k + 2 * kCamThreads];
        }
        const double pw[3] = {rec[kLmPt], rec[kLmPt + 1], rec[kLmPt + 2]};
        double r[2], Jc[2][DC], Jl[2][3];
        linearize_obs<DC, MASKED>(cam, pw, uv.x, uv.y, v.huber_delta, r, Jc, Jl);
#pragma unroll
        for (int a = 0; a < DC; ++a) acc[NH + a] += Jc[0][a] * r[0] + Jc[1][a] * r[1];
        if (WITH_SELF) {
            // The camera's own Schur term folded into the block through the observation's 2 x 2 matrix
            //     P = I - Jl Hll^-1 Jl^T :   Jc^T Jc - (Jc^T Jl) Hll^-1 (Jl^T Jc) = Jc^T P Jc,
            // and W Hll^-1 g_l = Jc^T (Jl (Hll^-1 g_l)): 2-vectors and a 2 x 2 instead of the 9 x 3 matrices W and Y = W Hll^-1 --
            // ~200 instead of ~400 multiply-adds per observation, 54 registers fewer, and no difference of two large sums.
            const double* Hi = rec;
            const double gl0 = rec[kLmG], gl1 = rec[kLmG + 1], gl2 = rec[kLmG + 2];
            double T[2][3];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int q = 0; q < 3; ++q) T[rr][q] = Jl[rr][0] * Hi[q] + Jl[rr][1] * Hi[3 + q] + Jl[rr][2] * Hi[6 + q];
            const double p00 = 1.0 - (T[0][0] * Jl[0][0] + T[0][1] * Jl[0][1] + T[0][2] * Jl[0][2]);
            const double p01 = -(T[0][0] * Jl[1][0] + T[0][1] * Jl[1][1] + T[0][2] * Jl[1][2]);
            const double p11 = 1.0 - (T[1][0] * Jl[1][0] + T[1][1] * Jl[1][1] + T[1][2] * Jl[1][2]);
            const double v0 = Hi[0] * gl0 + Hi[1] * gl1 + Hi[2] * gl2, v1 = Hi[3] * gl0 + Hi[4] * gl1 + Hi[5] * gl2,
                         v2 = Hi[6] * gl0 + Hi[7] * gl1 + Hi[8] * gl2;
            const double w0 = Jl[0][0] * v0 + Jl[0][1] * v1 + Jl[0][2] * v2, w1 = Jl[1][0] * v0 + Jl[1][1] * v1 + Jl[1][2] * v2;
            int idx = 0;
#pragma unroll
            for (int a = 0; a < DC; ++a) {
                const double q0 = p00 * Jc[0][a] + p01 * Jc[1][a], q1 = p01 * Jc[0][a] + p11 * Jc[1][a];
                acc[NH + DC + a] += Jc[0][a] * w0 + Jc[1][a] * w1;
#pragma unroll
                for (int bb = 0; bb <= a; ++bb) acc[idx++] += q0 * Jc[0][bb] + q1 * Jc[1][bb];
            }
        } else {
            int idx = 0;
#pragma unroll
            for (int a = 0; a < DC; ++a)
#pragma unroll
                for (int bb = 0; bb <= a; ++bb) acc[idx++] += Jc[0][a] * Jc[0][bb] + Jc[1][a] * Jc[1][bb];
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NH + 2 * DC; ++i) {
        double s = wave_sum(acc[i]);
        if (lane == 0) red[w][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NH + DC) {
        const int i = threadIdx.x;
        double s = red[0][i];
#pragma unroll
        for (int q = 1; q < NW; ++q) s += red[q][i];
        if (i < NH) {
            // packed lower index -> (a,b), a >= b
            int a = 0;
            while ((a + 1) * (a + 2) / 2 <= i) ++a;
            const int bb = i - a * (a + 1) / 2;
            double* blk = s_block_ptr<DC>(tm, c, c);
            double lam = lambda;
            if (v.cam_scale) { const double sc = v.cam_scale[(size_t)c * DC + a]; lam = lambda / (sc * sc); }
            const bool addl = v.lam_mask ? v.lam_mask[c] != 0 : add_lambda != 0;
            blk[a * kNB + bb] = s + ((a == bb && addl) ? lam : 0.0);
        } else {
            const int a = i - NH;
            const int j = NH + DC + a;
            double yg = red[0][j];
#pragma unroll
            for (int q = 1; q < NW; ++q) yg += red[q][j];
            g_c[(size_t)c * DC + a] = s;
            g_red[(size_t)c * DC + a] = -s + yg;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2a: landmark-major reduction, 8 lanes per landmark (32 landmarks per 256-thread block).
// H_ll = sum Jl^T Jl + lambda I, g_l = sum Jl^T r, eigen-gated inverse.
// ------------------------------------------------------------------------------------------
// Camera staging (BAView::o_slot): the workgroup's distinct cameras (+ EXTRA doubles each, e.g. the camera step) copied to
// LDS by all 256 threads, STR doubles per slot (even: 16-byte pieces) -- in three steps, so that a kernel can put its own
// loads between them: the landmark-major kernels are
// bound by the latency of their dependent loads (workgroup list -> cameras -> barrier, then pt_ptr -> observations), and
// interleaved the two chains cost two round trips instead of four.  Every load is unconditional on a clamped index
// (straight-line code: nothing keeps the compiler from issuing the caller's loads in between); only the LDS stores are
// predicated.  Item q of camera k: q < 5 the q-th 16-byte piece of the compact camera, else extra[cam][q - 5].
// CAMD: doubles per camera in the source array (kCamQStride: the compact form, kCamStride: the prepared records).
template <int STR, int EXTRA, int CAMD = kCamQStride, int NT = 256>   // NT: threads of the workgroup (all of them copy)
struct CamStager {
    static constexpr int PQ = CAMD / 2, IT = PQ + EXTRA, NJ = (kCamStageCap * IT + NT - 1) / NT;
    static_assert(STR % 2 == 0 && CAMD % 2 == 0 && STR >= CAMD + EXTRA, "slot pitch");
    uint32_t ci[NJ];
    double2 d[NJ];
    int n;
    __device__ __forceinline__ void issue_indices(const BAView& v) {
        // (the list is zero-padded to kCamStageCap entries: no need to wait for n before reading it)
        const size_t wg = (size_t)blockIdx.x + (size_t)v.lm_wg0;   // (the landmark-major kernels run over the rank's own workgroups: BAView::lm_wg0)
        n = v.o_slot ? v.wg_cam_n[wg] : 0;
        const uint32_t* __restrict__ list = v.wg_cam_list + wg * kCamStageCap;
#pragma unroll
        for (int j = 0; j < NJ; ++j) ci[j] = list[min(((int)threadIdx.x + NT * j) / IT, kCamStageCap - 1)];
    }
    __device__ __forceinline__ void issue_data(const double* __restrict__ cams, const double* __restrict__ extra) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int idx = (int)threadIdx.x + NT * j, q = idx % IT;
            if (EXTRA == 0 || q < PQ) d[j] = reinterpret_cast<const double2*>(cams + CAMD * (size_t)ci[j])[q];
            else d[j] = make_double2(extra[(size_t)ci[j] * EXTRA + (q - PQ)], 0.0);
        }
    }
    __device__ __forceinline__ void store(double* __restrict__ sCam) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int idx = (int)threadIdx.x + NT * j, k = idx / IT, q = idx - IT * k;
            if (k < n) {
                if (EXTRA == 0 || q < PQ) reinterpret_cast<double2*>(sCam + k * STR)[q] = d[j];
                else sCam[k * STR + CAMD + (q - PQ)] = d[j].x;
            }
        }
        __syncthreads();
    }
};

constexpr int kBsLanes = 2;   // lanes per landmark of k_back_substitute (see there)
constexpr int kLmLanes = 4;   // lanes per landmark in the landmark-major kernels.  final-13682 (3..9 observations per landmark): 8 lanes
                              // 1.05 + 0.97 ms (k_landmark_reduce + k_back_substitute), 4 lanes 0.87 + 0.77, 2 lanes 0.74 + 0.68, 1 lane
                              // 0.75 + 0.73; 4 keeps the tail of a landmark with a thousand observations at a few hundred microseconds
// (occupancy: 114 VGPRs = 4 waves per SIMD.  Forcing 5 / 6 with amdgpu_waves_per_eu spills 124 / 180 bytes per lane and the
// kernel goes from 0.78 to 1.94 / 3.10 ms -- k_back_substitute likewise 0.48 -> 0.85 / 2.12: profiles/r05_ab_waves_per_eu.txt)
template <int DC>
__global__ __launch_bounds__(kLmWg * kLmLanes) void k_landmark_reduce(BAView v, double lambda, double* __restrict__ hinv,
                                                           double* __restrict__ g_l, int* __restrict__ err_flag,
                                                           double* __restrict__ lmu, double* __restrict__ orec) {
    static_assert(kLmWg * kLmLanes <= 1024 && kLmWg * kBsLanes <= 1024, "a workgroup is kLmWg landmarks (the camera staging lists are built per kLmWg landmarks)");
    __shared__ double sCam[kCamStageCap * kCamQStride];
    const int g = threadIdx.x & (kLmLanes - 1);
    const int64_t l = ((int64_t)blockIdx.x + v.lm_wg0) * kLmWg + threadIdx.x / kLmLanes;
    const bool active = l < v.n_pt;
    double h[6] = {0, 0, 0, 0, 0, 0}, gl[3] = {0, 0, 0}, pw[3] = {0, 0, 0};
    // two round trips to memory before the arithmetic starts: (list, pt_ptr), then (cameras, point, first observation)
    CamStager<kCamQStride, 0, kCamQStride, kLmWg * kLmLanes> stager;
    stager.issue_indices(v);
    const int64_t lc = active ? l : 0;
    const int b = v.pt_ptr[lc], e = active ? v.pt_ptr[lc + 1] : b;
    stager.issue_data(v.camq, nullptr);
    pw[0] = v.pts[3 * lc]; pw[1] = v.pts[3 * lc + 1]; pw[2] = v.pts[3 * lc + 2];
    const int i_first = max(min(b + g, (int)v.n_obs - 1), 0);   // (clamped: the load is unconditional, the use is not)
    double2 uv_next = v.o_uv[i_first];
    int sl_next = v.o_slot ? (int)v.o_slot[i_first] : 255;
    stager.store(sCam);
    if (active) {
        for (int i = b + g; i < e; i += kLmLanes) {
            const double2 uv = uv_next;
            const int sl = sl_next;
            if (i + kLmLanes < e) {   // the next observation's record is in flight while this one is linearised
                uv_next = v.o_uv[i + kLmLanes];
                sl_next = v.o_slot ? (int)v.o_slot[i + kLmLanes] : 255;
            }
            Cam cam;
            if (sl != 255) load_cam_q(sCam + sl * kCamQStride, v.mask_code, cam);
            else load_cam_q(v.camq + kCamQStride * (size_t)v.o_cam[i], v.mask_code, cam);
            double r[2], Jc[2][DC], Jl[2][3];
            if (orec) {   // record form of the pair kernel: the observation's projection record, 32 bytes, landmark-major
                double rec[4];
                linearize_obs<DC>(cam, pw, uv.x, uv.y, v.huber_delta, r, Jc, Jl, rec);
                double2* q = reinterpret_cast<double2*>(orec + 4 * (size_t)i);
                q[0] = make_double2(rec[0], rec[1]); q[1] = make_double2(rec[2], rec[3]);
            } else {
                linearize_obs<DC>(cam, pw, uv.x, uv.y, v.huber_delta, r, Jc, Jl);
            }
            h[0] += Jl[0][0] * Jl[0][0] + Jl[1][0] * Jl[1][0];
            h[1] += Jl[0][1] * Jl[0][0] + Jl[1][1] * Jl[1][0];
            h[2] += Jl[0][1] * Jl[0][1] + Jl[1][1] * Jl[1][1];
            h[3] += Jl[0][2] * Jl[0][0] + Jl[1][2] * Jl[1][0];
            h[4] += Jl[0][2] * Jl[0][1] + Jl[1][2] * Jl[1][1];
            h[5] += Jl[0][2] * Jl[0][2] + Jl[1][2] * Jl[1][2];
#pragma unroll
            for (int a = 0; a < 3; ++a) gl[a] += Jl[0][a] * r[0] + Jl[1][a] * r[1];
        }
    }
#pragma unroll
    for (int m = 1; m < kLmLanes; m <<= 1) {
#pragma unroll
        for (int i = 0; i < 6; ++i) h[i] += __shfl_xor(h[i], m, kLmLanes);
#pragma unroll
        for (int i = 0; i < 3; ++i) gl[i] += __shfl_xor(gl[i], m, kLmLanes);
    }
    if (active && g == 0) {
        // With Jacobi scaling the reference inverts the block of the SCALED system, D Hll D + lambda I (eigenvalue
        // gate included); D (.)^-1 D is then the inverse of Hll + lambda D^-2 that the unscaled kernels need.
        double sc[3] = {1.0, 1.0, 1.0};
        if (v.pt_scale) { sc[0] = v.pt_scale[3 * l]; sc[1] = v.pt_scale[3 * l + 1]; sc[2] = v.pt_scale[3 * l + 2]; }
        double B[9] = {sc[0] * sc[0] * h[0] + lambda, sc[1] * sc[0] * h[1], sc[2] * sc[0] * h[3],
                       sc[1] * sc[0] * h[1], sc[1] * sc[1] * h[2] + lambda, sc[2] * sc[1] * h[4],
                       sc[2] * sc[0] * h[3], sc[2] * sc[1] * h[4], sc[2] * sc[2] * h[5] + lambda};
        double Bi[9];
        if (!invert_landmark_block(B, Bi)) {
            atomicExch(err_flag, 1);
#pragma unroll
            for (int i = 0; i < 9; ++i) Bi[i] = 0.0;
        }
        if (v.pt_scale) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) Bi[3 * a + b] *= sc[a] * sc[b];
        }
        {   // the record's memory layout (ba_kernels.h); mat3_try_inverse of a symmetric block is symmetric bit for bit
            double2* q = reinterpret_cast<double2*>(hinv + kLmStride * l);
            q[0] = make_double2(Bi[0], Bi[1]); q[1] = make_double2(Bi[2], Bi[4]); q[2] = make_double2(Bi[5], Bi[8]);
            q[3] = make_double2(pw[0], pw[1]); q[4] = make_double2(pw[2], gl[0]); q[5] = make_double2(gl[1], gl[2]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) g_l[3 * l + i] = gl[i];
        if (lmu) {  // matrix-free variant: the point travels with u_l in a 64-byte record
#pragma unroll
            for (int i = 0; i < 3; ++i) lmu[kLmuStride * l + i] = pw[i];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Prepared cameras: normalise the stored quaternion twice (SE3::from, se3.rs:200-206, 107-113), turn it
// into R once, and park [R t f k1 k2] in 16 doubles per camera for every per-observation kernel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prepare_cams(int64_t n_cam, const double* __restrict__ poses,
                                                        const double* __restrict__ intr, double* __restrict__ camp, int mask_code) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cam) return;
    Cam cam;
    double qn[4];
    load_cam(poses + 7 * c, intr + 3 * c, cam, qn);
    cam.m_pose = (mask_code & 4) ? 1.0 : 0.0; cam.m_lm = (mask_code & 2) ? 1.0 : 0.0; cam.m_intr = (mask_code & 1) ? 1.0 : 0.0;
    store_cam_prepared(cam, camp + kCamStride * c);
    store_cam_q(qn, cam, camp + kCamStride * n_cam + kCamQStride * c);   // the compact copy follows the 16-double records
}

// ------------------------------------------------------------------------------------------
// K3: back-substitution, 8 lanes per landmark: dl = Hll^-1 ((-g)_l - H_cl^T dc)
// (explicit_schur.rs:980-1029); H_cl^T dc = sum_i Jl_i^T (Jc_i dc_ci).
// ------------------------------------------------------------------------------------------
// MATVEC = true is the landmark half of the matrix-free Schur operator (A18, implicit_schur.rs:186-226):
// u_l = Hll^-1 (H_cl^T x), written next to the point into the 64-byte record lmu[l] = {pt, -, u, -}.
// (4 waves per SIMD: 129 -> 128 VGPRs buys the fourth workgroup per CU, 0.70 -> 0.65 ms; the kernel is bound by the
// latency of its dependent loads, not by bytes or arithmetic)
// REC = true: J comes from the observation's projection record (written by k_landmark_reduce of the same linearisation,
// orec, landmark-major like the observations) and the prepared camera instead of a second linearisation:
//     Jc_i dc = a (dt + dtheta x p_w) + (xn w, yn w)^T (t . dk),   H_cl^T dc += a^T (Jc_i dc)
// ~110 instead of ~500 fp64 instructions per observation (the kernel was bound by exactly those once its loads were
// interleaved); only for the modes that optimise every column group the factor has (no masks in the record form).
// LANES per landmark (round 5): 2 here against k_landmark_reduce's 4 -- 0.39 against 0.48 ms on final-13682 (the A/B with every
// landmark-major kernel at 2 lanes: this kernel -0.09, k_landmark_reduce +0.03); a workgroup is kLmWg landmarks either way (the
// camera staging lists are built per kLmWg landmarks), i.e. 128 threads here.
template <int DC, bool MATVEC, bool REC, int LANES = kBsLanes>
__global__ __launch_bounds__(kLmWg * LANES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_back_substitute(BAView v, const double* __restrict__ hinv,
                                                           const double* __restrict__ g_l,
                                                           const double* __restrict__ dc,
                                                           double* __restrict__ dl,
                                                           const double* __restrict__ orec,
                                                           const uint8_t* __restrict__ fix_pt, double* __restrict__ pts_trial) {
    // pts_trial (may be NULL; back-substitution only): the trial point p (+) dl of the landmark is written with its step -- the
    // retraction of the points (k_retract_points: a pass over 3 n_pt values) rides on this kernel (round 5, eager step evaluation)
    constexpr int CAMD = REC ? kCamStride : kCamQStride;
    constexpr int STR = CAMD + DC + ((CAMD + DC) & 1);   // camera | its step, 16-byte pieces
    __shared__ double sCam[kCamStageCap * STR];
    const int g = threadIdx.x & (LANES - 1);
    const int64_t l = ((int64_t)blockIdx.x + v.lm_wg0) * kLmWg + threadIdx.x / LANES;
    const bool active = l < v.n_pt;
    double acc[3] = {0, 0, 0};
    CamStager<STR, DC, CAMD, kLmWg * LANES> stager;   // (see k_landmark_reduce: the staging chain and the landmark's own chain interleaved)
    stager.issue_indices(v);
    const int64_t lc = active ? l : 0;
    const int b = v.pt_ptr[lc], e = active ? v.pt_ptr[lc + 1] : b;
    stager.issue_data(REC ? v.camp : v.camq, dc);
    // REC: the landmark's whole record now (Hll^-1 | point | g_l, one 128-byte line, the four lanes of a landmark read the
    // same address) -- the point comes with it and the epilogue has no load left to wait for
    double lrec[REC ? kLmStride : 1];
    if (REC) load_lm_record(hinv, (size_t)lc, lrec);
    const double pw[3] = {REC ? lrec[REC ? kLmPt : 0] : v.pts[3 * lc], REC ? lrec[REC ? kLmPt + 1 : 0] : v.pts[3 * lc + 1],
                          REC ? lrec[REC ? kLmPt + 2 : 0] : v.pts[3 * lc + 2]};
    const int i_first = max(min(b + g, (int)v.n_obs - 1), 0);
    const double2* __restrict__ rec2 = reinterpret_cast<const double2*>(orec);
    double2 uv_next = REC ? rec2[2 * (size_t)i_first] : v.o_uv[i_first];   // REC: (xn, yn) | (p_w.z, w)
    double2 rw_next = REC ? rec2[2 * (size_t)i_first + 1] : make_double2(0.0, 0.0);
    int sl_next = v.o_slot ? (int)v.o_slot[i_first] : 255;
    stager.store(sCam);
    if (active) {
        for (int i = b + g; i < e; i += LANES) {
            const double2 uv = uv_next, rw = rw_next;
            const int sl = sl_next;
            if (i + LANES < e) {
                uv_next = REC ? rec2[2 * (size_t)(i + LANES)] : v.o_uv[i + LANES];
                if (REC) rw_next = rec2[2 * (size_t)(i + LANES) + 1];
                sl_next = v.o_slot ? (int)v.o_slot[i + LANES] : 255;
            }
            double dcv[DC];
            if (REC) {
                double cv[kCamStride];
                if (sl != 255) {
#pragma unroll
                    for (int a = 0; a < kCamStride; a += 2) {
                        const double2 t = *reinterpret_cast<const double2*>(sCam + sl * STR + a);
                        cv[a] = t.x; cv[a + 1] = t.y;
                    }
#pragma unroll
                    for (int a = 0; a < DC; ++a) dcv[a] = sCam[sl * STR + kCamStride + a];
                } else {
                    const uint32_t c = v.o_cam[i];
#pragma unroll
                    for (int a = 0; a < kCamStride; a += 2) {
                        const double2 t = *reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)c + a);
                        cv[a] = t.x; cv[a + 1] = t.y;
                    }
#pragma unroll
                    for (int a = 0; a < DC; ++a) dcv[a] = dc[(size_t)c * DC + a];
                }
                RecJac J;
                jac_from_rec(cv, uv, rw, pw, J);
                const double q0 = dcv[0] + (dcv[4] * pw[2] - dcv[5] * pw[1]);   // dt + dtheta x p_w
                const double q1 = dcv[1] + (dcv[5] * pw[0] - dcv[3] * pw[2]);
                const double q2 = dcv[2] + (dcv[3] * pw[1] - dcv[4] * pw[0]);
                double tk = 0.0;
                if (DC == 9) tk = J.t[0] * dcv[DC - 3] + J.t[1] * dcv[DC - 2] + J.t[2] * dcv[DC - 1];
                const double s0 = J.a[0][0] * q0 + J.a[0][1] * q1 + J.a[0][2] * q2 + J.xw * tk;
                const double s1 = J.a[1][0] * q0 + J.a[1][1] * q1 + J.a[1][2] * q2 + J.yw * tk;
#pragma unroll
                for (int a = 0; a < 3; ++a) acc[a] += J.a[0][a] * s0 + J.a[1][a] * s1;
                continue;
            }
            Cam cam;
            if (sl != 255) {
                load_cam_q(sCam + sl * STR, v.mask_code, cam);
#pragma unroll
                for (int a = 0; a < DC; ++a) dcv[a] = sCam[sl * STR + kCamQStride + a];
            } else {
                const uint32_t c = v.o_cam[i];
                load_cam_q(v.camq + kCamQStride * (size_t)c, v.mask_code, cam);
#pragma unroll
                for (int a = 0; a < DC; ++a) dcv[a] = dc[(size_t)c * DC + a];
            }
            double r[2], Jc[2][DC], Jl[2][3];
            linearize_obs<DC>(cam, pw, uv.x, uv.y, v.huber_delta, r, Jc, Jl);
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int a = 0; a < DC; ++a) {
                const double d = dcv[a];
                s0 += Jc[0][a] * d; s1 += Jc[1][a] * d;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[a] += Jl[0][a] * s0 + Jl[1][a] * s1;
        }
    }
#pragma unroll
    for (int m = 1; m < LANES; m <<= 1)
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] += __shfl_xor(acc[i], m, LANES);
    if (active && g == 0) {
        double Hi[kLmStride];
        if (REC) {
#pragma unroll
            for (int a = 0; a < kLmStride; ++a) Hi[a] = lrec[REC ? a : 0];
        } else {
            load_lm_record(hinv, (size_t)l, Hi);
        }
        if (MATVEC) {
#pragma unroll
            for (int a = 0; a < 3; ++a) dl[kLmuStride * l + 4 + a] = Hi[3 * a] * acc[0] + Hi[3 * a + 1] * acc[1] + Hi[3 * a + 2] * acc[2];
        } else {
            const double rhs[3] = {-Hi[kLmG] - acc[0], -Hi[kLmG + 1] - acc[1], -Hi[kLmG + 2] - acc[2]};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double d = Hi[3 * a] * rhs[0] + Hi[3 * a + 1] * rhs[1] + Hi[3 * a + 2] * rhs[2];
                dl[3 * l + a] = d;
                if (pts_trial) pts_trial[3 * l + a] = v.pts[3 * l + a] + (fix_pt[3 * l + a] ? 0.0 : 1.0 * d);   // (k_retract_points, sign = 1)
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// A18 camera half of the matrix-free Schur operator, one wave per camera:
//   y_c = lambda x_c + sum_i Jc_i^T (Jc_i x_c - Jl_i u_l)      (= H_cc x - H_cl Hll^-1 H_cl^T x)
// ------------------------------------------------------------------------------------------
template <int DC>
__global__ __launch_bounds__(64) void k_implicit_cam(BAView v, const int* __restrict__ cam_ptr,
                                                       const double* __restrict__ lmu, const double* __restrict__ x,
                                                       double lambda, double* __restrict__ y) {
    const uint32_t c = blockIdx.x;
    Cam cam;
    load_cam_prepared(v.camp + kCamStride * (size_t)c, cam);
    double xc[DC], acc[DC];
#pragma unroll
    for (int a = 0; a < DC; ++a) { xc[a] = x[(size_t)c * DC + a]; acc[a] = 0.0; }
    const int b = cam_ptr[c], e = cam_ptr[c + 1];
    // the next observation's landmark record and measurement are in flight while this one is linearised, the index of the one
    // after too (as in k_cam_reduce): the gather of an iteration starts at once instead of behind a round trip for its address
    // (round 5: the loop had two dependent round trips per observation and nothing to hide them but occupancy)
    const int k_first = max(min(b + (int)threadIdx.x, e - 1), 0);
    uint32_t l_next = v.co_pt[k_first];
    double2 uv_next = v.co_uv[k_first];
    double2 qn0, qn1, qn2, qn3;
    { const double2* q = reinterpret_cast<const double2*>(lmu + kLmuStride * (size_t)l_next); qn0 = q[0]; qn1 = q[1]; qn2 = q[2]; qn3 = q[3]; }
    if (k_first + 64 < e) l_next = v.co_pt[k_first + 64];
    for (int k = b + (int)threadIdx.x; k < e; k += 64) {
        const double2 uv = uv_next;
        const double2 q0 = qn0, q1 = qn1, q2 = qn2, q3 = qn3;
        if (k + 64 < e) {
            uv_next = v.co_uv[k + 64];
            const double2* q = reinterpret_cast<const double2*>(lmu + kLmuStride * (size_t)l_next);
            qn0 = q[0]; qn1 = q[1]; qn2 = q[2]; qn3 = q[3];
            if (k + 128 < e) l_next = v.co_pt[k + 128];
        }
        const double pw[3] = {q0.x, q0.y, q1.x}, u[3] = {q2.x, q2.y, q3.x};
        double r[2], Jc[2][DC], Jl[2][3];
        linearize_obs<DC>(cam, pw, uv.x, uv.y, v.huber_delta, r, Jc, Jl);
        double s0 = -(Jl[0][0] * u[0] + Jl[0][1] * u[1] + Jl[0][2] * u[2]);
        double s1 = -(Jl[1][0] * u[0] + Jl[1][1] * u[1] + Jl[1][2] * u[2]);
#pragma unroll
        for (int a = 0; a < DC; ++a) { s0 += Jc[0][a] * xc[a]; s1 += Jc[1][a] * xc[a]; }
#pragma unroll
        for (int a = 0; a < DC; ++a) acc[a] += Jc[0][a] * s0 + Jc[1][a] * s1;
    }
#pragma unroll
    for (int a = 0; a < DC; ++a) {
        const double t = wave_sum(acc[a]);
        double lam = lambda;
        if (v.cam_scale) { const double sc = v.cam_scale[(size_t)c * DC + a]; lam = lambda / (sc * sc); }
        if (threadIdx.x == 0) y[(size_t)c * DC + a] = t + lam * xc[a];
    }
}

// diagonal DC x DC block of S per camera (lower triangle of the diagonal tile, written by k_cam_reduce with
// with_self) -> compact symmetric array sd[c][DC*DC]
template <int DC>
__global__ __launch_bounds__(256) void k_extract_diag_blocks(int64_t n_cam, TileMap tm, double* __restrict__ sd) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_cam * DC * DC) return;
    const uint32_t c = (uint32_t)(i / (DC * DC));
    const int e = (int)(i - (int64_t)c * DC * DC), a = e / DC, b = e - a * DC;
    const double* blk = s_block_ptr<DC>(tm, c, c);
    sd[i] = (a >= b) ? blk[a * kNB + b] : blk[b * kNB + a];
}

// General inverse by Gauss-Jordan with partial pivoting; false when a pivot is exactly zero
// (nalgebra's try_inverse -> None).
template <int N>
__device__ bool try_inverse(const double* A, double* inv) {
    double M[N][2 * N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) { M[i][j] = A[i * N + j]; M[i][N + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < N; ++c) {
        int piv = c;
        for (int r = c + 1; r < N; ++r)
            if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (M[piv][c] == 0.0) return false;
        if (piv != c)
            for (int j = 0; j < 2 * N; ++j) { const double t = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = t; }
        const double d = M[c][c];
        for (int j = 0; j < 2 * N; ++j) M[c][j] /= d;
        for (int r = 0; r < N; ++r) {
            if (r == c) continue;
            const double f = M[r][c];
            if (f != 0.0)
                for (int j = 0; j < 2 * N; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) inv[i * N + j] = M[i][N + j];
    return true;
}

// Schur-Jacobi preconditioner (compute_schur_jacobi_preconditioner, implicit_schur.rs:456-573): the
// reference inverts S_ii per camera-side VARIABLE, i.e. the 6x6 pose block and the 3x3 intrinsics block
// separately; singular -> + max(1e-6 |trace| / n, 1e-8) I -> identity.  One lane per (camera, block).
template <int N>
__device__ void precond_block(const double* sd, int DC, int off, double* minv) {
    double A[N * N], I[N * N];
    for (int a = 0; a < N; ++a)
        for (int b = 0; b < N; ++b) A[a * N + b] = sd[(off + a) * DC + off + b];
    if (!try_inverse<N>(A, I)) {
        double tr = 0.0;
        for (int a = 0; a < N; ++a) tr += A[a * N + a];
        const double reg = fmax(1e-6 * fabs(tr) / (double)N, 1e-8);
        for (int a = 0; a < N; ++a) A[a * N + a] += reg;
        if (!try_inverse<N>(A, I))
            for (int a = 0; a < N; ++a)
                for (int b = 0; b < N; ++b) I[a * N + b] = (a == b) ? 1.0 : 0.0;
    }
    for (int a = 0; a < N; ++a)
        for (int b = 0; b < N; ++b) minv[(off + a) * DC + off + b] = I[a * N + b];
}

template <int DC>
__global__ __launch_bounds__(64) void k_precond_blocks(int64_t n_cam, const double* __restrict__ sd, double* __restrict__ minv) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t c = t >> 1;
    const int which = (int)(t & 1);
    if (c >= n_cam) return;
    const double* S = sd + (size_t)c * DC * DC;
    double* M = minv + (size_t)c * DC * DC;
    if (which == 0) {
        precond_block<6>(S, DC, 0, M);
    } else if (DC == 9) {
        precond_block<3>(S, DC, 6, M);
        for (int a = 0; a < 6; ++a)
            for (int b = 6; b < 9; ++b) { M[a * DC + b] = 0.0; M[b * DC + a] = 0.0; }
    }
}

// z_c = Minv_c r_c
template <int DC>
__global__ __launch_bounds__(256) void k_precond_apply(int64_t n, const double* __restrict__ minv, const double* __restrict__ r,
                                                         double* __restrict__ z) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t c = i / DC;
    const int a = (int)(i - c * DC);
    const double* M = minv + (size_t)c * DC * DC + a * DC;
    const double* rc = r + (size_t)c * DC;
    double sacc = 0.0;
#pragma unroll
    for (int b = 0; b < DC; ++b) sacc += M[b] * rc[b];
    z[i] = sacc;
}

// ------------------------------------------------------------------------------------------
// A15 retraction.  dc is camera-major [cam][DC] (pose tangent first, then intrinsics).
// ------------------------------------------------------------------------------------------
template <int DC>
__global__ __launch_bounds__(256) void k_retract_cams(int64_t n_cam, const double* __restrict__ poses,
                                                        const double* __restrict__ intr,
                                                        const double* __restrict__ dc, double sign,
                                                        const uint8_t* __restrict__ fix_pose,
                                                        const uint8_t* __restrict__ fix_intr,
                                                        double* __restrict__ poses_out,
                                                        double* __restrict__ intr_out) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cam) return;
    double d[6], p[7], o[7];
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a] = fix_pose[6 * c + a] ? 0.0 : sign * dc[c * DC + a];
#pragma unroll
    for (int a = 0; a < 7; ++a) p[a] = poses[7 * c + a];
    se3_plus(p, d, o);
#pragma unroll
    for (int a = 0; a < 7; ++a) poses_out[7 * c + a] = o[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double di = 0.0;
        if (DC == 9) di = fix_intr[3 * c + a] ? 0.0 : sign * dc[c * DC + 6 + a];
        intr_out[3 * c + a] = intr[3 * c + a] + di;
    }
}

__global__ __launch_bounds__(256) void k_retract_points(int64_t n3, const double* __restrict__ pts,
                                                          const double* __restrict__ dl, double sign,
                                                          const uint8_t* __restrict__ fix_pt,
                                                          double* __restrict__ pts_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    const double d = fix_pt[i] ? 0.0 : sign * dl[i];
    pts_out[i] = pts[i] + d;
}

// ------------------------------------------------------------------------------------------
// A16 cost: per-block partial sums of |r~|^2 (fixed geometry -> reproducible).
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Per-wave camera cache in LDS (k_cost_partial).  A wave walks a CONTIGUOUS range of observations (landmark-major: its
// cameras stay inside a capture window), keeps the compact cameras it has seen in a direct-mapped table of
// kCamCacheSlots entries and gathers hits from LDS -- the per-lane gathers are bound by the bytes they pull through the
// L1, and the LDS is a second, wider path (cost 0.33 -> 0.25 ms on final-13682).  In k_landmark_reduce and
// k_back_substitute the same cache cost a wave of occupancy and gained nothing (0.71 / 0.69 -> 0.73 / 0.72 ms).
constexpr int kCamCacheSlots = 64;
struct CamCache {
    uint32_t tag[kCamCacheSlots];
    double2 data[kCamCacheSlots][kCamQStride / 2];
};
__device__ __forceinline__ void cam_cache_reset(CamCache& cc, int lane) {
    for (int s = lane; s < kCamCacheSlots; s += 64) cc.tag[s] = 0xFFFFFFFFu;
    __builtin_amdgcn_wave_barrier();
}
// Wave-synchronous: the LDS executes a wave's instructions in order, so the hits of a step are read before its misses
// replace entries.  When several missing lanes map to one slot they first all store their camera index to the tag; the
// re-read tells each whether it won (which of the colliding stores lands last is the hardware's choice, and need not be
// the same for a 4-byte and a 16-byte store): only the winner fills the entry.  The tag accesses are volatile so that
// the compiler does not forward a lane's own store to its re-read.
__device__ __forceinline__ void cam_cache_get(CamCache& cc, const double* __restrict__ camq, uint32_t c, bool active, double q[kCamQStride]) {
    const uint32_t slot = c & (kCamCacheSlots - 1);
    volatile __attribute__((address_space(3))) uint32_t* tag = (volatile __attribute__((address_space(3))) uint32_t*)&cc.tag[slot];
    const bool hit = active && *tag == c;
    double2 t[kCamQStride / 2];
    // (typed address spaces: left to itself the compiler selects between the two ADDRESSES and issues flat loads, which
    // take the LDS hits through the vector-memory path)
    typedef double __attribute__((ext_vector_type(2))) f64x2;
    if (hit) {
        const __attribute__((address_space(3))) f64x2* lsrc = (const __attribute__((address_space(3))) f64x2*)&cc.data[slot][0];
#pragma unroll
        for (int k = 0; k < kCamQStride / 2; ++k) { const f64x2 v2 = lsrc[k]; t[k] = make_double2(v2.x, v2.y); }
    } else if (active) {
        const __attribute__((address_space(1))) f64x2* gsrc = (const __attribute__((address_space(1))) f64x2*)(camq + kCamQStride * (size_t)c);
#pragma unroll
        for (int k = 0; k < kCamQStride / 2; ++k) { const f64x2 v2 = gsrc[k]; t[k] = make_double2(v2.x, v2.y); }
    }
    __builtin_amdgcn_wave_barrier();   // (compiler only) no entry is replaced before the hits are read
    const bool miss = active && !hit;
    if (miss) *tag = c;
    __builtin_amdgcn_wave_barrier();
    if (miss && *tag == c) {
#pragma unroll
        for (int k = 0; k < kCamQStride / 2; ++k) cc.data[slot][k] = t[k];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < kCamQStride / 2; ++k) { q[2 * k] = t[k].x; q[2 * k + 1] = t[k].y; }
}

__global__ __launch_bounds__(256) void k_cost_partial(BAView v, double* __restrict__ partial) {
    __shared__ double scratch[4];
    __shared__ CamCache cache[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    CamCache& cc = cache[w];
    cam_cache_reset(cc, lane);
    // contiguous range of this wave: whole 64-observation steps
    const int64_t n_waves = (int64_t)gridDim.x * 4, wave = (int64_t)blockIdx.x * 4 + w;
    const int64_t steps = (v.n_obs + 63) / 64, per = (steps + n_waves - 1) / n_waves;
    const int64_t s0 = wave * per, s1 = min(steps, s0 + per);
    double s = 0.0;
    // software-pipelined: the next step's observation record (camera, landmark, measurement) is in flight while this step's
    // camera and point are gathered and projected
    uint32_t c_next = 0, l_next = 0;
    double2 uv_next = make_double2(0.0, 0.0);
    if (s0 < s1) {
        const int64_t i0 = min(s0 * 64 + lane, v.n_obs - 1);
        c_next = v.o_cam[i0]; l_next = v.o_pt[i0]; uv_next = v.o_uv[i0];
    }
    for (int64_t st = s0; st < s1; ++st) {
        const int64_t i = st * 64 + lane;
        const bool active = i < v.n_obs;
        const uint32_t c = c_next, l = l_next;
        const double2 uv = uv_next;
        if (st + 1 < s1) {
            const int64_t in = min(i + 64, v.n_obs - 1);
            c_next = v.o_cam[in]; l_next = v.o_pt[in]; uv_next = v.o_uv[in];
        }
        double q[kCamQStride];
        cam_cache_get(cc, v.camq, c, active, q);
        Cam cam;
        load_cam_q(q, v.mask_code, cam);
        const double pw[3] = {v.pts[3 * (size_t)l], v.pts[3 * (size_t)l + 1], v.pts[3 * (size_t)l + 2]};
        double r[2];
        residual_obs(cam, pw, uv.x, uv.y, v.huber_delta, r);
        s += active ? r[0] * r[0] + r[1] * r[1] : 0.0;
    }
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// out[k] = sum_i partial[i*stride + k], k < nk, in index order (single block, reproducible)
__global__ __launch_bounds__(256) void k_sum_partials(const double* __restrict__ partial, int n, int nk,
                                                        double* __restrict__ out) {
    __shared__ double scratch[4];
    for (int k = 0; k < nk; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) s += partial[(size_t)i * nk + k];
        s = block_sum_256(s, scratch);
        if (threadIdx.x == 0) out[k] = s;
    }
}

// ------------------------------------------------------------------------------------------
// A14 step statistics over a vector pair (g, d): partial[b] = {sum g^2, sum d^2, sum d (lambda d - g)}
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_step_stats(int64_t n, const double* __restrict__ g,
                                                      const double* __restrict__ d, double lambda,
                                                      const double* __restrict__ gscale, double* __restrict__ partial) {
    __shared__ double scratch[4];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // with Jacobi scaling compute_step_generic prices the UNSCALED step against the SCALED gradient
        // (levenberg_marquardt.rs:746-760): restated as coded
        const double gi = gscale ? g[i] * gscale[i] : g[i], di = d[i];
        a += gi * gi; b += di * di; c += di * (lambda * di - gi);
    }
    a = block_sum_256(a, scratch);
    b = block_sum_256(b, scratch);
    c = block_sum_256(c, scratch);
    if (threadIdx.x == 0) { partial[3 * blockIdx.x] = a; partial[3 * blockIdx.x + 1] = b; partial[3 * blockIdx.x + 2] = c; }
}

// sum of squares of a parameter array (compute_parameter_norm, optimizer/mod.rs:458-467)
__global__ __launch_bounds__(256) void k_sumsq_partial(int64_t n, const double* __restrict__ x,
                                                         double* __restrict__ partial) {
    __shared__ double scratch[4];
    double a = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += x[i] * x[i];
    a = block_sum_256(a, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = a;
}

// ------------------------------------------------------------------------------------------
// Jacobi column scaling (process_jacobian_generic, optimizer/mod.rs:749-763): squared column norms of the
// corrected Jacobian (compute_column_norms, linearizer/mod.rs:229-239).  Runs once per optimize.
// ------------------------------------------------------------------------------------------
template <int DC>
__global__ __launch_bounds__(256) void k_column_norms_sq(BAView v, double* __restrict__ n2_cam, double* __restrict__ n2_pt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= v.n_obs) return;
    const uint32_t c = v.o_cam[i], l = v.o_pt[i];
    const double2 uv = v.o_uv[i];
    Cam cam;
    load_cam_prepared(v.camp + kCamStride * (size_t)c, cam);
    const double pw[3] = {v.pts[3 * (size_t)l], v.pts[3 * (size_t)l + 1], v.pts[3 * (size_t)l + 2]};
    double r[2], Jc[2][DC], Jl[2][3];
    linearize_obs<DC>(cam, pw, uv.x, uv.y, v.huber_delta, r, Jc, Jl);
#pragma unroll
    for (int a = 0; a < DC; ++a) atomicAdd(n2_cam + (size_t)c * DC + a, Jc[0][a] * Jc[0][a] + Jc[1][a] * Jc[1][a]);
#pragma unroll
    for (int a = 0; a < 3; ++a) atomicAdd(n2_pt + 3 * (size_t)l + a, Jl[0][a] * Jl[0][a] + Jl[1][a] * Jl[1][a]);
}

__global__ __launch_bounds__(256) void k_scaling_from_norms_sq(int64_t n, const double* __restrict__ n2, double* __restrict__ scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) scale[i] = 1.0 / (1.0 + sqrt(n2[i]));
}

__global__ __launch_bounds__(256) void k_vec_mul(int64_t n, const double* a, const double* b, double* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

__global__ __launch_bounds__(256) void k_vec_add(int64_t n, const double* a, const double* b, double* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

template <int DC>
__global__ __launch_bounds__(256) void k_scale_diag_blocks(int64_t n_cam, const double* __restrict__ scale, double* __restrict__ sd) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_cam * DC * DC) return;
    const int64_t c = i / (DC * DC);
    const int e = (int)(i - c * DC * DC), a = e / DC, b = e - a * DC;
    sd[i] *= scale[c * DC + a] * scale[c * DC + b];
}

// per-observation corrected residual and Jacobian blocks in the CALLER's observation order
// (parity/debug export: apexgpu_get_residual / apexgpu_get_jacobian_blocks)
template <int DC>
__global__ __launch_bounds__(256) void k_export_linearization(BAView v, const int* __restrict__ o_orig,
                                                                double* __restrict__ r_out,
                                                                double* __restrict__ jc_out,
                                                                double* __restrict__ jl_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= v.n_obs) return;
    const uint32_t c = v.o_cam[i], l = v.o_pt[i];
    const double2 uv = v.o_uv[i];
    Cam cam;
    load_cam_prepared(v.camp + kCamStride * (size_t)c, cam);
    const double pw[3] = {v.pts[3 * (size_t)l], v.pts[3 * (size_t)l + 1], v.pts[3 * (size_t)l + 2]};
    double r[2], Jc[2][DC], Jl[2][3];
    linearize_obs<DC>(cam, pw, uv.x, uv.y, v.huber_delta, r, Jc, Jl);
    const int64_t o = o_orig[i];
    if (r_out) { r_out[2 * o] = r[0]; r_out[2 * o + 1] = r[1]; }
    if (jc_out)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int a = 0; a < DC; ++a) jc_out[(2 * o + rr) * DC + a] = Jc[rr][a];
    if (jl_out)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int a = 0; a < 3; ++a) jl_out[(2 * o + rr) * 3 + a] = Jl[rr][a];
}

// ------------------------------------------------------------------------------------------
// launchers (the only symbols the host code sees)
// ------------------------------------------------------------------------------------------
// workgroups of a landmark-major launch: the rank's own landmark range (BAView::lm_wg0 / lm_wgn; every landmark on a single rank)
static inline int lm_grid(const BAView& v) { return v.lm_wgn > 0 ? v.lm_wgn : (int)((v.n_pt + kLmWg - 1) / kLmWg); }
static inline int grid_for(int64_t n, int per_block, int cap) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (cap > 0 && g > cap) g = cap;
    return (int)g;
}

void launch_cam_reduce(int dc, const BAView& v, const TileMap& tm, const int* cam_ptr, const int* cam_obs,
                       double lambda, int add_lambda, const double* hinv, const double* g_l, int with_self, double* g_c,
                       double* g_red, hipStream_t s) {
    if (v.n_cam == 0) return;
    const bool masked = !(v.mask_code == 7 || (dc == 6 && v.mask_code == 6));   // (the masked form costs a wave per SIMD at d_c = 9)
#define CAM_REDUCE(DCV, MK, WS) hipLaunchKernelGGL((k_cam_reduce<DCV, MK, WS>), dim3((unsigned)v.n_cam), dim3(kCamThreads), 0, s, v, tm, cam_ptr, cam_obs, lambda, add_lambda, hinv, g_l, g_c, g_red)
    if (with_self) {
        if (dc == 9 && !masked) CAM_REDUCE(9, false, true); else if (dc == 9) CAM_REDUCE(9, true, true);
        else if (!masked) CAM_REDUCE(6, false, true); else CAM_REDUCE(6, true, true);
    } else {
        if (dc == 9 && !masked) CAM_REDUCE(9, false, false); else if (dc == 9) CAM_REDUCE(9, true, false);
        else if (!masked) CAM_REDUCE(6, false, false); else CAM_REDUCE(6, true, false);
    }
#undef CAM_REDUCE
}

void launch_landmark_reduce(int dc, const BAView& v, double lambda, double* hinv, double* g_l, int* err_flag, double* lmu,
                            hipStream_t s, double* orec) {
    if (v.n_pt == 0) return;
    const int grid = lm_grid(v);
    if (dc == 9) hipLaunchKernelGGL(k_landmark_reduce<9>, dim3(grid), dim3(kLmWg * kLmLanes), 0, s, v, lambda, hinv, g_l, err_flag, lmu, orec);
    else hipLaunchKernelGGL(k_landmark_reduce<6>, dim3(grid), dim3(kLmWg * kLmLanes), 0, s, v, lambda, hinv, g_l, err_flag, lmu, orec);
}

void launch_prepare_cams(int64_t n_cam, const double* poses, const double* intr, double* camp, int mask_code, hipStream_t s) {
    if (n_cam > 0) hipLaunchKernelGGL(k_prepare_cams, dim3(grid_for(n_cam, 256, 0)), dim3(256), 0, s, n_cam, poses, intr, camp, mask_code);
}


// orec != nullptr: the record form (the records of THIS linearisation, k_landmark_reduce); ignored in the masked modes
static bool rec_form_ok(int dc, const BAView& v, const double* orec) { return orec != nullptr && v.mask_code == (dc == 9 ? 7 : 6); }
void launch_back_substitute(int dc, const BAView& v, const double* hinv, const double* g_l, const double* dcam,
                            double* dl, hipStream_t s, const double* orec, const uint8_t* fix_pt, double* pts_trial) {
    if (v.n_pt == 0) return;
    const int grid = lm_grid(v);
    if (rec_form_ok(dc, v, orec)) {
        if (dc == 9) hipLaunchKernelGGL((k_back_substitute<9, false, true>), dim3(grid), dim3(kLmWg * kBsLanes), 0, s, v, hinv, g_l, dcam, dl, orec, fix_pt, pts_trial);
        else hipLaunchKernelGGL((k_back_substitute<6, false, true>), dim3(grid), dim3(kLmWg * kBsLanes), 0, s, v, hinv, g_l, dcam, dl, orec, fix_pt, pts_trial);
        return;
    }
    if (dc == 9) hipLaunchKernelGGL((k_back_substitute<9, false, false>), dim3(grid), dim3(kLmWg * kBsLanes), 0, s, v, hinv, g_l, dcam, dl, nullptr, fix_pt, pts_trial);
    else hipLaunchKernelGGL((k_back_substitute<6, false, false>), dim3(grid), dim3(kLmWg * kBsLanes), 0, s, v, hinv, g_l, dcam, dl, nullptr, fix_pt, pts_trial);
}

void launch_retract(int dc, int64_t n_cam, int64_t n_pt, const double* poses, const double* intr, const double* pts,
                    const double* dcam, const double* dl, double sign, const uint8_t* fix_pose,
                    const uint8_t* fix_intr, const uint8_t* fix_pt, double* poses_out, double* intr_out,
                    double* pts_out, hipStream_t s) {
    if (n_cam > 0) {
        const int grid = grid_for(n_cam, 256, 0);
        if (dc == 9) hipLaunchKernelGGL(k_retract_cams<9>, dim3(grid), dim3(256), 0, s, n_cam, poses, intr, dcam, sign, fix_pose, fix_intr, poses_out, intr_out);
        else hipLaunchKernelGGL(k_retract_cams<6>, dim3(grid), dim3(256), 0, s, n_cam, poses, intr, dcam, sign, fix_pose, fix_intr, poses_out, intr_out);
    }
    if (n_pt > 0)
        hipLaunchKernelGGL(k_retract_points, dim3(grid_for(3 * n_pt, 256, 0)), dim3(256), 0, s, 3 * n_pt, pts, dl, sign, fix_pt, pts_out);
}

void launch_cost(const BAView& v, double* partial, int n_partial, double* out_sumsq, hipStream_t s) {
    hipLaunchKernelGGL(k_cost_partial, dim3(n_partial), dim3(256), 0, s, v, partial);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, partial, n_partial, 1, out_sumsq);
}

void launch_step_stats(int64_t n, const double* g, const double* d, double lambda, const double* gscale, double* partial,
                       int n_partial, double* out3, hipStream_t s) {
    hipLaunchKernelGGL(k_step_stats, dim3(n_partial), dim3(256), 0, s, n, g, d, lambda, gscale, partial);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, partial, n_partial, 3, out3);
}

void launch_column_norms_sq(int dc, const BAView& v, double* n2_cam, double* n2_pt, hipStream_t s) {
    if (v.n_obs == 0) return;
    const unsigned grid = (unsigned)((v.n_obs + 255) / 256);
    if (dc == 9) hipLaunchKernelGGL(k_column_norms_sq<9>, dim3(grid), dim3(256), 0, s, v, n2_cam, n2_pt);
    else hipLaunchKernelGGL(k_column_norms_sq<6>, dim3(grid), dim3(256), 0, s, v, n2_cam, n2_pt);
}
void launch_scaling_from_norms_sq(int64_t n, const double* n2, double* scale, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_scaling_from_norms_sq, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, n2, scale);
}
void launch_vec_mul(int64_t n, const double* a, const double* b, double* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_vec_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, a, b, out);
}
void launch_vec_add(int64_t n, const double* a, const double* b, double* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_vec_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, a, b, out);
}
void launch_scale_diag_blocks(int dc, int64_t n_cam, const double* scale, double* sd, hipStream_t s) {
    const int64_t n = n_cam * dc * dc;
    if (n == 0) return;
    if (dc == 9) hipLaunchKernelGGL(k_scale_diag_blocks<9>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n_cam, scale, sd);
    else hipLaunchKernelGGL(k_scale_diag_blocks<6>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n_cam, scale, sd);
}

void launch_sumsq(int64_t n, const double* x, double* partial, int n_partial, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_sumsq_partial, dim3(n_partial), dim3(256), 0, s, n, x, partial);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, partial, n_partial, 1, out);
}

// the three clears at the head of an assembly (g_red, g_c, the error flags) as one launch
__global__ __launch_bounds__(256) void k_clear3(double* __restrict__ a, double* __restrict__ b, int64_t n, int* __restrict__ f, int nf) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { a[i] = 0.0; b[i] = 0.0; }
    if (blockIdx.x == 0 && (int)threadIdx.x < nf) f[threadIdx.x] = 0;
}
void launch_clear3(double* a, double* b, int64_t n, int* f, int nf, hipStream_t s) {
    hipLaunchKernelGGL(k_clear3, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(256, (n + 1023) / 1024))), dim3(256), 0, s, a, b, n, f, nf);
}
// set-up (round 5): the measurement lists from the caller's array, on the device -- o_uv[i] = uv[o_orig[i]] (landmark-major),
// then co_uv[k] = o_uv[cam_obs[k]], co_pt[k] = o_pt[cam_obs[k]] (camera-major): 0.6 GB less to build on the host and to upload
__global__ __launch_bounds__(256) void k_gather_uv(int64_t n, const int* __restrict__ idx, const double2* __restrict__ src, double2* __restrict__ dst) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) dst[k] = src[(size_t)idx[k]];
}
__global__ __launch_bounds__(256) void k_gather_u32(int64_t n, const int* __restrict__ idx, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) dst[k] = src[(size_t)idx[k]];
}
void launch_gather_uv(int64_t n, const int* idx, const double* src, double* dst, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather_uv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, idx, reinterpret_cast<const double2*>(src), reinterpret_cast<double2*>(dst));
}
void launch_gather_u32(int64_t n, const int* idx, const uint32_t* src, uint32_t* dst, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, idx, src, dst);
}
// y = S x without S: landmark half (u_l into lmu), then camera half
void launch_implicit_matvec(int dc, const BAView& v, const int* cam_ptr, const double* hinv, double* lmu, const double* x,
                            double lambda, double* y, hipStream_t s, const double* orec) {
    if (v.n_pt > 0) {
        const int grid = lm_grid(v);
        if (rec_form_ok(dc, v, orec)) {
            if (dc == 9) hipLaunchKernelGGL((k_back_substitute<9, true, true>), dim3(grid), dim3(kLmWg * kBsLanes), 0, s, v, hinv, nullptr, x, lmu, orec, nullptr, nullptr);
            else hipLaunchKernelGGL((k_back_substitute<6, true, true>), dim3(grid), dim3(kLmWg * kBsLanes), 0, s, v, hinv, nullptr, x, lmu, orec, nullptr, nullptr);
        } else {
            if (dc == 9) hipLaunchKernelGGL((k_back_substitute<9, true, false>), dim3(grid), dim3(kLmWg * kBsLanes), 0, s, v, hinv, nullptr, x, lmu, nullptr, nullptr, nullptr);
            else hipLaunchKernelGGL((k_back_substitute<6, true, false>), dim3(grid), dim3(kLmWg * kBsLanes), 0, s, v, hinv, nullptr, x, lmu, nullptr, nullptr, nullptr);
        }
    }
    if (dc == 9) hipLaunchKernelGGL(k_implicit_cam<9>, dim3((unsigned)v.n_cam), dim3(64), 0, s, v, cam_ptr, lmu, x, lambda, y);
    else hipLaunchKernelGGL(k_implicit_cam<6>, dim3((unsigned)v.n_cam), dim3(64), 0, s, v, cam_ptr, lmu, x, lambda, y);
}
void launch_extract_diag_blocks(int dc, int64_t n_cam, const TileMap& tm, double* sd, hipStream_t s) {
    const int64_t n = n_cam * dc * dc;
    if (dc == 9) hipLaunchKernelGGL(k_extract_diag_blocks<9>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n_cam, tm, sd);
    else hipLaunchKernelGGL(k_extract_diag_blocks<6>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n_cam, tm, sd);
}
void launch_precond_blocks(int dc, int64_t n_cam, const double* sd, double* minv, hipStream_t s) {
    const unsigned grid = (unsigned)((2 * n_cam + 63) / 64);
    if (dc == 9) hipLaunchKernelGGL(k_precond_blocks<9>, dim3(grid), dim3(64), 0, s, n_cam, sd, minv);
    else hipLaunchKernelGGL(k_precond_blocks<6>, dim3(grid), dim3(64), 0, s, n_cam, sd, minv);
}
void launch_precond_apply(int dc, int64_t n_cam, const double* minv, const double* r, double* z, hipStream_t s) {
    const int64_t n = n_cam * dc;
    if (dc == 9) hipLaunchKernelGGL(k_precond_apply<9>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, minv, r, z);
    else hipLaunchKernelGGL(k_precond_apply<6>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, minv, r, z);
}

// two dot products in one pass: partial[b] = {sum a1 b1, sum a2 b2}
__global__ __launch_bounds__(256) void k_dot2_partial(int64_t n, const double* __restrict__ a1, const double* __restrict__ b1,
                                                        const double* __restrict__ a2, const double* __restrict__ b2,
                                                        double* __restrict__ partial) {
    __shared__ double scratch[4];
    double x = 0.0, y = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        x += a1[i] * b1[i];
        y += a2[i] * b2[i];
    }
    x = block_sum_256(x, scratch);
    y = block_sum_256(y, scratch);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = x; partial[2 * blockIdx.x + 1] = y; }
}
// out[0] = a1.b1, out[1] = a2.b2 (fixed geometry: reproducible)
void launch_dot2(int64_t n, const double* a1, const double* b1, const double* a2, const double* b2, double* partial,
                 int n_partial, double* out, hipStream_t s) {
    const int grid = (int)std::min<int64_t>(n_partial, std::max<int64_t>(1, (n + 1023) / 1024));
    hipLaunchKernelGGL(k_dot2_partial, dim3(grid), dim3(256), 0, s, n, a1, b1, a2, b2, partial);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, partial, grid, 2, out);
}

void launch_sum_partials(const double* partial, int n, int nk, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, partial, n, nk, out);
}

// the device's 3x3 gate + inverse on caller-supplied blocks (apexgpu_debug_invert_blocks): the same function
// k_landmark_reduce calls per landmark, so that the three regimes of explicit_schur.rs:377-442 and the margins of the
// trace / determinant shortcut can be probed on the GPU with exact inputs
__global__ __launch_bounds__(256) void k_debug_invert_blocks(int64_t n, const double* __restrict__ in, double* __restrict__ out,
                                                               int* __restrict__ ok) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double B[9], Bi[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) B[k] = in[9 * i + k];
    const bool good = invert_landmark_block(B, Bi);
#pragma unroll
    for (int k = 0; k < 9; ++k) out[9 * i + k] = good ? Bi[k] : 0.0;
    ok[i] = good ? 1 : 0;
}
void launch_debug_invert_blocks(int64_t n, const double* in, double* out, int* ok, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_debug_invert_blocks, dim3(grid_for(n, 256, 0)), dim3(256), 0, s, n, in, out, ok);
}

void launch_export_linearization(int dc, const BAView& v, const int* o_orig, double* r_out, double* jc_out,
                                 double* jl_out, hipStream_t s) {
    if (v.n_obs == 0) return;
    const int grid = grid_for(v.n_obs, 256, 0);
    if (dc == 9) hipLaunchKernelGGL(k_export_linearization<9>, dim3(grid), dim3(256), 0, s, v, o_orig, r_out, jc_out, jl_out);
    else hipLaunchKernelGGL(k_export_linearization<6>, dim3(grid), dim3(256), 0, s, v, o_orig, r_out, jc_out, jl_out);
}

// (set-up: the first launch of a kernel of this translation unit loads its code object -- tens of milliseconds for the big
// ones; Solver::set_structure pays that on a background thread while the host builds its lists: warm_device_code)
__global__ void k_warm_ba_kernels() {}
void warm_ba_kernels(hipStream_t s) { hipLaunchKernelGGL(k_warm_ba_kernels, dim3(1), dim3(64), 0, s); }

}  // namespace apex
