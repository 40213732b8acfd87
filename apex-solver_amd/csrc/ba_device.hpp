// ba_device.hpp -- per-observation bundle-adjustment math shared by every kernel.
//
// All functions are APEX_HD (host+device) so that tests/host_harness.cpp can run the very
// same code on the CPU against the oracle; the kernels in ba_kernels.hip call them per lane.
//
// Reference semantics (file:line under the apex-solver tree):
//   SE3::from(DVector)                crates/apex-manifolds/src/se3.rs:200-206, 107-113
//   SE3::act / SO3::rotation_matrix    se3.rs:322-328 ; so3.rs:193-195, 359-366
//   BALPinholeCameraStrict             crates/apex-camera-models/src/bal_pinhole.rs:154-156,
//                                      273-296, 400-435, 528-556, 649-672
//   ProjectionFactor::evaluate_internal src/factors/projection_factor.rs:184-296
//   Huber + Corrector                  src/core/loss_functions.rs:364-380 ; corrector.rs:143-181
//   3x3 inversion gate                 src/linalg/sparse/explicit_schur.rs:377-442
//   SE3 right-plus retraction          se3.rs:569-583, 272-293 ; so3.rs:558-612
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef APEX_HD
#if defined(__HIPCC__)
#define APEX_HD __host__ __device__ __forceinline__
#else
#define APEX_HD inline
#endif
#endif

namespace apex {

constexpr double kMinDepth = 1e-6;          // apex-camera-models/src/lib.rs:80
constexpr double kSmallAngle2 = 1e-10;      // apex-manifolds/src/lib.rs:61

// Camera as the kernels see it: rotation matrix of the (re)normalised quaternion, translation,
// intrinsics.  k_prepare_cams builds this once per parameter set (16 doubles per camera) so that
// the per-observation kernels pay no quaternion normalisation (2 sqrt + 8 div) per observation.
struct Cam {
    double R[9];  // row-major
    double t[3];
    double f, k1, k2;
    // OptimizeParams<POSE, LANDMARK, INTRINSIC> (src/factors/mod.rs:66-101) as column masks of the factor's Jacobian: a
    // block that is not optimised has no columns in the reference, i.e. zero columns here (its variables then get a zero
    // step from the damped system, exactly as the reference's unreferenced variables do)
    double m_pose = 1.0, m_lm = 1.0, m_intr = 1.0;
};
constexpr int kCamStride = 16;  // doubles per prepared camera: R(9) t(3) f k1 k2 | mask code (4 POSE + 2 LANDMARK + INTRINSIC)
constexpr double kMaskAll = 7.0;

// pose7 = [tx,ty,tz,qw,qx,qy,qz] as VariableEnum::to_vector() stores it (the quaternion may be
// slightly non-unit after a compose); normalised twice like from_translation_quaternion.
APEX_HD void quat_to_rot(const double q[4], double R[9]);

APEX_HD void load_cam(const double* __restrict__ pose7, const double* __restrict__ intr3, Cam& c, double* qn_out = nullptr) {
    c.t[0] = pose7[0]; c.t[1] = pose7[1]; c.t[2] = pose7[2];
    double w = pose7[3], x = pose7[4], y = pose7[5], z = pose7[6];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        double n = sqrt(w * w + x * x + y * y + z * z);
        w /= n; x /= n; y /= n; z /= n;
    }
    const double q[4] = {w, x, y, z};
    quat_to_rot(q, c.R);
    c.f = intr3[0]; c.k1 = intr3[1]; c.k2 = intr3[2];
    if (qn_out) { qn_out[0] = w; qn_out[1] = x; qn_out[2] = y; qn_out[3] = z; }
}

// The compact form of a prepared camera for the kernels that gather one camera PER LANE (the landmark-major ones):
// normalised quaternion (4) t (3) f k1 k2 = 80 bytes = five 16-byte loads instead of eight.  Those kernels are bound by
// the bytes their lanes pull through the L1 (128 of the ~150-220 bytes per observation were the camera); rebuilding R
// from the stored quaternion is the same quat_to_rot call on the same numbers as k_prepare_cams makes.
constexpr int kCamQStride = 10;
APEX_HD void store_cam_q(const double qn[4], const Cam& c, double* __restrict__ o) {
    o[0] = qn[0]; o[1] = qn[1]; o[2] = qn[2]; o[3] = qn[3];
    o[4] = c.t[0]; o[5] = c.t[1]; o[6] = c.t[2];
    o[7] = c.f; o[8] = c.k1; o[9] = c.k2;
}
APEX_HD void load_cam_q(const double* __restrict__ p, int mask_code, Cam& c) {
    const double q[4] = {p[0], p[1], p[2], p[3]};
    quat_to_rot(q, c.R);
    c.t[0] = p[4]; c.t[1] = p[5]; c.t[2] = p[6];
    c.f = p[7]; c.k1 = p[8]; c.k2 = p[9];
    c.m_pose = (mask_code & 4) ? 1.0 : 0.0; c.m_lm = (mask_code & 2) ? 1.0 : 0.0; c.m_intr = (mask_code & 1) ? 1.0 : 0.0;
}

APEX_HD void store_cam_prepared(const Cam& c, double* __restrict__ o) {
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = c.R[i];
    o[9] = c.t[0]; o[10] = c.t[1]; o[11] = c.t[2];
    o[12] = c.f; o[13] = c.k1; o[14] = c.k2;
    o[15] = 4.0 * c.m_pose + 2.0 * c.m_lm + c.m_intr;
}

APEX_HD void load_cam_prepared(const double* __restrict__ p, Cam& c) {
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R[i] = p[i];
    c.t[0] = p[9]; c.t[1] = p[10]; c.t[2] = p[11];
    c.f = p[12]; c.k1 = p[13]; c.k2 = p[14];
    const int code = (int)p[15];
    c.m_pose = (code & 4) ? 1.0 : 0.0; c.m_lm = (code & 2) ? 1.0 : 0.0; c.m_intr = (code & 1) ? 1.0 : 0.0;
}

// p_cam = R p_w + t  (SE3::act, se3.rs:322-328; the reference rotates with the quaternion, this is
// the same rotation through its matrix)
// The per-observation arithmetic below is written with EXPLICIT fma and compiled without further contraction: every
// kernel that linearises an observation must produce the SAME bits for r and J.  Left to the compiler, which of
// a*b + c*d becomes the fma depends on the surrounding code, g_l (k_landmark_reduce) and W (k_cam_reduce) then come from
// Jacobians that differ in the last bit, and cond(S) ~ 1e9 turns that into 1e-7 in the step (seen twice while
// restructuring kernels: the step moved from 5e-8 to 2e-7 against the oracle with bit-identical inputs).
#if defined(__clang__)
#define APEX_FP_EXACT _Pragma("clang fp contract(off)")
#else
#define APEX_FP_EXACT
#endif
APEX_HD void cam_transform(const Cam& c, const double pw[3], double pc[3]) {
    APEX_FP_EXACT
    pc[0] = fma(c.R[0], pw[0], fma(c.R[1], pw[1], fma(c.R[2], pw[2], c.t[0])));
    pc[1] = fma(c.R[3], pw[0], fma(c.R[4], pw[1], fma(c.R[5], pw[2], c.t[1])));
    pc[2] = fma(c.R[6], pw[0], fma(c.R[7], pw[1], fma(c.R[8], pw[2], c.t[2])));
}

APEX_HD void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// q v q*  as nalgebra's UnitQuaternion * Vector3: t = 2 (qv x v); t*w + qv x t + v
APEX_HD void quat_rotate(const double q[4], const double v[3], double o[3]) {
    double t[3], c[3];
    cross3(q + 1, v, t);
    t[0] *= 2.0; t[1] *= 2.0; t[2] *= 2.0;
    cross3(q + 1, t, c);
    o[0] = t[0] * q[0] + c[0] + v[0];
    o[1] = t[1] * q[0] + c[1] + v[1];
    o[2] = t[2] * q[0] + c[2] + v[2];
}

APEX_HD void quat_to_rot(const double q[4], double R[9]) {
    // No FMA contraction here: k_prepare_cams and the kernels that rebuild R from the compact camera (load_cam_q) must
    // produce the SAME bits -- a last-bit difference between the R behind g_l and the R behind W is amplified by
    // cond(S) in the step (measured: 5e-8 -> 6e-7 against the oracle).
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    double w = q[0], i = q[1], j = q[2], k = q[3];
    double ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    double ij = i * j * 2.0, wk = w * k * 2.0, wj = w * j * 2.0;
    double ik = i * k * 2.0, jk = j * k * 2.0, wi = w * i * 2.0;
    R[0] = ww + ii - jj - kk; R[1] = ij - wk;           R[2] = wj + ik;
    R[3] = wk + ij;           R[4] = ww - ii + jj - kk; R[5] = jk - wi;
    R[6] = ik - wj;           R[7] = wi + jk;           R[8] = ww - ii - jj + kk;
}

APEX_HD void quat_mul(const double a[4], const double b[4], double o[4]) {
    double c[3];
    cross3(a + 1, b + 1, c);
    o[0] = a[0] * b[0] - (a[1] * b[1] + a[2] * b[2] + a[3] * b[3]);
    o[1] = a[0] * b[1] + b[0] * a[1] + c[0];
    o[2] = a[0] * b[2] + b[0] * a[2] + c[1];
    o[3] = a[0] * b[3] + b[0] * a[3] + c[2];
}

// Reciprocal and reciprocal square root.  On the device: the hardware approximations refined by two Newton steps (full
// double precision, a third of the instructions of the IEEE division / square-root sequences: every per-observation
// kernel executes them once or twice per observation); on the host (tests/host_harness.cpp) the plain operations.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double apex_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double apex_rsqrt(double x) {   // x > 0
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    return y;
}
#else
inline double apex_rcp(double x) { return 1.0 / x; }
inline double apex_rsqrt(double x) { return 1.0 / sqrt(x); }
#endif

// Huber weight sqrt(rho'(s)) for s = |r|^2 (delta <= 0: no loss function).  For Huber
// rho'' <= 0, hence alpha = 0 and both J and r are scaled by sqrt(rho') (corrector.rs:156-162).
APEX_HD double huber_sqrt_rho1(double delta, double s) {
    if (delta > 0.0 && s > delta * delta) {
#if defined(__HIP_DEVICE_COMPILE__)
        const double t = delta * apex_rsqrt(s);   // delta / sqrt(s)
        return t * apex_rsqrt(t);                 // sqrt(t)
#else
        return sqrt(delta / sqrt(s));
#endif
    }
    return 1.0;
}

// Residual only (A16).  Returns validity; r is the CORRECTED residual.
APEX_HD bool residual_obs(const Cam& c, const double pw[3], double u_obs, double v_obs,
                          double huber_delta, double r[2]) {
    APEX_FP_EXACT
    double pc[3];
    cam_transform(c, pw, pc);
    if (!(pc[2] < -kMinDepth)) { r[0] = 0.0; r[1] = 0.0; return false; }
    double inz = -apex_rcp(pc[2]);
    double xn = pc[0] * inz, yn = pc[1] * inz;
    double r2 = fma(xn, xn, yn * yn), r4 = r2 * r2;
    double d = fma(c.k2, r4, fma(c.k1, r2, 1.0));
    double r0 = fma(c.f, xn * d, -u_obs);
    double r1 = fma(c.f, yn * d, -v_obs);
    double w = huber_sqrt_rho1(huber_delta, fma(r0, r0, r1 * r1));
    r[0] = r0 * w; r[1] = r1 * w;
    return true;
}

// Full linearisation of one observation (A1+A2+A4).  DC = 6: camera block = pose only
// (BundleAdjustment keys [pose,pt]); DC = 9: [pose | intrinsics] (SelfCalibration).
// Jc is 2 x DC (row-major [row][col]), Jl is 2 x 3; both CORRECTED (scaled by sqrt(rho')).
// MASKED = false: every block the factor has columns for is optimised (SelfCalibration; BundleAdjustment at DC = 6) -- the
// column masks are all ones and are not read (the register-critical kernels instantiate this form for those modes).
// rec4 (optional): the observation's PROJECTION RECORD (xn, yn, p_w.z, sqrt(rho')) -- with -1/z, which jac_from_rec rebuilds
// bit for bit from the point and the camera by this function's own operations, everything the Jacobian needs besides
// the camera (R, t, f, k1, k2) and the point.  (Rounds 3-4a stored -1/z in slot 2; the point's third component rides there
// now so that the pair kernel finds what it needs of a landmark -- six entries of Hll^-1 and p_w.x, p_w.y -- in ONE 64-byte line.)
// the the record form of the Schur pair kernel (schur_pairs.hip, jac_from_rec)
// rebuilds J from it instead of re-linearising the observation once per pair.  A point behind the camera: weight 0.
template <int DC, bool MASKED = true>
APEX_HD bool linearize_obs(const Cam& c, const double pw[3], double u_obs, double v_obs,
                           double huber_delta, double r[2], double Jc[2][DC], double Jl[2][3], double* rec4 = nullptr) {
    APEX_FP_EXACT
    double pc[3];
    cam_transform(c, pw, pc);
    if (!(pc[2] < -kMinDepth)) {
        if (rec4) { rec4[0] = 0.0; rec4[1] = 0.0; rec4[2] = pw[2]; rec4[3] = 0.0; }
        r[0] = 0.0; r[1] = 0.0;
#pragma unroll
        for (int a = 0; a < DC; ++a) { Jc[0][a] = 0.0; Jc[1][a] = 0.0; }
#pragma unroll
        for (int a = 0; a < 3; ++a) { Jl[0][a] = 0.0; Jl[1][a] = 0.0; }
        return false;
    }
    const double f = c.f, k1 = c.k1, k2 = c.k2;
    double inz = -apex_rcp(pc[2]);
    double xn = pc[0] * inz, yn = pc[1] * inz;
    double r2 = fma(xn, xn, yn * yn), r4 = r2 * r2;
    double dist = fma(k2, r4, fma(k1, r2, 1.0));
    double r0 = fma(f, xn * dist, -u_obs);
    double r1 = fma(f, yn * dist, -v_obs);
    // d(u,v)/d p_cam  (bal_pinhole.rs:400-435)
    double dd = fma(2.0 * k2, r2, k1);
    double dxn_dz = xn * inz, dyn_dz = yn * inz;
    const double tx = (xn * dd) * 2.0, ty = (yn * dd) * 2.0;
    double dxd_dxn = fma(tx, xn, dist);
    double dxd_dyn = tx * yn;
    double dyd_dxn = ty * xn;
    double dyd_dyn = fma(ty, yn, dist);
    double Jp[2][3];
    Jp[0][0] = f * (dxd_dxn * inz);
    Jp[0][1] = f * (dxd_dyn * inz);
    Jp[0][2] = f * fma(dxd_dxn, dxn_dz, dxd_dyn * dyn_dz);
    Jp[1][0] = f * (dyd_dxn * inz);
    Jp[1][1] = f * (dyd_dyn * inz);
    Jp[1][2] = f * fma(dyd_dxn, dxn_dz, dyd_dyn * dyn_dz);
    const double* R = c.R;
    double w = huber_sqrt_rho1(huber_delta, fma(r0, r0, r1 * r1));
    if (rec4) { rec4[0] = xn; rec4[1] = yn; rec4[2] = pw[2]; rec4[3] = w; }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        // landmark block = Jp R ; pose block = [Jp R | -(Jp R)[p_w]x]  (bal_pinhole.rs:528-556:
        // d p_cam/d delta = [R | -R [p_w]x], right perturbation delta = [rho; theta])
        double a0 = fma(Jp[rr][0], R[0], fma(Jp[rr][1], R[3], Jp[rr][2] * R[6]));
        double a1 = fma(Jp[rr][0], R[1], fma(Jp[rr][1], R[4], Jp[rr][2] * R[7]));
        double a2 = fma(Jp[rr][0], R[2], fma(Jp[rr][1], R[5], Jp[rr][2] * R[8]));
        const double wl = MASKED ? w * c.m_lm : w, wp = MASKED ? w * c.m_pose : w;
        Jl[rr][0] = a0 * wl; Jl[rr][1] = a1 * wl; Jl[rr][2] = a2 * wl;
        Jc[rr][0] = a0 * wp; Jc[rr][1] = a1 * wp; Jc[rr][2] = a2 * wp;
        Jc[rr][3] = fma(a2, pw[1], -(a1 * pw[2])) * wp;
        Jc[rr][4] = fma(a0, pw[2], -(a2 * pw[0])) * wp;
        Jc[rr][5] = fma(a1, pw[0], -(a0 * pw[1])) * wp;
    }
    if (DC == 9) {
        // d(u,v)/d(f,k1,k2)  (bal_pinhole.rs:649-672)
        const double wi = MASKED ? w * c.m_intr : w;
        Jc[0][DC - 3] = (xn * dist) * wi; Jc[0][DC - 2] = (f * xn * r2) * wi; Jc[0][DC - 1] = (f * xn * r4) * wi;
        Jc[1][DC - 3] = (yn * dist) * wi; Jc[1][DC - 2] = (f * yn * r2) * wi; Jc[1][DC - 1] = (f * yn * r4) * wi;
    }
    r[0] = r0 * w; r[1] = r1 * w;
    return true;
}

// ---- the Jacobian of one observation from its projection record (device code only) ---------------------------------
// J follows from the record and the camera without the projection:
//     Jl = a = (f w inz) [dxx dxy .; dxy dyy .] R      (2 x 3, corrected)
//     Jc = [ a | -a [p_w]x | (xn w, yn w)^T (dist, f r2, f r4) ]                                    (2 x 9)
#if defined(__HIPCC__)
struct RecJac {          // what one observation contributes to a pair: a (= Jl), the intrinsics factors
    double a[2][3];
    double xw, yw;        // xn w, yn w
    double t[3];          // dist, f r2, f r4
};
// cv: the staged camera (R row-major at 0..8, t at 9..11, f k1 k2 at 12..14); pw: the landmark (pw[2] == r23.x).
// -1/z is linearize_obs' own: the third row of cam_transform, then -apex_rcp -- same operations, same operands, same bits.  A
// point behind the camera has weight 0 in its record and any finite -1/z gives the zero Jacobian.
APEX_HD double rec_inz(const double* __restrict__ cv, const double pw[3], double w) {
    APEX_FP_EXACT
    const double pc2 = fma(cv[6], pw[0], fma(cv[7], pw[1], fma(cv[8], pw[2], cv[11])));
    return w == 0.0 ? 1.0 : -apex_rcp(pc2);
}
__device__ __forceinline__ void jac_from_rec(const double* __restrict__ cv, const double2 r01, const double2 r23, const double pw[3], RecJac& o) {
    const double xn = r01.x, yn = r01.y, w = r23.y;
    const double inz = rec_inz(cv, pw, w);
    const double f = cv[12], k1 = cv[13], k2 = cv[14];
    const double r2 = fma(yn, yn, xn * xn), r4 = r2 * r2;
    const double dist = fma(k2, r4, fma(k1, r2, 1.0));
    const double t2 = fma(4.0 * k2, r2, 2.0 * k1);            // 2 d(dist)/d(r2)
    const double txn = t2 * xn;
    const double dxx = fma(txn, xn, dist), dxy = txn * yn, dyy = fma(t2 * yn, yn, dist);
    const double s = (f * w) * inz;
    const double J00 = s * dxx, J01 = s * dxy, J11 = s * dyy;
    const double J02 = fma(xn, J00, yn * J01), J12 = fma(xn, J01, yn * J11);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o.a[0][c] = fma(J00, cv[c], fma(J01, cv[3 + c], J02 * cv[6 + c]));
        o.a[1][c] = fma(J01, cv[c], fma(J11, cv[3 + c], J12 * cv[6 + c]));
    }
    o.xw = xn * w; o.yw = yn * w;
    o.t[0] = dist; o.t[1] = f * r2; o.t[2] = f * r4;
}
#endif

// ---- 3x3 symmetric block: eigenvalue gate + inverse (A9) ---------------------------------
// B (symmetric, row-major 9) -> Binv.  Returns false iff the (regularised) matrix has a zero
// determinant (LinAlgError::SingularMatrix in the reference).
APEX_HD void sym3_eig_minmax(const double B[9], double& mn, double& mx) {
    double a[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = B[i];
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        if (off < 1e-300) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                double apq = a[3 * p + q];
                if (apq != 0.0) {
                    double app = a[3 * p + p], aqq = a[3 * q + q];
                    double tau = (aqq - app) / (2.0 * apq);
                    double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        double akp = a[3 * k + p], akq = a[3 * k + q];
                        a[3 * k + p] = cs * akp - sn * akq;
                        a[3 * k + q] = sn * akp + cs * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        double apk = a[3 * p + k], aqk = a[3 * q + k];
                        a[3 * p + k] = cs * apk - sn * aqk;
                        a[3 * q + k] = sn * apk + cs * aqk;
                    }
                }
            }
    }
    mn = fmin(a[0], fmin(a[4], a[8]));
    mx = fmax(a[0], fmax(a[4], a[8]));
}

APEX_HD bool mat3_try_inverse(const double m[9], double o[9]) {
    double m11 = m[0], m12 = m[1], m13 = m[2];
    double m21 = m[3], m22 = m[4], m23 = m[5];
    double m31 = m[6], m32 = m[7], m33 = m[8];
    double minor_m12_m23 = m22 * m33 - m32 * m23;
    double minor_m11_m23 = m21 * m33 - m31 * m23;
    double minor_m11_m22 = m21 * m32 - m31 * m22;
    double det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
    if (det == 0.0) return false;
    o[0] = minor_m12_m23 / det;
    o[1] = (m13 * m32 - m33 * m12) / det;
    o[2] = (m12 * m23 - m22 * m13) / det;
    o[3] = -minor_m11_m23 / det;
    o[4] = (m11 * m33 - m31 * m13) / det;
    o[5] = (m13 * m21 - m23 * m11) / det;
    o[6] = minor_m11_m22 / det;
    o[7] = (m12 * m31 - m32 * m11) / det;
    o[8] = (m11 * m22 - m21 * m12) / det;
    return true;
}

// invert_landmark_blocks_with_lambda with lambda argument 0.0 (explicit_schur.rs:365-367).
//
// The reference decides between three regimes from the eigenvalues (min_ev < 1e-12; cond > 1e10;
// else plain inverse).  For a symmetric positive definite block the cheap bounds
//     max_ev <= trace ,  min_ev >= det / max_ev^2 >= det / trace^2
// prove "plain inverse" whenever det >= 1e-12 trace^2 and trace^3 <= 1e10 det -- true for every
// well-observed landmark -- so the iterative eigen-solver (tens of microseconds of a wave: 30 sweeps
// of sqrt/div) only runs for the rare blocks where the bounds are inconclusive.  The decision is
// identical to the eigenvalue test wherever the shortcut applies.
APEX_HD bool invert_landmark_block(const double B[9], double Binv[9]) {
    const double tr = B[0] + B[4] + B[8];
    const double det = B[0] * (B[4] * B[8] - B[5] * B[7]) - B[1] * (B[3] * B[8] - B[5] * B[6]) +
                       B[2] * (B[3] * B[7] - B[4] * B[6]);
    const bool pd = B[0] > 0.0 && (B[0] * B[4] - B[1] * B[3]) > 0.0 && det > 0.0;  // Sylvester
    if (pd && det >= 2e-12 * tr * tr && tr * tr * tr <= 0.5e10 * det) return mat3_try_inverse(B, Binv);
    double mn, mx, M[9];
    sym3_eig_minmax(B, mn, mx);
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = B[i];
    if (mn < 1e-12) {
        double reg = 1e-6 + mx * 1e-6;  // lambda.max(1e-6) + max_ev * 1e-6 with lambda = 0
        M[0] += reg; M[4] += reg; M[8] += reg;
    } else if (mx / mn > 1e10) {
        double reg = mx * 1e-6;
        M[0] += reg; M[4] += reg; M[8] += reg;
    }
    return mat3_try_inverse(M, Binv);
}

// ---- SE3 right-plus (A15) -------------------------------------------------------------
APEX_HD void so3_exp(const double th[3], double q[4]) {
    double t2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    if (t2 > kSmallAngle2) {
        double hx = th[0] / 2.0, hy = th[1] / 2.0, hz = th[2] / 2.0;
        double n = sqrt(hx * hx + hy * hy + hz * hz);
        double s = sin(n) / n;
        q[0] = cos(n); q[1] = hx * s; q[2] = hy * s; q[3] = hz * s;
    } else {
        double w = 1.0, x = th[0] / 2.0, y = th[1] / 2.0, z = th[2] / 2.0;
        double n = sqrt(w * w + x * x + y * y + z * z);
        q[0] = w / n; q[1] = x / n; q[2] = y / n; q[3] = z / n;
    }
}

// V(theta) rho  with V = left Jacobian of SO(3) (so3.rs:595-612)
APEX_HD void so3_left_jacobian_mul(const double th[3], const double rho[3], double o[3]) {
    double a = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    double k1[3], k2[3];
    cross3(th, rho, k1);  // [th]x rho
    cross3(th, k1, k2);   // [th]x^2 rho
    if (a <= kSmallAngle2) {
        o[0] = rho[0] + 0.5 * k1[0]; o[1] = rho[1] + 0.5 * k1[1]; o[2] = rho[2] + 0.5 * k1[2];
        return;
    }
    double theta = sqrt(a), s = sin(theta), c = cos(theta);
    double c1 = (1.0 - c) / a, c2 = (theta - s) / (a * theta);
    o[0] = rho[0] + c1 * k1[0] + c2 * k2[0];
    o[1] = rho[1] + c1 * k1[1] + c2 * k2[1];
    o[2] = rho[2] + c1 * k1[2] + c2 * k2[2];
}

// pose7' = pose7 (+) delta6, stored un-normalised exactly like the reference keeps it.
APEX_HD void se3_plus(const double pose[7], const double delta[6], double out[7]) {
    double qe[4], te[3], qn[4], rt[3];
    so3_exp(delta + 3, qe);
    so3_left_jacobian_mul(delta + 3, delta, te);
    quat_mul(pose + 3, qe, qn);
    quat_rotate(pose + 3, te, rt);
    out[0] = rt[0] + pose[0]; out[1] = rt[1] + pose[1]; out[2] = rt[2] + pose[2];
    out[3] = qn[0]; out[4] = qn[1]; out[5] = qn[2]; out[6] = qn[3];
}

}  // namespace apex
