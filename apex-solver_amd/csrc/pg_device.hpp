// pg_device.hpp -- per-edge SE3 pose-graph math shared by the kernels in pg_kernels.hip.
//
// APEX_HD (host+device) like ba_device.hpp, so tests/host_harness.cpp runs the same code on the CPU.
//
// Reference semantics (file:line under the apex-solver tree):
//   BetweenFactor<SE3>::linearize      src/factors/between_factor.rs:268-322
//       r = Log((k1^-1 k0) * meas),  dr/dk0 = Jr^-1(r) Adj(meas^-1),
//       dr/dk1 = Jr^-1(r) (Adj(meas^-1) (-Adj((k1^-1 k0)^-1)))
//   LieGroup::between                  crates/apex-manifolds/src/lib.rs:401-419
//   SE3 inverse / compose / log / Adj  crates/apex-manifolds/src/se3.rs:242-320, 347-369
//   Q block and Jr^-1 (as coded)       se3.rs:520-558, 652-666
//   SO3 log, Jl^-1                     crates/apex-manifolds/src/so3.rs:313-357, 628-646
//
// Every 6x6 Jacobian on this path is block upper-triangular with equal diagonal blocks,
//     J = [ P  T ]        (Adj = [R, [t]x R; 0, R],  Jr^-1 = [D, B; 0, D])
//         [ 0  P ]
// so a Jacobian travels as the pair (P, T): 18 doubles instead of 36, and J_a^T J_b needs three
// 3x3 products instead of a 6x6x6 one.
#pragma once
#include "ba_device.hpp"

namespace apex {

struct Jac6 {  // [P T; 0 P], row-major 3x3 blocks
    double P[9];
    double T[9];
};

APEX_HD void m3_mul(const double* A, const double* B, double* C) {  // C = A B (C must not alias)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
APEX_HD void m3_tmul(const double* A, const double* B, double* C) {  // C = A^T B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
APEX_HD void hat3(const double v[3], double M[9]) {
    M[0] = 0.0;   M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2];  M[4] = 0.0;   M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0];  M[8] = 0.0;
}
// [t]x R
APEX_HD void hat_mul(const double t[3], const double* R, double* C) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        C[j] = -t[2] * R[3 + j] + t[1] * R[6 + j];
        C[3 + j] = t[2] * R[j] - t[0] * R[6 + j];
        C[6 + j] = -t[1] * R[j] + t[0] * R[3 + j];
    }
}

// pose as the kernels hold it: translation + unit quaternion [w,x,y,z] (k_pg_prepare normalises twice,
// like SE3::from(DVector), se3.rs:107-113, 200-206)
APEX_HD void pose_normalise(const double* __restrict__ v7, double* __restrict__ o7) {
    o7[0] = v7[0]; o7[1] = v7[1]; o7[2] = v7[2];
    double w = v7[3], x = v7[4], y = v7[5], z = v7[6];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const double n = sqrt(w * w + x * x + y * y + z * z);
        w /= n; x /= n; y /= n; z /= n;
    }
    o7[3] = w; o7[4] = x; o7[5] = y; o7[6] = z;
}

// a^-1 for a = (t, q)
APEX_HD void se3_inv(const double t[3], const double q[4], double ti[3], double qi[4]) {
    qi[0] = q[0]; qi[1] = -q[1]; qi[2] = -q[2]; qi[3] = -q[3];
    double r[3];
    quat_rotate(qi, t, r);
    ti[0] = -r[0]; ti[1] = -r[1]; ti[2] = -r[2];
}
// a * b
APEX_HD void se3_mul(const double ta[3], const double qa[4], const double tb[3], const double qb[4], double t[3], double q[4]) {
    double r[3];
    quat_mul(qa, qb, q);
    quat_rotate(qa, tb, r);
    t[0] = r[0] + ta[0]; t[1] = r[1] + ta[1]; t[2] = r[2] + ta[2];
}

// SO3::log (so3.rs:313-357): atan2 form, mirrored for w < 0
APEX_HD void so3_log(const double q[4], double th[3]) {
    const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    double coeff = 2.0;
    if (s2 > kSmallAngle2) {
        const double s = sqrt(s2), c = q[0];
        const double two = 2.0 * (c < 0.0 ? atan2(-s, -c) : atan2(s, c));
        coeff = two / s;
    }
    th[0] = q[1] * coeff; th[1] = q[2] * coeff; th[2] = q[3] * coeff;
}

// Jl^-1(theta) = I - 1/2 K + (1/a - (1+cos t)/(2 t sin t)) K^2 (so3.rs:628-646)
APEX_HD void so3_left_jacobian_inv(const double th[3], double D[9]) {
    const double a = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    double K[9], K2[9];
    hat3(th, K);
    m3_mul(K, K, K2);
    double c2 = 0.0;
    if (a > kSmallAngle2) {
        const double t = sqrt(a);
        c2 = 1.0 / a - (1.0 + cos(t)) / (2.0 * t * sin(t));
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = -0.5 * K[i] + c2 * K2[i];
    D[0] += 1.0; D[4] += 1.0; D[8] += 1.0;
}

// Q(rho, theta) exactly as the reference codes it (se3.rs:520-558), d coefficient included
APEX_HD void se3_q_block(const double rho[3], const double th[3], double Q[9]) {
    double Rk[9], Tk[9];
    hat3(rho, Rk);
    hat3(th, Tk);
    const double t2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    const double a = 0.5;
    double b = 1.0 / 6.0 + 1.0 / 120.0 * t2, c = -1.0 / 24.0 + 1.0 / 720.0 * t2, d = -1.0 / 60.0;
    if (t2 > kSmallAngle2) {
        const double tn = sqrt(t2), tn3 = tn * t2, tn4 = t2 * t2, tn5 = tn3 * t2;
        const double s = sin(tn), co = cos(tn);
        b = (tn - s) / tn3;
        c = (1.0 - t2 / 2.0 - co) / tn4;
        d = (c - 3.0) * (tn - s - tn3 / 6.0) / tn5;
    }
    double tr[9], rt[9], trt[9], rtt[9], trtt[9];
    m3_mul(Tk, Rk, tr);
    m3_mul(Rk, Tk, rt);
    m3_mul(tr, Tk, trt);
    m3_mul(rt, Tk, rtt);
    m3_mul(trt, Tk, trtt);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ij = 3 * i + j, ji = 3 * j + i;
            Q[ij] = Rk[ij] * a + (tr[ij] + rt[ij] + trt[ij]) * b - (rtt[ij] - rtt[ji] - trt[ij] * 3.0) * c - trtt[ij] * d;
        }
}

// residual only: r = Log((k1^-1 k0) * meas); poses are prepared (unit quaternions)
APEX_HD void between_residual(const double* __restrict__ k0, const double* __restrict__ k1, const double* __restrict__ m,
                              double r[6], double tA[3], double qA[4], double D[9]) {
    double t1i[3], q1i[4], tD[3], qD[4];
    se3_inv(k1, k1 + 3, t1i, q1i);
    se3_mul(t1i, q1i, k0, k0 + 3, tA, qA);
    se3_mul(tA, qA, m, m + 3, tD, qD);
    so3_log(qD, r + 3);
    so3_left_jacobian_inv(r + 3, D);
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = D[3 * i] * tD[0] + D[3 * i + 1] * tD[1] + D[3 * i + 2] * tD[2];
}

// residual + both Jacobians
APEX_HD void between_linearize(const double* __restrict__ k0, const double* __restrict__ k1, const double* __restrict__ m,
                               double r[6], Jac6& J0, Jac6& J1) {
    double tA[3], qA[4], D[9];
    between_residual(k0, k1, m, r, tA, qA, D);
    // Jr^-1(r) = [D, B; 0, D],  B = -D Q(-rho,-theta) D
    double B[9];
    {
        const double nrho[3] = {-r[0], -r[1], -r[2]}, nth[3] = {-r[3], -r[4], -r[5]};
        double Q[9], T[9];
        se3_q_block(nrho, nth, Q);
        m3_mul(D, Q, T);
        m3_mul(T, D, B);
#pragma unroll
        for (int i = 0; i < 9; ++i) B[i] = -B[i];
    }
    // Adj(meas^-1) = [Rm, Tm; 0, Rm]
    double Rm[9], Tm[9];
    {
        double tmi[3], qmi[4];
        se3_inv(m, m + 3, tmi, qmi);
        quat_to_rot(qmi, Rm);
        hat_mul(tmi, Rm, Tm);
    }
    // dr/dk0 = Jr^-1 Adj(meas^-1)
    m3_mul(D, Rm, J0.P);
    {
        double X[9], Y[9];
        m3_mul(D, Tm, X);
        m3_mul(B, Rm, Y);
#pragma unroll
        for (int i = 0; i < 9; ++i) J0.T[i] = X[i] + Y[i];
    }
    // Adj(A^-1) = [Ra, Ta; 0, Ra];  d1 = Adj(meas^-1) (-Adj(A^-1)) = -[Rm Ra, Rm Ta + Tm Ra; 0, Rm Ra]
    double M[9], N[9];
    {
        double tAi[3], qAi[4], Ra[9], Ta[9], X[9], Y[9];
        se3_inv(tA, qA, tAi, qAi);
        quat_to_rot(qAi, Ra);
        hat_mul(tAi, Ra, Ta);
        m3_mul(Rm, Ra, M);
        m3_mul(Rm, Ta, X);
        m3_mul(Tm, Ra, Y);
#pragma unroll
        for (int i = 0; i < 9; ++i) { M[i] = -M[i]; N[i] = -(X[i] + Y[i]); }
    }
    // dr/dk1 = Jr^-1 d1
    m3_mul(D, M, J1.P);
    {
        double X[9], Y[9];
        m3_mul(D, N, X);
        m3_mul(B, M, Y);
#pragma unroll
        for (int i = 0; i < 9; ++i) J1.T[i] = X[i] + Y[i];
    }
}

// sqrt(rho') of HuberLoss for squared norm s (loss_functions.rs:364-380); delta <= 0: no loss
APEX_HD double pg_huber_scale(double delta, double s) {
    if (delta > 0.0 && s > delta * delta) return sqrt(delta / sqrt(s));
    return 1.0;
}

// H_ab = J_a^T J_b (6x6 row-major) for J = [P T; 0 P]
APEX_HD void jtj(const Jac6& A, const Jac6& B, double H[36]) {
    double pp[9], pt[9], tp[9], tt[9];
    m3_tmul(A.P, B.P, pp);
    m3_tmul(A.P, B.T, pt);
    m3_tmul(A.T, B.P, tp);
    m3_tmul(A.T, B.T, tt);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            H[6 * i + j] = pp[3 * i + j];
            H[6 * i + 3 + j] = pt[3 * i + j];
            H[6 * (i + 3) + j] = tp[3 * i + j];
            H[6 * (i + 3) + 3 + j] = tt[3 * i + j] + pp[3 * i + j];
        }
}
// g_a = J_a^T r
APEX_HD void jtr(const Jac6& A, const double r[6], double g[6]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        g[i] = A.P[i] * r[0] + A.P[3 + i] * r[1] + A.P[6 + i] * r[2];
        g[3 + i] = (A.T[i] * r[0] + A.T[3 + i] * r[1] + A.T[6 + i] * r[2]) + (A.P[i] * r[3] + A.P[3 + i] * r[4] + A.P[6 + i] * r[5]);
    }
}

}  // namespace apex
