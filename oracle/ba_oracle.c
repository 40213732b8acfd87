/*
 * ba_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, fp64, CPU restatement of the apex-solver bundle-adjustment inner loop
 * (reference: amin-abouee/apex-solver v1.3.0, Rust).  It exists so that the HIP
 * path can be checked against the reference's algorithm on identical inputs.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (apex-solver_amd/) never does.
 *
 * Pinning status.  The Rust reference cannot be built here (no cargo/rustc) and it
 * holds no BAL-scale golden vectors, so this file is pinned by
 *   (i)  every known-answer unit test the reference has for this path, transcribed
 *        in tests/test_oracle_kat.py (explicit_schur.rs:1454-1576,1647-1663,
 *        1914-1959; bal_pinhole.rs:818-844,904-962; projection_factor.rs:396-522;
 *        corrector.rs:309-349; optimizer/mod.rs:991-997; se3.rs:1111-1134;
 *        levenberg_marquardt.rs:1566-1618), and
 *   (ii) an independent numpy/scipy restatement (tests/np_ref.py: scipy.sparse
 *        J^T J and a direct solve of the full damped normal equations, central-
 *        difference Jacobians), with the agreed outputs committed as fixtures in
 *        tests/golden/.
 * At BAL scale (per-iteration dx, cost) the reference itself asserts only
 * convergence, so parity there is "pinned by restatement", not by reference output.
 *
 * Third-party arithmetic that is not under /root/reference (crates are not
 * vendored; Cargo.lock is git-ignored so only the semver ranges are known):
 *   nalgebra 0.33  UnitQuaternion product / to_rotation_matrix / q*v /
 *                  from_scaled_axis, Matrix3::try_inverse, symmetric_eigenvalues
 *   faer 0.24      sparse J^T J, SymbolicLlt/Llt, solve
 * Their published algorithms are restated here (see the per-function notes); they
 * affect rounding only, except at the documented thresholds.
 *
 * Every function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORA_OK 0
#define ORA_ERR_SINGULAR (-2)      /* LinAlgError::SingularMatrix */
#define ORA_ERR_FACTORIZATION (-1) /* LinAlgError::FactorizationFailed */
#define ORA_ERR_INPUT (-5)         /* LinAlgError::InvalidInput */

#define MIN_DEPTH 1e-6              /* crates/apex-camera-models/src/lib.rs:80 */
#define SMALL_ANGLE_THRESHOLD 1e-10 /* crates/apex-manifolds/src/lib.rs:61 */

/* ------------------------------------------------------------------------- */
/* A3: SE3 / SO3 pieces                                                       */
/* ------------------------------------------------------------------------- */

/* SE3::from(DVector) (se3.rs:200-206) -> from_translation_quaternion (se3.rs:107-113):
 * quaternion.normalize() then UnitQuaternion::from_quaternion (normalises again).
 * v = [tx,ty,tz,qw,qx,qy,qz]; q out = (w,x,y,z). */
static void se3_from_vec(const double v[7], double t[3], double q[4]) {
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2];
    double w = v[3], x = v[4], y = v[5], z = v[6];
    for (int pass = 0; pass < 2; ++pass) {
        double n = sqrt(w * w + x * x + y * y + z * z);
        w /= n; x /= n; y /= n; z /= n;
    }
    q[0] = w; q[1] = x; q[2] = y; q[3] = z;
}

static void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* SO3::act (so3.rs:359-366) = nalgebra UnitQuaternion * Vector3:
 * t = 2 (qv x v); result = t*w + qv x t + v. */
static void quat_rotate(const double q[4], const double v[3], double o[3]) {
    double t[3], c[3];
    cross3(q + 1, v, t);
    t[0] *= 2.0; t[1] *= 2.0; t[2] *= 2.0;
    cross3(q + 1, t, c);
    o[0] = t[0] * q[0] + c[0] + v[0];
    o[1] = t[1] * q[0] + c[1] + v[1];
    o[2] = t[2] * q[0] + c[2] + v[2];
}

/* SO3::rotation_matrix (so3.rs:193-195) = nalgebra to_rotation_matrix; R row-major. */
static void quat_to_rot(const double q[4], double R[9]) {
    double w = q[0], i = q[1], j = q[2], k = q[3];
    double ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    double ij = i * j * 2.0, wk = w * k * 2.0, wj = w * j * 2.0;
    double ik = i * k * 2.0, jk = j * k * 2.0, wi = w * i * 2.0;
    R[0] = ww + ii - jj - kk; R[1] = ij - wk;           R[2] = wj + ik;
    R[3] = wk + ij;           R[4] = ww - ii + jj - kk; R[5] = jk - wi;
    R[6] = ik - wj;           R[7] = wi + jk;           R[8] = ww - ii - jj + kk;
}

/* SE3::act (se3.rs:322-328): R p + t */
static void se3_act(const double t[3], const double q[4], const double p[3], double o[3]) {
    quat_rotate(q, p, o);
    o[0] += t[0]; o[1] += t[1]; o[2] += t[2];
}

/* Hamilton product as nalgebra Quaternion*Quaternion (SO3::compose so3.rs:280-299). */
static void quat_mul(const double a[4], const double b[4], double o[4]) {
    double c[3];
    cross3(a + 1, b + 1, c);
    o[0] = a[0] * b[0] - (a[1] * b[1] + a[2] * b[2] + a[3] * b[3]);
    o[1] = a[0] * b[1] + b[0] * a[1] + c[0];
    o[2] = a[0] * b[2] + b[0] * a[2] + c[1];
    o[3] = a[0] * b[3] + b[0] * a[3] + c[2];
}

/* SO3Tangent::exp (so3.rs:558-578).  theta^2 > 1e-10: from_scaled_axis =
 * exp of the pure quaternion theta/2; else normalised (1, theta/2). */
static void so3_exp(const double th[3], double q[4]) {
    double t2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    if (t2 > SMALL_ANGLE_THRESHOLD) {
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

/* SO3Tangent::left_jacobian (so3.rs:595-612), row-major 3x3. */
static void so3_left_jacobian(const double th[3], double V[9]) {
    double a = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    double K[9] = {0.0, -th[2], th[1], th[2], 0.0, -th[0], -th[1], th[0], 0.0};
    if (a <= SMALL_ANGLE_THRESHOLD) {
        for (int i = 0; i < 9; ++i) V[i] = 0.5 * K[i];
        V[0] += 1.0; V[4] += 1.0; V[8] += 1.0;
        return;
    }
    double theta = sqrt(a), s = sin(theta), c = cos(theta);
    double c1 = (1.0 - c) / a, c2 = (theta - s) / (a * theta);
    double K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += K[3 * i + k] * K[3 * k + j];
            K2[3 * i + j] = acc;
        }
    for (int i = 0; i < 9; ++i) V[i] = c1 * K[i] + c2 * K2[i];
    V[0] += 1.0; V[4] += 1.0; V[8] += 1.0;
}

/* A15: SE3 right-plus  T' = T o Exp(delta)  (lib.rs:269-283, se3.rs:569-583, 272-293).
 * pose7 in/out = [t, qw,qx,qy,qz] exactly as VariableEnum::to_vector stores it (the
 * composed quaternion is NOT renormalised; the next SE3::from() does that). */
void ora_se3_plus(const double pose[7], const double delta[6], double out[7]) {
    double t[3], q[4];
    /* var.value is an SE3 built by SE3::from / compose; its quaternion is what is stored */
    t[0] = pose[0]; t[1] = pose[1]; t[2] = pose[2];
    q[0] = pose[3]; q[1] = pose[4]; q[2] = pose[5]; q[3] = pose[6];
    double qe[4], V[9], te[3];
    so3_exp(delta + 3, qe);
    so3_left_jacobian(delta + 3, V);
    for (int i = 0; i < 3; ++i) te[i] = V[3 * i] * delta[0] + V[3 * i + 1] * delta[1] + V[3 * i + 2] * delta[2];
    double qn[4], rt[3];
    quat_mul(q, qe, qn);
    quat_rotate(q, te, rt);
    out[0] = rt[0] + t[0]; out[1] = rt[1] + t[1]; out[2] = rt[2] + t[2];
    out[3] = qn[0]; out[4] = qn[1]; out[5] = qn[2]; out[6] = qn[3];
}

/* ------------------------------------------------------------------------- */
/* A2: BALPinholeCameraStrict                                                 */
/* ------------------------------------------------------------------------- */

/* project (bal_pinhole.rs:273-296); returns 0 when z >= -MIN_DEPTH (:154-156). */
int ora_bal_project(const double intr[3], const double pc[3], double uv[2]) {
    if (!(pc[2] < -MIN_DEPTH)) return 0;
    double inz = -1.0 / pc[2];
    double xn = pc[0] * inz, yn = pc[1] * inz;
    double r2 = xn * xn + yn * yn, r4 = r2 * r2;
    double d = 1.0 + intr[1] * r2 + intr[2] * r4;
    uv[0] = intr[0] * (xn * d);
    uv[1] = intr[0] * (yn * d);
    return 1;
}

/* jacobian_point (bal_pinhole.rs:400-435): d(u,v)/d p_cam, row-major 2x3. */
void ora_bal_jacobian_point(const double intr[3], const double pc[3], double J[6]) {
    double f = intr[0], k1 = intr[1], k2 = intr[2];
    double inz = -1.0 / pc[2];
    double xn = pc[0] * inz, yn = pc[1] * inz;
    double r2 = xn * xn + yn * yn, r4 = r2 * r2;
    double dist = 1.0 + k1 * r2 + k2 * r4;
    double dd = k1 + 2.0 * k2 * r2;
    double dxn_dz = xn * inz, dyn_dz = yn * inz;
    double dxd_dxn = dist + xn * dd * 2.0 * xn;
    double dxd_dyn = xn * dd * 2.0 * yn;
    double dyd_dxn = yn * dd * 2.0 * xn;
    double dyd_dyn = dist + yn * dd * 2.0 * yn;
    J[0] = f * (dxd_dxn * inz);
    J[1] = f * (dxd_dyn * inz);
    J[2] = f * (dxd_dxn * dxn_dz + dxd_dyn * dyn_dz);
    J[3] = f * (dyd_dxn * inz);
    J[4] = f * (dyd_dyn * inz);
    J[5] = f * (dyd_dxn * dxn_dz + dyd_dyn * dyn_dz);
}

/* jacobian_intrinsics (bal_pinhole.rs:649-672): d(u,v)/d(f,k1,k2), row-major 2x3. */
void ora_bal_jacobian_intrinsics(const double intr[3], const double pc[3], double J[6]) {
    double f = intr[0], k1 = intr[1], k2 = intr[2];
    double inz = -1.0 / pc[2];
    double xn = pc[0] * inz, yn = pc[1] * inz;
    double r2 = xn * xn + yn * yn, r4 = r2 * r2;
    double dist = 1.0 + k1 * r2 + k2 * r4;
    J[0] = xn * dist; J[1] = f * xn * r2; J[2] = f * xn * r4;
    J[3] = yn * dist; J[4] = f * yn * r2; J[5] = f * yn * r4;
}

/* ------------------------------------------------------------------------- */
/* A4: Huber + Corrector                                                      */
/* ------------------------------------------------------------------------- */

/* HuberLoss::evaluate (loss_functions.rs:364-380) + Corrector::new
 * (corrector.rs:143-181).  Returns sqrt(rho'), sets *alpha_sq_norm.  For Huber
 * rho'' <= 0 always, so alpha_sq_norm = 0 and residual_scaling = sqrt(rho'). */
double ora_huber_corrector(double delta, double s, double *residual_scaling, double *alpha_sq_norm) {
    double rho1, rho2;
    if (s > delta * delta) {
        double r = sqrt(s);
        rho1 = delta / r;
        if (rho1 < -1.7976931348623157e308) rho1 = -1.7976931348623157e308; /* .max(f64::MIN) */
        rho2 = -rho1 / (2.0 * s);
    } else {
        rho1 = 1.0; rho2 = 0.0;
    }
    double sq = sqrt(rho1);
    if (s == 0.0 || rho2 <= 0.0) {
        *residual_scaling = sq; *alpha_sq_norm = 0.0;
        return sq;
    }
    double d = 1.0 + 2.0 * s * rho2 / rho1;
    if (d < 0.0) d = 0.0;
    double alpha = 1.0 - sqrt(d);
    *residual_scaling = sq / (1.0 - alpha);
    *alpha_sq_norm = alpha / s;
    return sq;
}

/* ------------------------------------------------------------------------- */
/* A1: one observation (ProjectionFactor::linearize / evaluate_internal,       */
/*     projection_factor.rs:306-364, 184-296) + loss correction               */
/*     (linearizer/mod.rs:143-149)                                            */
/* ------------------------------------------------------------------------- */
/* Jpose 2x6, Jpt 2x3, Jintr 2x3 row-major; any may be NULL when !want_jac.
 * huber_delta <= 0: no loss function.  Returns 1 if the projection was valid. */
int ora_linearize_obs(const double pose[7], const double intr[3], const double pt[3],
                      const double uv_obs[2], double huber_delta, int want_jac,
                      double r[2], double Jpose[12], double Jpt[6], double Jintr[6]) {
    double t[3], q[4], pc[3], uv[2];
    se3_from_vec(pose, t, q);
    se3_act(t, q, pt, pc);
    if (want_jac) {
        memset(Jpose, 0, 12 * sizeof(double));
        memset(Jpt, 0, 6 * sizeof(double));
        memset(Jintr, 0, 6 * sizeof(double));
    }
    if (!ora_bal_project(intr, pc, uv)) {
        /* invalid projection: zero residual, zero Jacobian rows (:227-238) */
        r[0] = 0.0; r[1] = 0.0;
        /* the corrector still runs on s = 0: scale sqrt(rho'(0)) = 1, nothing changes */
        return 0;
    }
    r[0] = uv[0] - uv_obs[0];
    r[1] = uv[1] - uv_obs[1];
    if (want_jac) {
        double Jp[6], R[9], D[18];
        /* jacobian_pose (bal_pinhole.rs:528-556): d p_cam / d delta = [R | -R [p_w]x] */
        ora_bal_jacobian_point(intr, pc, Jp);
        quat_to_rot(q, R);
        double S[9] = {0.0, -pt[2], pt[1], pt[2], 0.0, -pt[0], -pt[1], pt[0], 0.0};
        for (int rr = 0; rr < 3; ++rr)
            for (int c = 0; c < 6; ++c) {
                if (c < 3) D[6 * rr + c] = R[3 * rr + c];
                else {
                    double acc = 0.0;
                    for (int k = 0; k < 3; ++k) acc += R[3 * rr + k] * S[3 * k + (c - 3)];
                    D[6 * rr + c] = -acc;
                }
            }
        for (int rr = 0; rr < 2; ++rr)
            for (int c = 0; c < 6; ++c) {
                double acc = 0.0;
                for (int k = 0; k < 3; ++k) acc += Jp[3 * rr + k] * D[6 * k + c];
                Jpose[6 * rr + c] = acc;
            }
        /* landmark block: d_uv_d_pcam * R (projection_factor.rs:262-276) */
        for (int rr = 0; rr < 2; ++rr)
            for (int c = 0; c < 3; ++c) {
                double acc = 0.0;
                for (int k = 0; k < 3; ++k) acc += Jp[3 * rr + k] * R[3 * k + c];
                Jpt[3 * rr + c] = acc;
            }
        ora_bal_jacobian_intrinsics(intr, pc, Jintr);
    }
    if (huber_delta > 0.0) {
        double s = r[0] * r[0] + r[1] * r[1];
        double rs, a2;
        double sq = ora_huber_corrector(huber_delta, s, &rs, &a2);
        if (want_jac) {
            /* correct_jacobian (corrector.rs:233-254); a2 == 0 for Huber */
            if (a2 == 0.0) {
                for (int i = 0; i < 12; ++i) Jpose[i] *= sq;
                for (int i = 0; i < 6; ++i) { Jpt[i] *= sq; Jintr[i] *= sq; }
            } else {
                double *blocks[3] = {Jpose, Jpt, Jintr};
                int widths[3] = {6, 3, 3};
                for (int b = 0; b < 3; ++b)
                    for (int c = 0; c < widths[b]; ++c) {
                        double j0 = blocks[b][c], j1 = blocks[b][widths[b] + c];
                        double rtj = r[0] * j0 + r[1] * j1;
                        blocks[b][c] = (j0 - r[0] * rtj * a2) * sq;
                        blocks[b][widths[b] + c] = (j1 - r[1] * rtj * a2) * sq;
                    }
            }
        }
        r[0] *= rs; r[1] *= rs; /* correct_residuals (corrector.rs:292-298) */
    }
    return 1;
}

/* ------------------------------------------------------------------------- */
/* A9: 3x3 landmark-block inversion with the eigenvalue gate                  */
/*     (explicit_schur.rs:365-442)                                            */
/* ------------------------------------------------------------------------- */

/* Eigenvalues of a symmetric 3x3 by cyclic Jacobi (nalgebra uses tridiagonal QR;
 * only min/max versus the 1e-12 and 1e10 thresholds matter). */
static void sym3_eigenvalues(const double B[9], double ev[3]) {
    double a[9];
    memcpy(a, B, sizeof a);
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = a[3 * p + q];
                if (apq == 0.0) continue;
                double app = a[3 * p + p], aqq = a[3 * q + q];
                double tau = (aqq - app) / (2.0 * apq);
                double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = a[3 * k + p], akq = a[3 * k + q];
                    a[3 * k + p] = c * akp - s * akq;
                    a[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = a[3 * p + k], aqk = a[3 * q + k];
                    a[3 * p + k] = c * apk - s * aqk;
                    a[3 * q + k] = s * apk + c * aqk;
                }
            }
    }
    ev[0] = a[0]; ev[1] = a[4]; ev[2] = a[8];
}

/* nalgebra Matrix3::try_inverse: cofactor formula, fails iff determinant == 0. */
static int mat3_try_inverse(const double m[9], double o[9]) {
    double m11 = m[0], m12 = m[1], m13 = m[2];
    double m21 = m[3], m22 = m[4], m23 = m[5];
    double m31 = m[6], m32 = m[7], m33 = m[8];
    double minor_m12_m23 = m22 * m33 - m32 * m23;
    double minor_m11_m23 = m21 * m33 - m31 * m23;
    double minor_m11_m22 = m21 * m32 - m31 * m22;
    double det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
    if (det == 0.0) return 0;
    o[0] = minor_m12_m23 / det;
    o[1] = (m13 * m32 - m33 * m12) / det;
    o[2] = (m12 * m23 - m22 * m13) / det;
    o[3] = -minor_m11_m23 / det;
    o[4] = (m11 * m33 - m31 * m13) / det;
    o[5] = (m13 * m21 - m23 * m11) / det;
    o[6] = minor_m11_m22 / det;
    o[7] = (m12 * m31 - m32 * m11) / det;
    o[8] = (m11 * m22 - m21 * m12) / det;
    return 1;
}

/* invert_landmark_blocks_with_lambda (explicit_schur.rs:377-442); the LM path calls it
 * with lambda = 0.0 (:365-367, :1215).  blocks/out: n x 9 row-major. */
int ora_invert_landmark_blocks(int64_t n, const double *blocks, double lambda, double *out) {
    const double COND = 1e10, MIN_EV = 1e-12, REG = 1e-6;
    for (int64_t i = 0; i < n; ++i) {
        const double *B = blocks + 9 * i;
        double ev[3], M[9];
        sym3_eigenvalues(B, ev);
        double mn = fmin(ev[0], fmin(ev[1], ev[2]));
        double mx = fmax(ev[0], fmax(ev[1], ev[2]));
        memcpy(M, B, sizeof M);
        if (mn < MIN_EV) {
            double reg = fmax(lambda, REG) + mx * REG;
            M[0] += reg; M[4] += reg; M[8] += reg;
        } else if (mx / mn > COND) {
            double reg = mx * REG;
            M[0] += reg; M[4] += reg; M[8] += reg;
        }
        if (!mat3_try_inverse(M, out + 9 * i)) return ORA_ERR_SINGULAR;
    }
    return ORA_OK;
}

/* ------------------------------------------------------------------------- */
/* A10: compute_schur_complement (explicit_schur.rs:771-925)                  */
/* ------------------------------------------------------------------------- */
/* Hcc: dense row-major n_c x n_c (already damped by the caller, :1186-1205).
 * H_cl is given landmark by landmark as the reference's merged row list:
 * rows row_ptr[l]..row_ptr[l+1], each (cam_row index, [v0 v1 v2]), ascending index.
 * S out: dense row-major, symmetrised, |v| <= 1e-12 set to 0 (the reference drops
 * those entries when it converts to CSC, :913-921). */
void ora_schur_complement(int64_t n_c, const double *Hcc, int64_t n_pt, const int64_t *row_ptr,
                          const int64_t *cam_rows, const double *hcl_vals, const double *hll_inv,
                          double *S) {
    memcpy(S, Hcc, (size_t)n_c * (size_t)n_c * sizeof(double));
    double *contrib = NULL;
    int64_t cap = 0;
    for (int64_t l = 0; l < n_pt; ++l) {
        int64_t b = row_ptr[l], e = row_ptr[l + 1], n = e - b;
        if (n == 0) continue;
        if (n > cap) { cap = 2 * n; contrib = (double *)realloc(contrib, (size_t)cap * 3 * sizeof(double)); }
        const double *Hi = hll_inv + 9 * l;
        for (int64_t i = 0; i < n; ++i) {
            const double *h = hcl_vals + 3 * (b + i);
            contrib[3 * i + 0] = h[0] * Hi[0] + h[1] * Hi[3] + h[2] * Hi[6];
            contrib[3 * i + 1] = h[0] * Hi[1] + h[1] * Hi[4] + h[2] * Hi[7];
            contrib[3 * i + 2] = h[0] * Hi[2] + h[1] * Hi[5] + h[2] * Hi[8];
        }
        for (int64_t i = 0; i < n; ++i) {
            double *Srow = S + cam_rows[b + i] * n_c;
            const double *ci = contrib + 3 * i;
            for (int64_t j = 0; j < n; ++j) {
                const double *hj = hcl_vals + 3 * (b + j);
                Srow[cam_rows[b + j]] -= ci[0] * hj[0] + ci[1] * hj[1] + ci[2] * hj[2];
            }
        }
    }
    free(contrib);
    for (int64_t i = 0; i < n_c; ++i)
        for (int64_t j = i + 1; j < n_c; ++j) {
            double avg = (S[i * n_c + j] + S[j * n_c + i]) * 0.5;
            S[i * n_c + j] = avg; S[j * n_c + i] = avg;
        }
    for (int64_t i = 0; i < n_c * n_c; ++i)
        if (!(fabs(S[i]) > 1e-12)) S[i] = 0.0;
}

/* A11: compute_reduced_gradient (explicit_schur.rs:928-977).  g_c, g_p are the
 * NEGATIVE gradient blocks (:1152-1155, :1166). */
void ora_reduced_gradient(int64_t n_c, const double *g_c, int64_t n_pt, const double *g_p,
                          const int64_t *row_ptr, const int64_t *cam_rows, const double *hcl_vals,
                          const double *hll_inv, double *g_red) {
    double *acc = (double *)calloc((size_t)n_c, sizeof(double));
    for (int64_t l = 0; l < n_pt; ++l) {
        const double *Hi = hll_inv + 9 * l, *g = g_p + 3 * l;
        double y[3];
        for (int i = 0; i < 3; ++i) y[i] = Hi[3 * i] * g[0] + Hi[3 * i + 1] * g[1] + Hi[3 * i + 2] * g[2];
        /* the reference sweeps CSC columns (3 per landmark); same sums, column-by-column */
        for (int c = 0; c < 3; ++c)
            for (int64_t i = row_ptr[l]; i < row_ptr[l + 1]; ++i)
                acc[cam_rows[i]] += hcl_vals[3 * i + c] * y[c];
    }
    for (int64_t i = 0; i < n_c; ++i) g_red[i] = g_c[i] - acc[i];
    free(acc);
}

/* A11: back_substitute (explicit_schur.rs:980-1029). */
void ora_back_substitute(int64_t n_pt, const double *delta_c, const double *g_p,
                         const int64_t *row_ptr, const int64_t *cam_rows, const double *hcl_vals,
                         const double *hll_inv, double *delta_p) {
    for (int64_t l = 0; l < n_pt; ++l) {
        double rhs[3];
        for (int c = 0; c < 3; ++c) {
            double acc = 0.0;
            for (int64_t i = row_ptr[l]; i < row_ptr[l + 1]; ++i)
                acc += hcl_vals[3 * i + c] * delta_c[cam_rows[i]];
            rhs[c] = g_p[3 * l + c] - acc;
        }
        const double *Hi = hll_inv + 9 * l;
        for (int i = 0; i < 3; ++i)
            delta_p[3 * l + i] = Hi[3 * i] * rhs[0] + Hi[3 * i + 1] * rhs[1] + Hi[3 * i + 2] * rhs[2];
    }
}

/* ------------------------------------------------------------------------- */
/* A12: solve_with_cholesky (explicit_schur.rs:539-634)                       */
/* ------------------------------------------------------------------------- */

/* Dense lower Cholesky in place (row-major, lower triangle), blocked for speed.
 * faer's Llt fails on a non-positive pivot; so does this.  Returns 0 on success. */
static int dense_llt(int64_t n, double *A) {
    const int64_t NB = 64;
    for (int64_t k0 = 0; k0 < n; k0 += NB) {
        int64_t kb = (k0 + NB < n) ? NB : n - k0;
        /* factor the diagonal block */
        for (int64_t j = k0; j < k0 + kb; ++j) {
            double d = A[j * n + j];
            for (int64_t p = k0; p < j; ++p) d -= A[j * n + p] * A[j * n + p];
            if (!(d > 0.0)) return 1;
            d = sqrt(d);
            A[j * n + j] = d;
            for (int64_t i = j + 1; i < k0 + kb; ++i) {
                double s = A[i * n + j];
                for (int64_t p = k0; p < j; ++p) s -= A[i * n + p] * A[j * n + p];
                A[i * n + j] = s / d;
            }
        }
        /* panel: rows below solve against the diagonal block */
#pragma omp parallel for schedule(static)
        for (int64_t i = k0 + kb; i < n; ++i) {
            for (int64_t j = k0; j < k0 + kb; ++j) {
                double s = A[i * n + j];
                for (int64_t p = k0; p < j; ++p) s -= A[i * n + p] * A[j * n + p];
                A[i * n + j] = s / A[j * n + j];
            }
        }
        /* trailing update (lower part only) */
#pragma omp parallel for schedule(dynamic, 8)
        for (int64_t i = k0 + kb; i < n; ++i) {
            const double *Li = A + i * n + k0;
            for (int64_t j = k0 + kb; j <= i; ++j) {
                const double *Lj = A + j * n + k0;
                double s = 0.0;
                for (int64_t p = 0; p < kb; ++p) s += Li[p] * Lj[p];
                A[i * n + j] -= s;
            }
        }
    }
    return 0;
}

static void dense_llt_solve(int64_t n, const double *L, double *x) {
    for (int64_t i = 0; i < n; ++i) {
        double s = x[i];
        for (int64_t p = 0; p < i; ++p) s -= L[i * n + p] * x[p];
        x[i] = s / L[i * n + i];
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int64_t p = i + 1; p < n; ++p) s -= L[p * n + i] * x[p];
        x[i] = s / L[i * n + i];
    }
}

/* S dense row-major symmetric; b, x length n.  On failure of the plain factorisation
 * retries with S + reg*I, reg = max(trace/n, max|diag|, 1) * 10^(k-4), k = 0..4.
 * *reg_used receives the regularisation that succeeded (0 for the first attempt). */
int ora_solve_cholesky(int64_t n, const double *S, const double *b, double *x, double *reg_used) {
    double *L = (double *)malloc((size_t)n * (size_t)n * sizeof(double));
    if (!L) return ORA_ERR_INPUT;
    if (reg_used) *reg_used = 0.0;
    memcpy(L, S, (size_t)n * (size_t)n * sizeof(double));
    if (dense_llt(n, L) == 0) {
        memcpy(x, b, (size_t)n * sizeof(double));
        dense_llt_solve(n, L, x);
        free(L);
        return ORA_OK;
    }
    double trace = 0.0, max_diag = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        trace += S[i * n + i];
        max_diag = fmax(max_diag, fabs(S[i * n + i]));
    }
    double base = fmax(fmax(trace / (double)n, max_diag), 1.0);
    for (int attempt = 0; attempt < 5; ++attempt) {
        double reg = base * pow(10.0, (double)(attempt - 4));
        memcpy(L, S, (size_t)n * (size_t)n * sizeof(double));
        for (int64_t i = 0; i < n; ++i) L[i * n + i] += reg;
        if (dense_llt(n, L) == 0) {
            memcpy(x, b, (size_t)n * sizeof(double));
            dense_llt_solve(n, L, x);
            if (reg_used) *reg_used = reg;
            free(L);
            return ORA_OK;
        }
    }
    free(L);
    return ORA_ERR_SINGULAR;
}

/* ---- the SPARSE form of solve_with_cholesky --------------------------------------------------
 * The reference does not factorise S densely: compute_schur_complement returns S as a sparse
 * matrix with every |v| <= 1e-12 dropped (explicit_schur.rs:913-921) and solve_with_cholesky hands
 * it to faer's sparse LL^T (SymbolicLlt::try_new + Llt::try_new_with_symbolic, :544-550), i.e. a
 * fill-reducing ordering and a factor that lives inside the pattern's fill.  faer is not in the
 * reference tree; what is restated here is that contract with the textbook tools: the pattern
 * of the sparsified S at block granularity (`blk` = the camera block, the unit every non-zero
 * of S comes in), a reverse Cuthill-McKee order of that graph, and a blocked Cholesky confined
 * to the ENVELOPE of the permuted matrix (the fill of an envelope stays inside it).  The numbers
 * differ from ora_solve_cholesky's only by the order of the sums (tests/test_oracle_kat.py holds
 * the two together); the work is n * band^2 instead of n^3 / 3.  This is the solve bench.py's
 * cpu_baseline times ("solve": "sparse"); parity keeps using the dense one, whose summation
 * order does not depend on an ordering heuristic.  Regularisation ladder as above. */
static void rcm_order(int64_t nb, const int64_t *ptr, const int64_t *adj, int64_t *perm /* new -> old */) {
    int64_t *deg = (int64_t *)malloc((size_t)nb * 8), *q = (int64_t *)malloc((size_t)nb * 8);
    int64_t *lvl = (int64_t *)malloc((size_t)nb * 8);
    char *seen = (char *)calloc((size_t)nb, 1);
    for (int64_t v = 0; v < nb; ++v) deg[v] = ptr[v + 1] - ptr[v];
    int64_t n_out = 0;
    for (int64_t s0 = 0; s0 < nb; ++s0) {
        if (seen[s0]) continue;
        /* pseudo-peripheral start of this component: repeat BFS from the last, lowest-degree node of the deepest level */
        int64_t start = s0;
        for (int round = 0; round < 4; ++round) {
            int64_t h = 0, t = 0, depth = 0;
            for (int64_t v = 0; v < nb; ++v) lvl[v] = -1;
            q[t++] = start; lvl[start] = 0;
            while (h < t) {
                const int64_t v = q[h++];
                for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e) {
                    const int64_t w = adj[e];
                    if (!seen[w] && lvl[w] < 0) { lvl[w] = lvl[v] + 1; if (lvl[w] > depth) depth = lvl[w]; q[t++] = w; }
                }
            }
            int64_t best = start;
            for (int64_t i = 0; i < t; ++i)
                if (lvl[q[i]] == depth && (best == start || deg[q[i]] < deg[best])) best = q[i];
            if (best == start) break;
            start = best;
        }
        /* Cuthill-McKee: BFS, neighbours by increasing degree */
        int64_t h = n_out, t = n_out;
        perm[t++] = start; seen[start] = 1;
        while (h < t) {
            const int64_t v = perm[h++];
            const int64_t t0 = t;
            for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e) {
                const int64_t w = adj[e];
                if (!seen[w]) { seen[w] = 1; perm[t++] = w; }
            }
            for (int64_t i = t0 + 1; i < t; ++i) {   /* insertion sort of the new nodes by degree */
                const int64_t w = perm[i];
                int64_t j = i;
                while (j > t0 && deg[perm[j - 1]] > deg[w]) { perm[j] = perm[j - 1]; --j; }
                perm[j] = w;
            }
        }
        n_out = t;
    }
    for (int64_t i = 0; i < nb / 2; ++i) { const int64_t a = perm[i]; perm[i] = perm[nb - 1 - i]; perm[nb - 1 - i] = a; }
    free(deg); free(q); free(lvl); free(seen);
}

/* Cholesky of a dense-stored matrix whose lower triangle lives in the envelope first[i] <= j <= i
 * (first[] per row; last[j] = the last row whose envelope reaches column j, non-decreasing): the
 * blocked right-looking algorithm of dense_llt with every loop cut to the rows a block column can touch. */
static int envelope_llt(int64_t n, double *A, const int64_t *last) {
    const int64_t NB = 64;
    for (int64_t k0 = 0; k0 < n; k0 += NB) {
        const int64_t kb = (k0 + NB < n) ? NB : n - k0;
        const int64_t rend = last[k0 + kb - 1] + 1;   /* rows >= rend have zeros in columns < k0 + kb */
        for (int64_t j = k0; j < k0 + kb; ++j) {
            double d = A[j * n + j];
            for (int64_t p = k0; p < j; ++p) d -= A[j * n + p] * A[j * n + p];
            if (!(d > 0.0)) return 1;
            d = sqrt(d);
            A[j * n + j] = d;
            for (int64_t i = j + 1; i < k0 + kb; ++i) {
                double s = A[i * n + j];
                for (int64_t p = k0; p < j; ++p) s -= A[i * n + p] * A[j * n + p];
                A[i * n + j] = s / d;
            }
        }
#pragma omp parallel for schedule(static)
        for (int64_t i = k0 + kb; i < rend; ++i) {
            for (int64_t j = k0; j < k0 + kb; ++j) {
                double s = A[i * n + j];
                for (int64_t p = k0; p < j; ++p) s -= A[i * n + p] * A[j * n + p];
                A[i * n + j] = s / A[j * n + j];
            }
        }
#pragma omp parallel for schedule(dynamic, 8)
        for (int64_t i = k0 + kb; i < rend; ++i) {
            const double *Li = A + i * n + k0;
            for (int64_t j = k0 + kb; j <= i; ++j) {
                const double *Lj = A + j * n + k0;
                double s = 0.0;
                for (int64_t p = 0; p < kb; ++p) s += Li[p] * Lj[p];
                A[i * n + j] -= s;
            }
        }
    }
    return 0;
}

/* S dense row-major symmetric (as compute_schur_complement builds it); blk: block size of the pattern
 * (n % blk == 0, else 1 is used).  stats (may be NULL): [0] non-zero blocks of the sparsified S (lower, incl.
 * diagonal), [1] scalar entries inside the envelope, [2] half bandwidth (scalars) after the ordering. */
int ora_solve_cholesky_sparse(int64_t n, int64_t blk, const double *S, const double *b, double *x, double *reg_used,
                              double *stats) {
    if (blk < 1 || n % blk != 0) blk = 1;
    const int64_t nb = n / blk;
    if (reg_used) *reg_used = 0.0;
    /* pattern of the sparsified S (:913-921: |v| > 1e-12 kept), symmetric, block granularity */
    char *nz = (char *)calloc((size_t)nb * (size_t)nb, 1);
#pragma omp parallel for schedule(static)
    for (int64_t bi = 0; bi < nb; ++bi)
        for (int64_t bj = 0; bj <= bi; ++bj) {
            char any = 0;
            for (int64_t r = 0; r < blk && !any; ++r)
                for (int64_t c = 0; c < blk; ++c) {
                    const double v = 0.5 * (S[(bi * blk + r) * n + bj * blk + c] + S[(bj * blk + c) * n + bi * blk + r]);
                    if (fabs(v) > 1e-12) { any = 1; break; }
                }
            nz[bi * nb + bj] = any;
        }
    int64_t *ptr = (int64_t *)calloc((size_t)nb + 1, 8);
    int64_t n_blocks = 0;
    for (int64_t bi = 0; bi < nb; ++bi)
        for (int64_t bj = 0; bj < bi; ++bj)
            if (nz[bi * nb + bj]) { ++ptr[bi + 1]; ++ptr[bj + 1]; ++n_blocks; }
    for (int64_t v = 0; v < nb; ++v) ptr[v + 1] += ptr[v];
    int64_t *adj = (int64_t *)malloc((size_t)(ptr[nb] ? ptr[nb] : 1) * 8), *fill = (int64_t *)calloc((size_t)nb, 8);
    for (int64_t bi = 0; bi < nb; ++bi)
        for (int64_t bj = 0; bj < bi; ++bj)
            if (nz[bi * nb + bj]) { adj[ptr[bi] + fill[bi]++] = bj; adj[ptr[bj] + fill[bj]++] = bi; }
    int64_t *perm = (int64_t *)malloc((size_t)nb * 8), *inv = (int64_t *)malloc((size_t)nb * 8);
    rcm_order(nb, ptr, adj, perm);
    for (int64_t i = 0; i < nb; ++i) inv[perm[i]] = i;
    /* envelope of the permuted pattern */
    int64_t *firstb = (int64_t *)malloc((size_t)nb * 8), *last = (int64_t *)malloc((size_t)n * 8);
    for (int64_t i = 0; i < nb; ++i) {
        int64_t f = i;
        const int64_t v = perm[i];
        for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e) if (inv[adj[e]] < f) f = inv[adj[e]];
        firstb[i] = f;
    }
    {
        int64_t *lastb = (int64_t *)malloc((size_t)nb * 8);
        for (int64_t j = 0; j < nb; ++j) lastb[j] = j;
        for (int64_t i = 0; i < nb; ++i)
            for (int64_t j = firstb[i]; j <= i; ++j) if (lastb[j] < i) lastb[j] = i;
        for (int64_t j = 1; j < nb; ++j) if (lastb[j] < lastb[j - 1]) lastb[j] = lastb[j - 1];
        for (int64_t j = 0; j < n; ++j) last[j] = lastb[j / blk] * blk + blk - 1;
        free(lastb);
    }
    double env = 0.0, band = 0.0;
    for (int64_t i = 0; i < nb; ++i) {
        env += (double)(i - firstb[i] + 1) * (double)(blk * blk);
        if ((double)(i - firstb[i]) * (double)blk > band) band = (double)(i - firstb[i]) * (double)blk;
    }
    if (stats) { stats[0] = (double)(n_blocks + nb); stats[1] = env; stats[2] = band; }
    /* P S P^T (lower envelope only), the permuted right-hand side */
    double *A = (double *)calloc((size_t)n * (size_t)n, 8), *L = (double *)calloc((size_t)n * (size_t)n, 8);   /* zero outside the envelope, and they stay zero */
    double *pb = (double *)malloc((size_t)n * 8);
    if (!A || !L) { free(A); free(L); free(pb); free(nz); free(ptr); free(adj); free(fill); free(perm); free(inv); free(firstb); free(last); return ORA_ERR_INPUT; }
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = 0; i < nb; ++i)
        for (int64_t j = firstb[i]; j <= i; ++j) {
            const int64_t oi = perm[i], oj = perm[j];
            if (!nz[oi > oj ? oi * nb + oj : oj * nb + oi]) continue;
            for (int64_t r = 0; r < blk; ++r)
                for (int64_t c = 0; c < blk; ++c) {
                    /* symmetrised as :903-911; entries at or below the drop threshold inside a kept block are dropped too */
                    const double v = 0.5 * (S[(oi * blk + r) * n + oj * blk + c] + S[(oj * blk + c) * n + oi * blk + r]);
                    A[(i * blk + r) * n + j * blk + c] = fabs(v) > 1e-12 ? v : 0.0;
                }
        }
    for (int64_t i = 0; i < nb; ++i)
        for (int64_t r = 0; r < blk; ++r) pb[i * blk + r] = b[perm[i] * blk + r];
    int rc = ORA_ERR_SINGULAR;
    double trace = 0.0, max_diag = 0.0;
    for (int64_t i = 0; i < n; ++i) { trace += A[i * n + i]; max_diag = fmax(max_diag, fabs(A[i * n + i])); }
    const double base = fmax(fmax(trace / (double)n, max_diag), 1.0);
    for (int attempt = -1; attempt < 5 && rc != ORA_OK; ++attempt) {
        const double reg = attempt < 0 ? 0.0 : base * pow(10.0, (double)(attempt - 4));
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            const int64_t f = firstb[i / blk] * blk;
            memcpy(L + i * n + f, A + i * n + f, (size_t)(i - f + 1) * 8);
            L[i * n + i] += reg;
        }
        if (envelope_llt(n, L, last) != 0) continue;
        double *y = (double *)malloc((size_t)n * 8);
        for (int64_t i = 0; i < n; ++i) {
            double s = pb[i];
            for (int64_t p = firstb[i / blk] * blk; p < i; ++p) s -= L[i * n + p] * y[p];
            y[i] = s / L[i * n + i];
        }
        for (int64_t i = n - 1; i >= 0; --i) {
            y[i] /= L[i * n + i];
            for (int64_t p = firstb[i / blk] * blk; p < i; ++p) y[p] -= L[i * n + p] * y[i];
        }
        for (int64_t i = 0; i < nb; ++i)
            for (int64_t r = 0; r < blk; ++r) x[perm[i] * blk + r] = y[i * blk + r];
        free(y);
        if (reg_used) *reg_used = reg;
        rc = ORA_OK;
    }
    free(A); free(L); free(pb); free(nz); free(ptr); free(adj); free(fill); free(perm); free(inv); free(firstb); free(last);
    return rc;
}

/* solve_with_pcg (explicit_schur.rs:639-756): Jacobi-preconditioned CG on the
 * explicit S.  Returns the number of iterations performed in *iters. */
int ora_solve_pcg(int64_t n, const double *S, const double *b, int64_t max_iter, double tol,
                  double *x, int64_t *iters) {
    double *pre = (double *)malloc((size_t)n * sizeof(double));
    double *r = (double *)malloc((size_t)n * sizeof(double));
    double *z = (double *)malloc((size_t)n * sizeof(double));
    double *p = (double *)malloc((size_t)n * sizeof(double));
    double *ap = (double *)malloc((size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; ++i) {
        double d = S[i * n + i];
        /* a dropped (|d|<=1e-12) diagonal is structurally absent: precond stays 1.0 */
        pre[i] = (fabs(d) > 1e-12) ? 1.0 / d : 1.0;
        x[i] = 0.0; r[i] = b[i]; z[i] = pre[i] * r[i]; p[i] = z[i];
    }
    double rz_old = 0.0, rn0 = 0.0;
    for (int64_t i = 0; i < n; ++i) { rz_old += r[i] * z[i]; rn0 += r[i] * r[i]; }
    rn0 = sqrt(rn0);
    double abs_tol = tol * fmax(rn0, 1.0);
    int64_t it = 0;
    for (; it < max_iter; ++it) {
        /* column sweep A*p as the reference does (symmetric S: same as row dot) */
        for (int64_t i = 0; i < n; ++i) ap[i] = 0.0;
        for (int64_t c = 0; c < n; ++c) {
            double pc = p[c];
            const double *col = S + c * n; /* S symmetric: column c == row c */
            for (int64_t rr = 0; rr < n; ++rr) ap[rr] += col[rr] * pc;
        }
        double pap = 0.0;
        for (int64_t i = 0; i < n; ++i) pap += p[i] * ap[i];
        if (fabs(pap) < 1e-30) break;
        double alpha = rz_old / pap;
        for (int64_t i = 0; i < n; ++i) x[i] += alpha * p[i];
        for (int64_t i = 0; i < n; ++i) r[i] -= alpha * ap[i];
        double rn = 0.0;
        for (int64_t i = 0; i < n; ++i) rn += r[i] * r[i];
        rn = sqrt(rn);
        if (rn < abs_tol) { ++it; break; }
        for (int64_t i = 0; i < n; ++i) z[i] = pre[i] * r[i];
        double rz_new = 0.0;
        for (int64_t i = 0; i < n; ++i) rz_new += r[i] * z[i];
        if (fabs(rz_old) < 1e-30) { ++it; break; }
        double beta = rz_new / rz_old;
        for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
        rz_old = rz_new;
    }
    if (iters) *iters = it;
    free(pre); free(r); free(z); free(p); free(ap);
    return ORA_OK;
}

/* ------------------------------------------------------------------------- */
/* A18: IterativeSchurSolver (src/linalg/sparse/implicit_schur.rs): the reduced  */
/* system is never formed; S x = H_cc x - H_cp (H_pp^-1 (H_cp^T x))             */
/* (apply_schur_operator_fast :163-251), Schur-Jacobi preconditioner = inverse  */
/* of the diagonal blocks of S per camera-side VARIABLE (:456-573: pose 6x6 and */
/* intrinsics 3x3 blocks), PCG with tol*max(|b|,1) (:577-679).                  */
/* ------------------------------------------------------------------------- */
/* general inverse by Gauss-Jordan with partial pivoting; 0 when a pivot is exactly zero
 * (DMatrix::try_inverse -> None) */
static int dense_try_inverse(int n, const double *A, double *inv) {
    double M[9 * 18];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { M[i * 2 * n + j] = A[i * n + j]; M[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (fabs(M[r * 2 * n + c]) > fabs(M[piv * 2 * n + c])) piv = r;
        if (M[piv * 2 * n + c] == 0.0) return 0;
        if (piv != c)
            for (int j = 0; j < 2 * n; ++j) { double t = M[c * 2 * n + j]; M[c * 2 * n + j] = M[piv * 2 * n + j]; M[piv * 2 * n + j] = t; }
        double d = M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) M[c * 2 * n + j] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = M[r * 2 * n + c];
            if (f != 0.0)
                for (int j = 0; j < 2 * n; ++j) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) inv[i * n + j] = M[i * 2 * n + n + j];
    return 1;
}

/* y = S x without S (implicit_schur.rs:163-251).  Hcc dense nc x nc (damped); H_cp as the per-landmark
 * merged row lists used above; Hinv = damped, gated 3x3 inverses. */
static void implicit_apply(int64_t nc, const double *Hcc, int64_t npt, const int64_t *row_ptr, const int64_t *cam_rows,
                           const double *hcl, const double *Hinv, const double *x, double *y, double *tmp_lm) {
    for (int64_t i = 0; i < nc; ++i) {
        double s = 0.0;
        const double *row = Hcc + i * nc;
        for (int64_t j = 0; j < nc; ++j) s += row[j] * x[j];
        y[i] = s;
    }
    for (int64_t b = 0; b < npt; ++b) {
        double t[3] = {0, 0, 0};
        for (int64_t k = row_ptr[b]; k < row_ptr[b + 1]; ++k)
            for (int c = 0; c < 3; ++c) t[c] += hcl[3 * k + c] * x[cam_rows[k]];
        const double *Hi = Hinv + 9 * b;
        for (int a = 0; a < 3; ++a) tmp_lm[3 * b + a] = Hi[3 * a] * t[0] + Hi[3 * a + 1] * t[1] + Hi[3 * a + 2] * t[2];
    }
    for (int64_t b = 0; b < npt; ++b)
        for (int64_t k = row_ptr[b]; k < row_ptr[b + 1]; ++k)
            y[cam_rows[k]] -= hcl[3 * k] * tmp_lm[3 * b] + hcl[3 * k + 1] * tmp_lm[3 * b + 1] + hcl[3 * k + 2] * tmp_lm[3 * b + 2];
}

/* blocks: n_blocks x (start column, size <= 9).  Returns iterations in *iters. */
int ora_solve_implicit_pcg(int64_t nc, const double *Hcc, int64_t npt, const int64_t *row_ptr, const int64_t *cam_rows,
                           const double *hcl, const double *Hinv, const double *b, int64_t n_blocks,
                           const int64_t *blk_start, const int64_t *blk_size, int64_t max_iter, double tol, double *x,
                           int64_t *iters) {
    /* Schur-Jacobi blocks: S_ii = H_cc[ii] - sum_l H_cp[i,l] H_pp^-1 H_cp[i,l]^T, inverted (:456-573) */
    double *Minv = (double *)calloc((size_t)n_blocks * 81, 8);
    double *Sii = (double *)calloc((size_t)n_blocks * 81, 8);
    int64_t *blk_of_row = (int64_t *)malloc((size_t)nc * 8);
    for (int64_t i = 0; i < nc; ++i) blk_of_row[i] = -1;
    for (int64_t q = 0; q < n_blocks; ++q) {
        const int64_t s0 = blk_start[q], n = blk_size[q];
        for (int64_t a = 0; a < n; ++a) {
            blk_of_row[s0 + a] = q;
            for (int64_t c = 0; c < n; ++c) Sii[81 * q + a * n + c] = Hcc[(s0 + a) * nc + (s0 + c)];
        }
    }
    for (int64_t l = 0; l < npt; ++l) {
        const double *Hi = Hinv + 9 * l;
        for (int64_t k1 = row_ptr[l]; k1 < row_ptr[l + 1]; ++k1) {
            const int64_t q = blk_of_row[cam_rows[k1]];
            const int64_t s0 = blk_start[q], n = blk_size[q];
            double t[3];
            for (int c = 0; c < 3; ++c) t[c] = hcl[3 * k1] * Hi[c] + hcl[3 * k1 + 1] * Hi[3 + c] + hcl[3 * k1 + 2] * Hi[6 + c];
            for (int64_t k2 = row_ptr[l]; k2 < row_ptr[l + 1]; ++k2) {
                if (blk_of_row[cam_rows[k2]] != q) continue;
                Sii[81 * q + (cam_rows[k1] - s0) * n + (cam_rows[k2] - s0)] -=
                    t[0] * hcl[3 * k2] + t[1] * hcl[3 * k2 + 1] + t[2] * hcl[3 * k2 + 2];
            }
        }
    }
    for (int64_t q = 0; q < n_blocks; ++q) {
        const int n = (int)blk_size[q];
        double *A = Sii + 81 * q, *I = Minv + 81 * q;
        if (!dense_try_inverse(n, A, I)) {
            double tr = 0.0;
            for (int a = 0; a < n; ++a) tr += A[a * n + a];
            double reg = fmax(1e-6 * fabs(tr) / (double)n, 1e-8);
            for (int a = 0; a < n; ++a) A[a * n + a] += reg;
            if (!dense_try_inverse(n, A, I))
                for (int a = 0; a < n; ++a)
                    for (int c = 0; c < n; ++c) I[a * n + c] = (a == c) ? 1.0 : 0.0;
        }
    }
#define APPLY_PRECOND(src, dst)                                                         \
    for (int64_t q = 0; q < n_blocks; ++q) {                                            \
        const int64_t s0 = blk_start[q], n = blk_size[q];                               \
        for (int64_t a = 0; a < n; ++a) {                                               \
            double s = 0.0;                                                             \
            for (int64_t c = 0; c < n; ++c) s += Minv[81 * q + a * n + c] * (src)[s0 + c]; \
            (dst)[s0 + a] = s;                                                          \
        }                                                                               \
    }
    double *r = (double *)malloc((size_t)nc * 8), *z = (double *)calloc((size_t)nc, 8), *pp = (double *)malloc((size_t)nc * 8);
    double *ap = (double *)malloc((size_t)nc * 8), *tmp = (double *)malloc((size_t)npt * 3 * 8 + 8);
    double bn = 0.0, rz_old = 0.0;
    for (int64_t i = 0; i < nc; ++i) { x[i] = 0.0; r[i] = b[i]; bn += b[i] * b[i]; }
    APPLY_PRECOND(r, z)
    for (int64_t i = 0; i < nc; ++i) { pp[i] = z[i]; rz_old += r[i] * z[i]; }
    const double abs_tol = tol * fmax(sqrt(bn), 1.0);
    int64_t it = 0;
    for (; it < max_iter; ++it) {
        implicit_apply(nc, Hcc, npt, row_ptr, cam_rows, hcl, Hinv, pp, ap, tmp);
        double pap = 0.0;
        for (int64_t i = 0; i < nc; ++i) pap += pp[i] * ap[i];
        if (fabs(pap) < 1e-20) break;
        const double alpha = rz_old / pap;
        double rn = 0.0;
        for (int64_t i = 0; i < nc; ++i) { x[i] += alpha * pp[i]; r[i] -= alpha * ap[i]; rn += r[i] * r[i]; }
        if (sqrt(rn) < abs_tol) { ++it; break; }
        APPLY_PRECOND(r, z)
        double rz_new = 0.0;
        for (int64_t i = 0; i < nc; ++i) rz_new += r[i] * z[i];
        if (fabs(rz_old) < 1e-30) { ++it; break; }
        const double beta = rz_new / rz_old;
        for (int64_t i = 0; i < nc; ++i) pp[i] = z[i] + beta * pp[i];
        rz_old = rz_new;
    }
#undef APPLY_PRECOND
    if (iters) *iters = it;
    free(Minv); free(Sii); free(blk_of_row); free(r); free(z); free(pp); free(ap); free(tmp);
    return ORA_OK;
}

/* A6-A13 on an arbitrary dense Jacobian whose columns are [cam_dof camera columns |
 * 3*n_pt landmark columns] -- the shape of the reference's unit-test fixture
 * create_schur_test_setup (explicit_schur.rs:1304-1363).  Same sequence as
 * solve_augmented_equation (:1129-1234); lambda = 0 gives solve_normal_equation. */
int ora_schur_solve_dense_jacobian(int64_t n_rows, int64_t cam_dof, int64_t n_pt, const double *J,
                                   const double *r, double lambda, int variant, int cg_max_iter,
                                   double cg_tol, double *step_out, double *grad_out) {
    int64_t n = cam_dof + 3 * n_pt;
    double *H = (double *)calloc((size_t)n * (size_t)n, 8);
    double *g = (double *)calloc((size_t)n, 8);
    for (int64_t k = 0; k < n_rows; ++k)
        for (int64_t i = 0; i < n; ++i) {
            double ji = J[k * n + i];
            if (ji == 0.0) continue;
            g[i] += ji * r[k];
            for (int64_t j = 0; j < n; ++j) H[i * n + j] += ji * J[k * n + j];
        }
    if (grad_out) memcpy(grad_out, g, (size_t)n * 8);
    double *Hcc = (double *)malloc((size_t)cam_dof * (size_t)cam_dof * 8);
    double *Hll = (double *)malloc((size_t)n_pt * 9 * 8);
    for (int64_t i = 0; i < cam_dof; ++i)
        for (int64_t j = 0; j < cam_dof; ++j) Hcc[i * cam_dof + j] = H[i * n + j] + (i == j ? lambda : 0.0);
    for (int64_t l = 0; l < n_pt; ++l)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                Hll[9 * l + 3 * a + b] = H[(cam_dof + 3 * l + a) * n + cam_dof + 3 * l + b] + (a == b ? lambda : 0.0);
    /* merged row lists: structurally non-zero rows of the three landmark columns */
    int64_t *row_ptr = (int64_t *)calloc((size_t)n_pt + 1, 8);
    int64_t *cam_rows = (int64_t *)malloc((size_t)n_pt * (size_t)cam_dof * 8 + 8);
    double *hcl = (double *)malloc((size_t)n_pt * (size_t)cam_dof * 3 * 8 + 8);
    int64_t w = 0;
    for (int64_t l = 0; l < n_pt; ++l) {
        for (int64_t i = 0; i < cam_dof; ++i) {
            const double *h = H + i * n + cam_dof + 3 * l;
            if (h[0] != 0.0 || h[1] != 0.0 || h[2] != 0.0) {
                cam_rows[w] = i; hcl[3 * w] = h[0]; hcl[3 * w + 1] = h[1]; hcl[3 * w + 2] = h[2]; ++w;
            }
        }
        row_ptr[l + 1] = w;
    }
    double *g_c = (double *)malloc((size_t)cam_dof * 8), *g_p = (double *)malloc((size_t)n_pt * 24);
    for (int64_t i = 0; i < cam_dof; ++i) g_c[i] = -g[i];
    for (int64_t i = 0; i < 3 * n_pt; ++i) g_p[i] = -g[cam_dof + i];
    double *Hinv = (double *)malloc((size_t)n_pt * 72), *S = (double *)malloc((size_t)cam_dof * (size_t)cam_dof * 8);
    double *gred = (double *)malloc((size_t)cam_dof * 8);
    int rc = ora_invert_landmark_blocks(n_pt, Hll, 0.0, Hinv);
    if (rc == ORA_OK) {
        ora_schur_complement(cam_dof, Hcc, n_pt, row_ptr, cam_rows, hcl, Hinv, S);
        ora_reduced_gradient(cam_dof, g_c, n_pt, g_p, row_ptr, cam_rows, hcl, Hinv, gred);
        int64_t its;
        if (variant == 2) { /* IterativeSchurSolver on the fixture: camera variables are 6-DOF blocks */
            int64_t nb = cam_dof / 6, *bs = (int64_t *)malloc((size_t)nb * 8), *bz = (int64_t *)malloc((size_t)nb * 8);
            for (int64_t q = 0; q < nb; ++q) { bs[q] = 6 * q; bz[q] = 6; }
            rc = ora_solve_implicit_pcg(cam_dof, Hcc, n_pt, row_ptr, cam_rows, hcl, Hinv, gred, nb, bs, bz, cg_max_iter, cg_tol,
                                        step_out, &its);
            free(bs); free(bz);
        } else if (variant == 1) rc = ora_solve_pcg(cam_dof, S, gred, cg_max_iter, cg_tol, step_out, &its);
        else if (variant == 3) rc = ora_solve_cholesky_sparse(cam_dof, 1, S, gred, step_out, NULL, NULL);
        else rc = ora_solve_cholesky(cam_dof, S, gred, step_out, NULL);
    }
    if (rc == ORA_OK) ora_back_substitute(n_pt, step_out, g_p, row_ptr, cam_rows, hcl, Hinv, step_out + cam_dof);
    free(H); free(g); free(Hcc); free(Hll); free(row_ptr); free(cam_rows); free(hcl); free(g_c); free(g_p);
    free(Hinv); free(S); free(gred);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* The bundle-adjustment problem (bin/bundle_adjustment.rs:232-298, 391-441)   */
/* ------------------------------------------------------------------------- */

typedef struct {
    int64_t n_cam, n_pt, n_obs;
    int mode; /* 0: BundleAdjustment keys [pose,pt]; 1: SelfCalibration keys [pose,pt,intr]; 2..6: OnlyPose, OnlyLandmarks,
               * OnlyIntrinsics, PoseAndIntrinsics, LandmarksAndIntrinsics (src/factors/mod.rs:82-101) */
    uint32_t *cam_idx, *pt_idx;
    double *obs_uv;
    int64_t *intr_col, *pose_col, *pt_col; /* reference global columns */
    int64_t cam_dof, total_dof;
    double huber_delta;
    uint8_t *fix_pose, *fix_intr, *fix_pt; /* per-DOF fixed masks (problem.rs:185-197) */
    double *poses, *intr, *points;         /* current values as VariableEnum::to_vector */
    /* landmark-major observation lists */
    int64_t *pt_ptr, *pt_obs;
    /* last linearisation */
    double *r, *Jpose, *Jpt, *Jintr;
    double *grad; /* +J^T r, global column order (get_gradient, :1240-1242) */
    int cg_max_iter; double cg_tol; int64_t last_pcg_iters; double last_reg;
    double *scaling; /* Jacobi column scaling (optimizer/mod.rs:749-763), total_dof; NULL = off */
    double sparse_stats[3]; /* of the last variant-3 solve: blocks of the sparsified S, envelope entries, half bandwidth */
} ora_problem;

void ora_problem_destroy(ora_problem *p) {
    if (!p) return;
    free(p->cam_idx); free(p->pt_idx); free(p->obs_uv); free(p->intr_col); free(p->pose_col);
    free(p->pt_col); free(p->fix_pose); free(p->fix_intr); free(p->fix_pt); free(p->poses);
    free(p->intr); free(p->points); free(p->pt_ptr); free(p->pt_obs); free(p->r); free(p->Jpose);
    free(p->Jpt); free(p->Jintr); free(p->grad); free(p->scaling);
    free(p);
}

static void *dupmem(const void *src, size_t bytes) {
    void *d = malloc(bytes ? bytes : 1);
    if (src) memcpy(d, src, bytes); else memset(d, 0, bytes);
    return d;
}

ora_problem *ora_problem_create(int64_t n_cam, int64_t n_pt, int64_t n_obs, int mode,
                                const uint32_t *cam_idx, const uint32_t *pt_idx, const double *obs_uv,
                                const int64_t *intr_col, const int64_t *pose_col, const int64_t *pt_col,
                                double huber_delta, const uint8_t *fix_pose, const uint8_t *fix_intr,
                                const uint8_t *fix_pt) {
    ora_problem *p = (ora_problem *)calloc(1, sizeof *p);
    p->n_cam = n_cam; p->n_pt = n_pt; p->n_obs = n_obs; p->mode = mode;
    p->cam_idx = (uint32_t *)dupmem(cam_idx, (size_t)n_obs * 4);
    p->pt_idx = (uint32_t *)dupmem(pt_idx, (size_t)n_obs * 4);
    p->obs_uv = (double *)dupmem(obs_uv, (size_t)n_obs * 16);
    p->intr_col = (int64_t *)dupmem(intr_col, (size_t)n_cam * 8);
    p->pose_col = (int64_t *)dupmem(pose_col, (size_t)n_cam * 8);
    p->pt_col = (int64_t *)dupmem(pt_col, (size_t)n_pt * 8);
    p->cam_dof = 9 * n_cam; p->total_dof = 9 * n_cam + 3 * n_pt;
    p->huber_delta = huber_delta;
    p->fix_pose = (uint8_t *)dupmem(fix_pose, (size_t)n_cam * 6);
    p->fix_intr = (uint8_t *)dupmem(fix_intr, (size_t)n_cam * 3);
    p->fix_pt = (uint8_t *)dupmem(fix_pt, (size_t)n_pt * 3);
    p->poses = (double *)calloc((size_t)n_cam * 7, 8);
    p->intr = (double *)calloc((size_t)n_cam * 3, 8);
    p->points = (double *)calloc((size_t)n_pt * 3, 8);
    p->pt_ptr = (int64_t *)calloc((size_t)n_pt + 1, 8);
    p->pt_obs = (int64_t *)calloc((size_t)n_obs + 1, 8);
    for (int64_t i = 0; i < n_obs; ++i) p->pt_ptr[pt_idx[i] + 1]++;
    for (int64_t l = 0; l < n_pt; ++l) p->pt_ptr[l + 1] += p->pt_ptr[l];
    int64_t *fill = (int64_t *)dupmem(p->pt_ptr, (size_t)n_pt * 8);
    for (int64_t i = 0; i < n_obs; ++i) p->pt_obs[fill[pt_idx[i]]++] = i;
    free(fill);
    p->r = (double *)calloc((size_t)n_obs * 2, 8);
    p->Jpose = (double *)calloc((size_t)n_obs * 12, 8);
    p->Jpt = (double *)calloc((size_t)n_obs * 6, 8);
    p->Jintr = (double *)calloc((size_t)n_obs * 6, 8);
    p->grad = (double *)calloc((size_t)p->total_dof, 8);
    p->cg_max_iter = 200; p->cg_tol = 1e-6; /* SparseSchurComplementSolver::new, :211-212 */
    return p;
}

void ora_set_params(ora_problem *p, const double *poses, const double *intr, const double *points) {
    memcpy(p->poses, poses, (size_t)p->n_cam * 7 * 8);
    memcpy(p->intr, intr, (size_t)p->n_cam * 3 * 8);
    memcpy(p->points, points, (size_t)p->n_pt * 3 * 8);
}
void ora_get_params(const ora_problem *p, double *poses, double *intr, double *points) {
    memcpy(poses, p->poses, (size_t)p->n_cam * 7 * 8);
    memcpy(intr, p->intr, (size_t)p->n_cam * 3 * 8);
    memcpy(points, p->points, (size_t)p->n_pt * 3 * 8);
}
void ora_set_cg_params(ora_problem *p, int max_iter, double tol) { p->cg_max_iter = max_iter; p->cg_tol = tol; }

/* compute_cost (optimizer/mod.rs:358-361): 0.5 * norm_l2(r)^2 */
double ora_compute_cost(int64_t n, const double *r) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += r[i] * r[i];
    double nr = sqrt(s);
    return 0.5 * nr * nr;
}

/* A16: Problem::compute_residual_sparse (problem.rs:864-899, 985-1024). r_out may be NULL. */
double ora_residuals(ora_problem *p, double *r_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < p->n_obs; ++i) {
        uint32_t c = p->cam_idx[i], l = p->pt_idx[i];
        ora_linearize_obs(p->poses + 7 * c, p->intr + 3 * c, p->points + 3 * l, p->obs_uv + 2 * i,
                          p->huber_delta, 0, p->r + 2 * i, NULL, NULL, NULL);
    }
    if (r_out) memcpy(r_out, p->r, (size_t)p->n_obs * 16);
    return ora_compute_cost(2 * p->n_obs, p->r);
}

/* OptimizeParams<POSE, LANDMARK, INTRINSIC> of a mode as 4 POSE + 2 LANDMARK + INTRINSIC (src/factors/mod.rs:66-101):
 * evaluate_internal (projection_factor.rs:184-296) allocates Jacobian columns for the optimised blocks only; the blocks
 * that are not optimised are constants of the factor (fixed_pose / fixed_landmarks / its camera model).  The variable set
 * is the bin's (every pose_*, intr_*, pt_* exists, bundle_adjustment.rs:232-257), so a block without columns is a block
 * of zero columns in the global Jacobian. */
static int ora_mode_mask(int mode) {
    static const int m[7] = {6, 7, 4, 2, 1, 5, 3};
    return (mode >= 0 && mode < 7) ? m[mode] : 7;
}

/* A5: assemble (linearizer/cpu/sparse.rs:119-184): residual + Jacobian blocks.
 * In mode 0 the factor has no intrinsics key; Jintr is kept (for inspection) but is not
 * part of the Jacobian. Outputs may be NULL. */
double ora_linearize(ora_problem *p, double *r_out, double *Jpose_out, double *Jpt_out, double *Jintr_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < p->n_obs; ++i) {
        uint32_t c = p->cam_idx[i], l = p->pt_idx[i];
        ora_linearize_obs(p->poses + 7 * c, p->intr + 3 * c, p->points + 3 * l, p->obs_uv + 2 * i,
                          p->huber_delta, 1, p->r + 2 * i, p->Jpose + 12 * i, p->Jpt + 6 * i,
                          p->Jintr + 6 * i);
        const int mk = ora_mode_mask(p->mode);
        if (!(mk & 4)) memset(p->Jpose + 12 * i, 0, 96);
        if (!(mk & 2)) memset(p->Jpt + 6 * i, 0, 48);
        if (!(mk & 1) && p->mode >= 2) memset(p->Jintr + 6 * i, 0, 48);   /* (mode 0 keeps it for inspection, as before) */
    }
    if (r_out) memcpy(r_out, p->r, (size_t)p->n_obs * 16);
    if (Jpose_out) memcpy(Jpose_out, p->Jpose, (size_t)p->n_obs * 96);
    if (Jpt_out) memcpy(Jpt_out, p->Jpt, (size_t)p->n_obs * 48);
    if (Jintr_out) memcpy(Jintr_out, p->Jintr, (size_t)p->n_obs * 48);
    return ora_compute_cost(2 * p->n_obs, p->r);
}

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/* A6-A13: SparseSchurComplementSolver::solve_augmented_equation
 * (explicit_schur.rs:1129-1234) on the last linearisation.
 * variant 0 = Sparse (Cholesky; dense LL^T here), 3 = the same through ora_solve_cholesky_sparse (the timed CPU baseline),
 * 1 = Iterative (Jacobi-PCG on explicit S),
 * 2 = IterativeSchurSolver semantics (matrix-free PCG, Schur-Jacobi preconditioner; implicit_schur.rs).
 * step_out / grad_out: total_dof, reference global column order.
 * S_out (cam_dof^2, row-major, camera-side columns in reference order) and
 * gred_out (cam_dof) are optional. */
int ora_solve_augmented(ora_problem *p, double lambda, int variant, double *step_out,
                        double *grad_out, double *S_out, double *gred_out) {
    const int64_t nc = p->cam_dof, npt = p->n_pt, nobs = p->n_obs, cam0 = 0, land0 = nc;
    const int has_intr = ora_mode_mask(p->mode) & 1;
    /* With Jacobi scaling the LM loop hands the solver J * diag(scaling) (process_jacobian_generic,
     * optimizer/mod.rs:749-763; apply_column_scaling, linearizer/mod.rs:241-253): everything below then
     * runs on the scaled blocks and step_out / grad_out are the SCALED step and gradient, exactly what
     * LinearSolver::solve_augmented_equation / get_gradient return to compute_step_generic. */
    double *JPOSE = p->Jpose, *JPT = p->Jpt, *JINTR = p->Jintr;
    if (p->scaling) {
        JPOSE = (double *)malloc((size_t)nobs * 12 * 8 + 8); JPT = (double *)malloc((size_t)nobs * 6 * 8 + 8);
        JINTR = (double *)malloc((size_t)nobs * 6 * 8 + 8);
        for (int64_t i = 0; i < nobs; ++i) {
            uint32_t c = p->cam_idx[i], l = p->pt_idx[i];
            for (int rr = 0; rr < 2; ++rr) {
                for (int a = 0; a < 6; ++a) JPOSE[12 * i + 6 * rr + a] = p->Jpose[12 * i + 6 * rr + a] * p->scaling[p->pose_col[c] + a];
                for (int a = 0; a < 3; ++a) JPT[6 * i + 3 * rr + a] = p->Jpt[6 * i + 3 * rr + a] * p->scaling[p->pt_col[l] + a];
                for (int a = 0; a < 3; ++a) JINTR[6 * i + 3 * rr + a] = p->Jintr[6 * i + 3 * rr + a] * p->scaling[p->intr_col[c] + a];
            }
        }
    }
    /* H = J^T J (:1146-1150) restricted to the three block families; g = J^T r (:1151) */
    double *Hcc = (double *)calloc((size_t)nc * (size_t)nc, 8);
    double *Hll = (double *)calloc((size_t)npt * 9, 8);
    double *g = p->grad;
    memset(g, 0, (size_t)p->total_dof * 8);
    if (!Hcc || !Hll) { free(Hcc); free(Hll); return ORA_ERR_INPUT; }
    for (int64_t i = 0; i < nobs; ++i) {
        uint32_t c = p->cam_idx[i], l = p->pt_idx[i];
        const double *Jp = JPOSE + 12 * i, *Jl = JPT + 6 * i, *Ji = JINTR + 6 * i;
        const double *r = p->r + 2 * i;
        int64_t pc = p->pose_col[c], ic = p->intr_col[c], lc = p->pt_col[l];
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b)
                Hcc[(pc + a) * nc + (pc + b)] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
            g[pc + a] += Jp[a] * r[0] + Jp[6 + a] * r[1];
        }
        if (has_intr) {
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b)
                    Hcc[(ic + a) * nc + (ic + b)] += Ji[a] * Ji[b] + Ji[3 + a] * Ji[3 + b];
                for (int b = 0; b < 6; ++b) {
                    double v = Ji[a] * Jp[b] + Ji[3 + a] * Jp[6 + b];
                    Hcc[(ic + a) * nc + (pc + b)] += v;
                    Hcc[(pc + b) * nc + (ic + a)] += v;
                }
                g[ic + a] += Ji[a] * r[0] + Ji[3 + a] * r[1];
            }
        }
        double *B = Hll + 9 * ((lc - land0) / 3);
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) B[3 * a + b] += Jl[a] * Jl[b] + Jl[3 + a] * Jl[3 + b];
            g[lc + a] += Jl[a] * r[0] + Jl[3 + a] * r[1];
        }
    }
    if (grad_out) memcpy(grad_out, g, (size_t)p->total_dof * 8);

    /* H_cl as merged row lists per landmark BLOCK INDEX in reference landmark order
     * (:811-865).  Landmark block b <-> the landmark whose pt_col = land0 + 3b. */
    int64_t *blk_of_pt = (int64_t *)malloc((size_t)npt * 8);
    int64_t *pt_of_blk = (int64_t *)malloc((size_t)npt * 8);
    for (int64_t l = 0; l < npt; ++l) { blk_of_pt[l] = (p->pt_col[l] - land0) / 3; pt_of_blk[blk_of_pt[l]] = l; }
    const int rows_per_obs = has_intr ? 9 : 6;
    int64_t *row_ptr = (int64_t *)calloc((size_t)npt + 1, 8);
    /* count distinct (camera) per landmark: merge duplicates of the same camera */
    int64_t *cams_sorted = (int64_t *)malloc((size_t)nobs * 8 + 8);
    int64_t *lst_ptr = (int64_t *)calloc((size_t)npt + 1, 8);
    {
        int64_t w = 0;
        for (int64_t b = 0; b < npt; ++b) {
            int64_t l = pt_of_blk[b];
            int64_t beg = w;
            for (int64_t k = p->pt_ptr[l]; k < p->pt_ptr[l + 1]; ++k) cams_sorted[w++] = p->cam_idx[p->pt_obs[k]];
            qsort(cams_sorted + beg, (size_t)(w - beg), 8, cmp_i64);
            int64_t u = beg;
            for (int64_t k = beg; k < w; ++k)
                if (k == beg || cams_sorted[k] != cams_sorted[k - 1]) cams_sorted[u++] = cams_sorted[k];
            w = u;
            lst_ptr[b + 1] = w;
            row_ptr[b + 1] = row_ptr[b] + (w - beg) * rows_per_obs;
        }
    }
    int64_t nrows = row_ptr[npt];
    int64_t *cam_rows = (int64_t *)malloc((size_t)nrows * 8 + 8);
    double *hcl = (double *)calloc((size_t)nrows * 3 + 3, 8);
    for (int64_t b = 0; b < npt; ++b) {
        int64_t l = pt_of_blk[b];
        int64_t ncam = lst_ptr[b + 1] - lst_ptr[b];
        /* ascending global row: all intr rows (cols < 3 n_cam) first, then pose rows;
         * within a family ascending column == ascending sorted camera *rank*.  Sort rows. */
        int64_t base = row_ptr[b], w = base;
        for (int64_t k = 0; k < ncam; ++k) {
            int64_t c = cams_sorted[lst_ptr[b] + k];
            if (has_intr) for (int a = 0; a < 3; ++a) cam_rows[w++] = p->intr_col[c] + a - cam0;
            for (int a = 0; a < 6; ++a) cam_rows[w++] = p->pose_col[c] + a - cam0;
        }
        qsort(cam_rows + base, (size_t)(w - base), 8, cmp_i64);
        /* accumulate values: for every observation of l find its rows by binary search */
        for (int64_t k = p->pt_ptr[l]; k < p->pt_ptr[l + 1]; ++k) {
            int64_t i = p->pt_obs[k];
            uint32_t c = p->cam_idx[i];
            const double *Jp = JPOSE + 12 * i, *Jl = JPT + 6 * i, *Ji = JINTR + 6 * i;
            for (int a = 0; a < rows_per_obs; ++a) {
                int64_t row = (a < 6) ? p->pose_col[c] + a : p->intr_col[c] + (a - 6);
                int64_t *f = (int64_t *)bsearch(&row, cam_rows + base, (size_t)(w - base), 8, cmp_i64);
                int64_t idx = f - cam_rows;
                double j0 = (a < 6) ? Jp[a] : Ji[a - 6], j1 = (a < 6) ? Jp[6 + a] : Ji[3 + (a - 6)];
                for (int cc = 0; cc < 3; ++cc) hcl[3 * idx + cc] += j0 * Jl[cc] + j1 * Jl[3 + cc];
            }
        }
    }

    /* negative gradient blocks (:1152-1155, :1166) */
    double *g_c = (double *)malloc((size_t)nc * 8), *g_p = (double *)malloc((size_t)npt * 3 * 8);
    for (int64_t i = 0; i < nc; ++i) g_c[i] = -g[cam0 + i];
    for (int64_t i = 0; i < 3 * npt; ++i) g_p[i] = -g[land0 + i];

    /* damping: lambda*I on H_cc and on every H_ll block (:1186-1212) */
    for (int64_t i = 0; i < nc; ++i) Hcc[i * nc + i] += lambda;
    for (int64_t b = 0; b < npt; ++b) { Hll[9 * b] += lambda; Hll[9 * b + 4] += lambda; Hll[9 * b + 8] += lambda; }

    int rc = ORA_OK;
    double *Hinv = (double *)malloc((size_t)npt * 9 * 8);
    double *S = (double *)malloc((size_t)nc * (size_t)nc * 8);
    double *gred = (double *)malloc((size_t)nc * 8);
    double *dc = (double *)malloc((size_t)nc * 8), *dp = (double *)malloc((size_t)npt * 3 * 8);
    rc = ora_invert_landmark_blocks(npt, Hll, 0.0, Hinv);
    if (rc == ORA_OK) {
        ora_schur_complement(nc, Hcc, npt, row_ptr, cam_rows, hcl, Hinv, S);
        ora_reduced_gradient(nc, g_c, npt, g_p, row_ptr, cam_rows, hcl, Hinv, gred);
        if (S_out) memcpy(S_out, S, (size_t)nc * (size_t)nc * 8);
        if (gred_out) memcpy(gred_out, gred, (size_t)nc * 8);
        if (variant == 2) {
            /* camera-side variable blocks of the reference's SchurBlockStructure: one per intr_* (3) and pose_* (6) */
            int64_t nb = 2 * p->n_cam, *bs = (int64_t *)malloc((size_t)nb * 8), *bz = (int64_t *)malloc((size_t)nb * 8);
            for (int64_t c = 0; c < p->n_cam; ++c) {
                bs[2 * c] = p->pose_col[c] - cam0; bz[2 * c] = 6;
                bs[2 * c + 1] = p->intr_col[c] - cam0; bz[2 * c + 1] = 3;
            }
            rc = ora_solve_implicit_pcg(nc, Hcc, npt, row_ptr, cam_rows, hcl, Hinv, gred, nb, bs, bz, p->cg_max_iter, p->cg_tol,
                                        dc, &p->last_pcg_iters);
            free(bs); free(bz);
        } else if (variant == 1) rc = ora_solve_pcg(nc, S, gred, p->cg_max_iter, p->cg_tol, dc, &p->last_pcg_iters);
        else if (variant == 3) rc = ora_solve_cholesky_sparse(nc, 3, S, gred, dc, &p->last_reg, p->sparse_stats);   /* columns: intr_* (3 each), then pose_* (6 = 2 x 3 each) */
        else rc = ora_solve_cholesky(nc, S, gred, dc, &p->last_reg);
    }
    if (rc == ORA_OK) {
        ora_back_substitute(npt, dc, g_p, row_ptr, cam_rows, hcl, Hinv, dp);
        /* combine_updates (:1248-1279) */
        for (int64_t i = 0; i < nc; ++i) step_out[cam0 + i] = dc[i];
        for (int64_t i = 0; i < 3 * npt; ++i) step_out[land0 + i] = dp[i];
    }
    free(Hcc); free(Hll); free(blk_of_pt); free(pt_of_blk); free(row_ptr); free(cams_sorted);
    free(lst_ptr); free(cam_rows); free(hcl); free(g_c); free(g_p); free(Hinv); free(S); free(gred);
    free(dc); free(dp);
    if (p->scaling) { free(JPOSE); free(JPT); free(JINTR); }
    return rc;
}

/* AssemblyBackend::compute_column_norms (linearizer/mod.rs:229-239) of the last linearisation's corrected
 * Jacobian, global column order.  Columns no factor touches (intr_* in BundleAdjustment mode) have norm 0. */
int ora_column_norms(const ora_problem *p, double *norms_out) {
    memset(norms_out, 0, (size_t)p->total_dof * 8);
    for (int64_t i = 0; i < p->n_obs; ++i) {
        uint32_t c = p->cam_idx[i], l = p->pt_idx[i];
        for (int rr = 0; rr < 2; ++rr) {
            for (int a = 0; a < 6; ++a) { double v = p->Jpose[12 * i + 6 * rr + a]; norms_out[p->pose_col[c] + a] += v * v; }
            for (int a = 0; a < 3; ++a) { double v = p->Jpt[6 * i + 3 * rr + a]; norms_out[p->pt_col[l] + a] += v * v; }
            if (ora_mode_mask(p->mode) & 1)
                for (int a = 0; a < 3; ++a) { double v = p->Jintr[6 * i + 3 * rr + a]; norms_out[p->intr_col[c] + a] += v * v; }
        }
    }
    for (int64_t j = 0; j < p->total_dof; ++j) norms_out[j] = sqrt(norms_out[j]);
    return ORA_OK;
}

/* The scaling vector process_jacobian_generic keeps from iteration 0 (optimizer/mod.rs:754-758);
 * NULL switches scaling off. */
int ora_set_column_scaling(ora_problem *p, const double *scaling) {
    free(p->scaling); p->scaling = NULL;
    if (scaling) p->scaling = (double *)dupmem(scaling, (size_t)p->total_dof * 8);
    return ORA_OK;
}

int64_t ora_last_pcg_iters(const ora_problem *p) { return p->last_pcg_iters; }
double ora_last_reg(const ora_problem *p) { return p->last_reg; }
void ora_last_sparse_stats(const ora_problem *p, double out[3]) { for (int k = 0; k < 3; ++k) out[k] = p->sparse_stats[k]; }

/* A15: apply_parameter_step (optimizer/mod.rs:309-331) with sign = +1, or
 * apply_negative_parameter_step (:343-356) with sign = -1.  Fixed DOF are zeroed in
 * the step first (problem.rs:185-197, 275-284).  Returns step.norm_l2() of the
 * UNMASKED step, as the reference does. */
double ora_apply_step(ora_problem *p, const double *step, double sign) {
    for (int64_t c = 0; c < p->n_cam; ++c) {
        double d[6], out[7];
        for (int a = 0; a < 6; ++a) {
            d[a] = sign * step[p->pose_col[c] + a];
            if (p->fix_pose[6 * c + a]) d[a] = 0.0;
        }
        ora_se3_plus(p->poses + 7 * c, d, out);
        memcpy(p->poses + 7 * c, out, sizeof out);
        for (int a = 0; a < 3; ++a) {
            double di = sign * step[p->intr_col[c] + a];
            if (p->fix_intr[3 * c + a]) di = 0.0;
            p->intr[3 * c + a] += di;
        }
    }
    for (int64_t l = 0; l < p->n_pt; ++l)
        for (int a = 0; a < 3; ++a) {
            double dl = sign * step[p->pt_col[l] + a];
            if (p->fix_pt[3 * l + a]) dl = 0.0;
            p->points[3 * l + a] += dl;
        }
    double s = 0.0;
    for (int64_t i = 0; i < p->total_dof; ++i) s += step[i] * step[i];
    return sqrt(s);
}

/* compute_parameter_norm (optimizer/mod.rs:458-467) */
double ora_parameter_norm(const ora_problem *p) {
    double s = 0.0;
    for (int64_t i = 0; i < 7 * p->n_cam; ++i) s += p->poses[i] * p->poses[i];
    for (int64_t i = 0; i < 3 * p->n_cam; ++i) s += p->intr[i] * p->intr[i];
    for (int64_t i = 0; i < 3 * p->n_pt; ++i) s += p->points[i] * p->points[i];
    return sqrt(s);
}

/* ------------------------------------------------------------------------- */
/* A14: the LM loop (levenberg_marquardt.rs:823-1031, 702-817)                 */
/* ------------------------------------------------------------------------- */
/* status codes = index into OptimizationStatus (optimizer/mod.rs:189-216) */
enum {
    ORA_ST_CONVERGED = 0, ORA_ST_MAX_ITER = 1, ORA_ST_COST_TOL = 2, ORA_ST_PARAM_TOL = 3,
    ORA_ST_GRAD_TOL = 4, ORA_ST_NUMERICAL_FAILURE = 5, ORA_ST_TIMEOUT = 7, ORA_ST_TR_TOO_SMALL = 8,
    ORA_ST_MIN_COST = 9, ORA_ST_INVALID_NUMERICAL = 11, ORA_ST_LINEAR_SOLVE_FAILED = 100
};

typedef struct {
    int max_iterations;         /* 50 default, 20 for_bundle_adjustment (:323, :524) */
    double cost_tolerance;      /* 1e-6 */
    double parameter_tolerance; /* 1e-8 */
    double gradient_tolerance;  /* 1e-10 */
    double damping;             /* 1e-3 */
    double damping_min;         /* 1e-12 */
    double damping_max;         /* 1e12 */
    double damping_nu;          /* 2.0 */
    double trust_region_radius;     /* 1e4 (constant; :339) */
    double min_trust_region_radius; /* 1e-32 */
    double min_cost_threshold;      /* <0: None */
    int variant;                    /* 0 Sparse(Cholesky) 1 Iterative(PCG) */
    int use_jacobi_scaling;         /* false by default (:352) */
} ora_lm_config;

/* update_damping (levenberg_marquardt.rs:702-717).  Returns 1 if the step is accepted. */
int ora_update_damping(double rho, double *damping, double *nu, double damping_min, double damping_max) {
    if (rho > 0.0) {
        double coff = 2.0 * rho - 1.0;
        *damping *= fmax(1.0 / 3.0, 1.0 - coff * coff * coff);
        *damping = fmax(*damping, damping_min);
        *nu = 2.0;
        return 1;
    }
    *damping *= *nu;
    *nu *= 2.0;
    *damping = fmin(*damping, damping_max);
    return 0;
}

/* compute_step_quality (optimizer/mod.rs:668-675) */
double ora_step_quality(double current_cost, double new_cost, double predicted) {
    double actual = current_cost - new_cost;
    if (fabs(predicted) < 1e-15) return actual > 0.0 ? 1.0 : 0.0;
    return actual / predicted;
}

/* per-iteration history rows: [cost_after, damping_after, rho, accepted, grad_norm,
 * step_norm, predicted_reduction, new_cost_trial] */
#define ORA_HIST_COLS 8

int ora_lm_optimize(ora_problem *p, ora_lm_config *cfg, double *hist, int hist_rows,
                    int *iterations_out, double *initial_cost_out, double *final_cost_out,
                    double *steps_out /* optional: hist_rows x total_dof */) {
    double lambda = cfg->damping, nu = cfg->damping_nu;
    double cost = ora_residuals(p, NULL); /* initialize_optimization_state (mod.rs:550-552) */
    if (initial_cost_out) *initial_cost_out = cost;
    double *step = (double *)malloc((size_t)p->total_dof * 8);
    double *grad = (double *)malloc((size_t)p->total_dof * 8);
    int iteration = 0, status = ORA_ST_MAX_ITER;
    for (;;) {
        ora_linearize(p, NULL, NULL, NULL, NULL);
        if (cfg->use_jacobi_scaling && iteration == 0) { /* process_jacobian_generic (mod.rs:749-763) */
            ora_column_norms(p, step);
            for (int64_t i = 0; i < p->total_dof; ++i) step[i] = 1.0 / (1.0 + step[i]);
            ora_set_column_scaling(p, step);
        }
        int rc = ora_solve_augmented(p, lambda, cfg->variant, step, grad, NULL, NULL);
        if (rc != ORA_OK) { status = ORA_ST_LINEAR_SOLVE_FAILED; break; }
        if (p->scaling) /* apply_inverse_scaling (:749-757); the gradient stays the scaled one */
            for (int64_t i = 0; i < p->total_dof; ++i) step[i] *= p->scaling[i];
        double gn = 0.0, sn = 0.0, pred = 0.0;
        for (int64_t i = 0; i < p->total_dof; ++i) {
            gn += grad[i] * grad[i];
            sn += step[i] * step[i];
            pred += step[i] * (lambda * step[i] - grad[i]); /* compute_predicted_reduction :721-727 */
        }
        gn = sqrt(gn); sn = sqrt(sn); pred *= 0.5;
        if (steps_out && iteration < hist_rows)
            memcpy(steps_out + (size_t)iteration * (size_t)p->total_dof, step, (size_t)p->total_dof * 8);
        /* evaluate_and_apply_step (:770-817) */
        ora_apply_step(p, step, 1.0);
        double new_cost = ora_residuals(p, NULL);
        double rho = ora_step_quality(cost, new_cost, pred);
        double cost_reduction = 0.0;
        int accepted = ora_update_damping(rho, &lambda, &nu, cfg->damping_min, cfg->damping_max);
        if (accepted) {
            cost_reduction = cost - new_cost; cost = new_cost;
        } else {
            ora_apply_step(p, step, -1.0);
        }
        if (hist && iteration < hist_rows) {
            double *h = hist + (size_t)iteration * ORA_HIST_COLS;
            h[0] = cost; h[1] = lambda; h[2] = rho; h[3] = accepted; h[4] = gn; h[5] = sn; h[6] = pred; h[7] = new_cost;
        }
        /* check_convergence (mod.rs:591-658) */
        double pnorm = ora_parameter_norm(p);
        double cost_before = accepted ? cost + cost_reduction : cost;
        int st = -1;
        if (!isfinite(cost) || !isfinite(sn) || !isfinite(gn)) st = ORA_ST_INVALID_NUMERICAL;
        else if (iteration >= cfg->max_iterations) st = ORA_ST_MAX_ITER;
        else if (accepted) {
            if (gn < cfg->gradient_tolerance) st = ORA_ST_GRAD_TOL;
            if (st < 0 && iteration > 0) {
                double rel_step_tol = cfg->parameter_tolerance * (pnorm + cfg->parameter_tolerance);
                if (sn <= rel_step_tol) st = ORA_ST_PARAM_TOL;
                else {
                    double cc = fabs(cost_before - cost);
                    if (cc / fmax(cost_before, 1e-10) < cfg->cost_tolerance) st = ORA_ST_COST_TOL;
                }
            }
            if (st < 0 && cfg->min_cost_threshold >= 0.0 && cost < cfg->min_cost_threshold) st = ORA_ST_MIN_COST;
            if (st < 0 && cfg->trust_region_radius < cfg->min_trust_region_radius) st = ORA_ST_TR_TOO_SMALL;
        }
        if (st >= 0) { status = st; ++iteration; break; }
        ++iteration;
    }
    cfg->damping = lambda; cfg->damping_nu = nu;
    if (iterations_out) *iterations_out = iteration;
    if (final_cost_out) *final_cost_out = cost;
    if (cfg->use_jacobi_scaling) ora_set_column_scaling(p, NULL); /* the scaling lives in the optimizer, not the problem */
    free(step); free(grad);
    return status;
}

/* ------------------------------------------------------------------------- */
/* REFEREE: the exact step of the last linearisation, in __float128            */
/* ------------------------------------------------------------------------- */
/* NOT a restatement of reference code.  The reference solves S dc = g_red in fp64
 * (solve_with_cholesky, explicit_schur.rs:539-634); on bundle adjustment the gauge is only damped,
 * cond(S) is 1e9..1e10, and two correct fp64 solvers differ by eps*cond.  "The device step matches the
 * reference" can then only mean: the device step is as close to the EXACT solution of the reference's
 * equations as the reference's own fp64 path is.  This function supplies that exact solution:
 *
 *   inputs   the fp64 residuals and Jacobian blocks of the last ora_linearize / ora_set_linearization
 *            (scaled by the Jacobi column scaling in fp64 when one is set, exactly as the fp64 path does)
 *   exact    H = J^T J, g = J^T r, Hll + lambda I (+ the eigenvalue gate's regularisation, DECIDED in fp64
 *            exactly as ora_invert_landmark_blocks decides it, explicit_schur.rs:377-442), its inverse,
 *            S = Hcc + lambda I - sum W Hll^-1 W^T, g_red, all accumulated in __float128 (113-bit
 *            significand: products of two doubles are exact, sums lose ~1e-34 relative)
 *   solve    fp64 Cholesky of the rounded S as the preconditioner of an iterative refinement whose residual
 *            g_red - S x is formed in __float128, run until it stalls below 1e-28 |g_red|
 *   output   dc and the back-substituted dl, rounded to fp64 once at the end, reference global column order.
 *
 * No regularisation ladder: the referee is defined only where the undamped-ladder factorisation succeeds
 * (every parity case); otherwise ORA_ERR_FACTORIZATION.  info (optional, 4 doubles): refinement sweeps,
 * final |residual| / |g_red|, first-sweep |correction| / |x| (= the forward error of the plain fp64 solve of
 * the rounded S), number of landmark blocks the gate regularised. */
typedef __float128 q128;

static int q_inv3(const q128 m[9], q128 o[9]) {
    q128 c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    q128 det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (det == 0) return 0;
    o[0] = c00 / det; o[1] = (m[2] * m[7] - m[1] * m[8]) / det; o[2] = (m[1] * m[5] - m[2] * m[4]) / det;
    o[3] = c01 / det; o[4] = (m[0] * m[8] - m[2] * m[6]) / det; o[5] = (m[2] * m[3] - m[0] * m[5]) / det;
    o[6] = c02 / det; o[7] = (m[1] * m[6] - m[0] * m[7]) / det; o[8] = (m[0] * m[4] - m[1] * m[3]) / det;
    return 1;
}

/* replace the stored linearisation (e.g. by the blocks a device exported): the referee then judges a solver on ITS
 * OWN equations.  Row order = observation order; Jpose n_obs x 2 x 6, Jpt / Jintr n_obs x 2 x 3, r n_obs x 2. */
void ora_set_linearization(ora_problem *p, const double *r, const double *Jpose, const double *Jpt, const double *Jintr) {
    memcpy(p->r, r, (size_t)p->n_obs * 16);
    memcpy(p->Jpose, Jpose, (size_t)p->n_obs * 96);
    memcpy(p->Jpt, Jpt, (size_t)p->n_obs * 48);
    if (Jintr) memcpy(p->Jintr, Jintr, (size_t)p->n_obs * 48); else memset(p->Jintr, 0, (size_t)p->n_obs * 48);
}

int ora_solve_augmented_quad(ora_problem *p, double lambda, double *step_out, double *info) {
    const int64_t nc = p->cam_dof, npt = p->n_pt, nobs = p->n_obs, land0 = nc;
    const int has_intr = ora_mode_mask(p->mode) & 1;
    if (nc > 6000 || nobs > 4000000) return ORA_ERR_INPUT;   /* dense quad S: 16 nc^2 bytes */
    /* per observation: the 9 camera-side columns (global column, 2 values) and the 3 landmark columns, scaled in fp64 */
    double *Jc = (double *)malloc((size_t)nobs * 18 * 8 + 8), *Jl = (double *)malloc((size_t)nobs * 6 * 8 + 8);
    int64_t *col = (int64_t *)malloc((size_t)nobs * 9 * 8 + 8);
    for (int64_t i = 0; i < nobs; ++i) {
        uint32_t c = p->cam_idx[i], l = p->pt_idx[i];
        for (int a = 0; a < 9; ++a) {
            int64_t cc = (a < 6) ? p->pose_col[c] + a : p->intr_col[c] + (a - 6);
            double s = p->scaling ? p->scaling[cc] : 1.0;
            for (int rr = 0; rr < 2; ++rr) {
                double v = (a < 6) ? p->Jpose[12 * i + 6 * rr + a] : (has_intr ? p->Jintr[6 * i + 3 * rr + (a - 6)] : 0.0);
                Jc[18 * i + 2 * a + rr] = p->scaling ? v * s : v;
            }
            col[9 * i + a] = cc;
        }
        for (int a = 0; a < 3; ++a) {
            double s = p->scaling ? p->scaling[p->pt_col[l] + a] : 1.0;
            for (int rr = 0; rr < 2; ++rr) Jl[6 * i + 2 * a + rr] = p->scaling ? p->Jpt[6 * i + 3 * rr + a] * s : p->Jpt[6 * i + 3 * rr + a];
        }
    }
    q128 *S = (q128 *)calloc((size_t)nc * (size_t)nc, sizeof(q128));
    q128 *gc = (q128 *)calloc((size_t)nc, sizeof(q128));        /* -g_c, then g_red */
    q128 *Hinv = (q128 *)malloc((size_t)npt * 9 * sizeof(q128));
    q128 *gl = (q128 *)calloc((size_t)npt * 3, sizeof(q128));    /* -g_l */
    q128 *W = (q128 *)malloc((size_t)nobs * 27 * sizeof(q128) + 16);
    double *A = (double *)malloc((size_t)nc * (size_t)nc * 8);
    int rc = ORA_OK;
    int64_t n_reg = 0;
    if (!S || !gc || !Hinv || !gl || !W || !A) rc = ORA_ERR_INPUT;
    if (rc == ORA_OK) {
        /* Hcc, g_c, W_i = Jc_i^T Jl_i (9 x 3), exact products, quad sums */
        for (int64_t i = 0; i < nobs; ++i) {
            const double *jc = Jc + 18 * i, *jl = Jl + 6 * i, *r = p->r + 2 * i;
            const int64_t *cl = col + 9 * i;
            for (int a = 0; a < 9; ++a) {
                for (int b = 0; b < 9; ++b)
                    S[cl[a] * nc + cl[b]] += (q128)jc[2 * a] * jc[2 * b] + (q128)jc[2 * a + 1] * jc[2 * b + 1];
                gc[cl[a]] -= (q128)jc[2 * a] * r[0] + (q128)jc[2 * a + 1] * r[1];
                for (int b = 0; b < 3; ++b)
                    W[27 * i + 3 * a + b] = (q128)jc[2 * a] * jl[2 * b] + (q128)jc[2 * a + 1] * jl[2 * b + 1];
            }
        }
        for (int64_t i = 0; i < nc; ++i) S[i * nc + i] += lambda;
        /* landmark blocks: exact Hll + lambda, the gate decided on the fp64 block the fp64 path builds */
        for (int64_t l = 0; l < npt && rc == ORA_OK; ++l) {
            q128 H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int64_t k = p->pt_ptr[l]; k < p->pt_ptr[l + 1]; ++k) {
                int64_t i = p->pt_obs[k];
                const double *jl = Jl + 6 * i, *r = p->r + 2 * i;
                for (int a = 0; a < 3; ++a) {
                    for (int b = 0; b < 3; ++b) {
                        H[3 * a + b] += (q128)jl[2 * a] * jl[2 * b] + (q128)jl[2 * a + 1] * jl[2 * b + 1];
                        B[3 * a + b] += jl[2 * a] * jl[2 * b] + jl[2 * a + 1] * jl[2 * b + 1];
                    }
                    gl[3 * l + a] -= (q128)jl[2 * a] * r[0] + (q128)jl[2 * a + 1] * r[1];
                }
            }
            B[0] += lambda; B[4] += lambda; B[8] += lambda;
            H[0] += lambda; H[4] += lambda; H[8] += lambda;
            double ev[3];
            sym3_eigenvalues(B, ev);
            double mn = fmin(ev[0], fmin(ev[1], ev[2])), mx = fmax(ev[0], fmax(ev[1], ev[2]));
            double reg = 0.0;
            if (mn < 1e-12) reg = 1e-6 + mx * 1e-6;
            else if (mx / mn > 1e10) reg = mx * 1e-6;
            if (reg != 0.0) { H[0] += reg; H[4] += reg; H[8] += reg; ++n_reg; }
            if (!q_inv3(H, Hinv + 9 * l)) rc = ORA_ERR_SINGULAR;
        }
    }
    if (rc == ORA_OK) {
        /* S -= W_i Hinv W_j^T over all ordered pairs of a landmark's observations (i == j included), g_red.
         * Parallel over landmarks would race on S: parallel over the ROW observation's camera instead -- every thread
         * owns the S rows of a residue class of cameras. */
#pragma omp parallel
        {
            int nth = 1, tid = 0;
#ifdef _OPENMP
            extern int omp_get_num_threads(void); extern int omp_get_thread_num(void);
            nth = omp_get_num_threads(); tid = omp_get_thread_num();
#endif
            for (int64_t l = 0; l < npt; ++l) {
                const q128 *Hi = Hinv + 9 * l;
                for (int64_t k = p->pt_ptr[l]; k < p->pt_ptr[l + 1]; ++k) {
                    const int64_t i = p->pt_obs[k];
                    if ((int)(p->cam_idx[i] % (uint32_t)nth) != tid) continue;
                    q128 Y[27];   /* W_i Hinv */
                    for (int a = 0; a < 9; ++a)
                        for (int b = 0; b < 3; ++b)
                            Y[3 * a + b] = W[27 * i + 3 * a] * Hi[b] + W[27 * i + 3 * a + 1] * Hi[3 + b] + W[27 * i + 3 * a + 2] * Hi[6 + b];
                    for (int a = 0; a < 9; ++a)
                        gc[col[9 * i + a]] -= Y[3 * a] * gl[3 * l] + Y[3 * a + 1] * gl[3 * l + 1] + Y[3 * a + 2] * gl[3 * l + 2];
                    for (int64_t k2 = p->pt_ptr[l]; k2 < p->pt_ptr[l + 1]; ++k2) {
                        const int64_t j = p->pt_obs[k2];
                        for (int a = 0; a < 9; ++a) {
                            q128 *Srow = S + col[9 * i + a] * nc;
                            for (int b = 0; b < 9; ++b)
                                Srow[col[9 * j + b]] -= Y[3 * a] * W[27 * j + 3 * b] + Y[3 * a + 1] * W[27 * j + 3 * b + 1] + Y[3 * a + 2] * W[27 * j + 3 * b + 2];
                        }
                    }
                }
            }
        }
        for (int64_t i = 0; i < nc * nc; ++i) A[i] = (double)S[i];
        if (dense_llt(nc, A) != 0) rc = ORA_ERR_FACTORIZATION;
    }
    double sweeps = 0.0, rel_res = 0.0, first_corr = 0.0;
    if (rc == ORA_OK) {
        q128 *x = (q128 *)calloc((size_t)nc, sizeof(q128)), *res = (q128 *)malloc((size_t)nc * sizeof(q128));
        double *d = (double *)malloc((size_t)nc * 8);
        double gn = 0.0;
        for (int64_t i = 0; i < nc; ++i) { res[i] = gc[i]; gn += (double)gc[i] * (double)gc[i]; }
        gn = sqrt(gn);
        double prev = 1e300;
        for (int it = 0; it < 40; ++it) {
            /* scale the residual so that its fp64 image keeps full relative precision however small it has become */
            double rn = 0.0;
            for (int64_t i = 0; i < nc; ++i) rn = fmax(rn, fabs((double)res[i]));
            if (rn == 0.0) break;
            for (int64_t i = 0; i < nc; ++i) d[i] = (double)(res[i] / rn);
            dense_llt_solve(nc, A, d);
            double dn = 0.0, xn = 0.0;
            for (int64_t i = 0; i < nc; ++i) { x[i] += (q128)d[i] * rn; dn += d[i] * d[i] * rn * rn; xn += (double)x[i] * (double)x[i]; }
            if (it == 1) first_corr = sqrt(dn) / fmax(sqrt(xn), 1e-300);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < nc; ++i) {
                q128 s = gc[i];
                const q128 *Si = S + i * nc;
                for (int64_t j = 0; j < nc; ++j) s -= Si[j] * x[j];
                res[i] = s;
            }
            double r2 = 0.0;
            for (int64_t i = 0; i < nc; ++i) r2 += (double)res[i] * (double)res[i];
            rel_res = sqrt(r2) / fmax(gn, 1e-300);
            sweeps = it + 1;
            if (rel_res < 1e-28 || rel_res > 0.5 * prev) break;
            prev = rel_res;
        }
        /* back-substitution (explicit_schur.rs:980-1029) in quad; output rounded once */
        for (int64_t i = 0; i < nc; ++i) step_out[i] = (double)x[i];
        for (int64_t l = 0; l < npt; ++l) {
            q128 rhs[3] = {gl[3 * l], gl[3 * l + 1], gl[3 * l + 2]};
            for (int64_t k = p->pt_ptr[l]; k < p->pt_ptr[l + 1]; ++k) {
                const int64_t i = p->pt_obs[k];
                for (int a = 0; a < 9; ++a)
                    for (int b = 0; b < 3; ++b) rhs[b] -= W[27 * i + 3 * a + b] * x[col[9 * i + a]];
            }
            const q128 *Hi = Hinv + 9 * l;
            const int64_t lc = p->pt_col[l];
            for (int a = 0; a < 3; ++a) step_out[lc + a] = (double)(Hi[3 * a] * rhs[0] + Hi[3 * a + 1] * rhs[1] + Hi[3 * a + 2] * rhs[2]);
        }
        (void)land0;
        free(x); free(res); free(d);
    }
    if (info) { info[0] = sweeps; info[1] = rel_res; info[2] = first_corr; info[3] = (double)n_reg; }
    free(Jc); free(Jl); free(col); free(S); free(gc); free(Hinv); free(gl); free(W); free(A);
    return rc;
}
