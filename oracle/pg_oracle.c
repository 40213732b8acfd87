/* pg_oracle.c -- CPU restatement of apex-solver's SE3 pose-graph path (BASELINE.json configs[1]).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under apex-solver_amd/ links, loads or calls this file; it is
 * the checker used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * What it restates (file:line under the apex-solver tree):
 *   SO3 log / Jl / Jl^-1                       crates/apex-manifolds/src/so3.rs:313-357, 595-646
 *   SE3 from vec / inverse / compose / log     crates/apex-manifolds/src/se3.rs:107-113, 200-206, 242-320
 *   SE3 adjoint                                se3.rs:347-369
 *   Q block, Jr, Jl, Jr^-1, Jl^-1              se3.rs:516-558, 594-706
 *   LieGroup::between                          crates/apex-manifolds/src/lib.rs:401-419
 *   BetweenFactor<SE3>::linearize              src/factors/between_factor.rs:268-322
 *   loss correction                            src/core/corrector.rs:143-181, loss_functions.rs:364-380
 *   SparseCholeskySolver::solve_augmented_equation   src/linalg/sparse/cholesky.rs:159-230
 *   LM loop                                    src/optimizer/levenberg_marquardt.rs:702-817, 823-1031
 *   problem set-up of the G2O binary           bin/pose_graph_g2o.rs:748-830
 *
 * Pinning: the reference is Rust and cannot be built in this image (no cargo/rustc), so the restatement
 * is pinned by the reference's own unit-test assertions re-run as known-answer tests
 * (tests/test_pg_oracle_kat.py: identity -> zero residual, finite-difference Jacobians at the
 * reference's test poses, exp/log round trips, Jr * Jr^-1 = I at the reference's tangent, between,
 * adjoint determinant, small-angle log) and by an independent numpy restatement (tests/np_ref_pg.py).
 *
 * The linear solve is a dense-envelope ("skyline") Cholesky of H + lambda I in vertex order: the
 * reference's faer Llt differs only in elimination order; failure = a non-positive pivot.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PG_OK 0
#define PG_ERR_SINGULAR (-2)
#define PG_SMALL_ANGLE 1e-10 /* apex-manifolds/src/lib.rs:61 */

/* ------------------------------------------------------------------------------------------- */
/* small dense helpers (row-major)                                                              */
/* ------------------------------------------------------------------------------------------- */
static void m3_mul(const double *A, const double *B, double *C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
            T[3 * i + j] = s;
        }
    memcpy(C, T, sizeof T);
}
static void m3_vec(const double *A, const double *v, double *o) {
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
    memcpy(o, t, sizeof t);
}
static void m3_T(const double *A, double *o) {
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * j + i];
    memcpy(o, t, sizeof t);
}
static void hat3(const double v[3], double M[9]) { /* so3.rs:652-664 */
    M[0] = 0.0;   M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2];  M[4] = 0.0;   M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0];  M[8] = 0.0;
}
static void m6_mul(const double *A, const double *B, double *C) {
    double T[36];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += A[6 * i + k] * B[6 * k + j];
            T[6 * i + j] = s;
        }
    memcpy(C, T, sizeof T);
}

/* ------------------------------------------------------------------------------------------- */
/* quaternions [w,x,y,z]                                                                        */
/* ------------------------------------------------------------------------------------------- */
static void q_mul(const double a[4], const double b[4], double o[4]) {
    double t[4];
    t[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    t[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    t[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    t[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    memcpy(o, t, sizeof t);
}
static void q_conj(const double a[4], double o[4]) { o[0] = a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = -a[3]; }
static void q_rot(const double q[4], const double v[3], double o[3]) { /* UnitQuaternion * Vector3 */
    const double *u = q + 1;
    double t[3] = {2.0 * (u[1] * v[2] - u[2] * v[1]), 2.0 * (u[2] * v[0] - u[0] * v[2]), 2.0 * (u[0] * v[1] - u[1] * v[0])};
    double c[3] = {u[1] * t[2] - u[2] * t[1], u[2] * t[0] - u[0] * t[2], u[0] * t[1] - u[1] * t[0]};
    for (int i = 0; i < 3; ++i) o[i] = (t[i] * q[0] + c[i]) + v[i];
}
static void q_to_R(const double q[4], double R[9]) { /* so3.rs:193-195 (to_rotation_matrix) */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double ww = w * w, xx = x * x, yy = y * y, zz = z * z;
    double xy = x * y * 2.0, wz = w * z * 2.0, wy = w * y * 2.0, xz = x * z * 2.0, yz = y * z * 2.0, wx = w * x * 2.0;
    R[0] = ww + xx - yy - zz; R[1] = xy - wz;           R[2] = wy + xz;
    R[3] = wz + xy;           R[4] = ww - xx + yy - zz; R[5] = yz - wx;
    R[6] = xz - wy;           R[7] = wx + yz;           R[8] = ww - xx - yy + zz;
}

/* SE3::from(DVector) -> from_translation_quaternion: normalises twice (se3.rs:107-113, 200-206) */
void pgo_se3_from_vec(const double v[7], double t[3], double q[4]) {
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2];
    double w = v[3], x = v[4], y = v[5], z = v[6];
    for (int pass = 0; pass < 2; ++pass) {
        double n = sqrt(w * w + x * x + y * y + z * z);
        w /= n; x /= n; y /= n; z /= n;
    }
    q[0] = w; q[1] = x; q[2] = y; q[3] = z;
}

/* ------------------------------------------------------------------------------------------- */
/* SO3                                                                                          */
/* ------------------------------------------------------------------------------------------- */
void pgo_so3_log(const double q[4], double th[3]) { /* so3.rs:313-357 */
    double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    double coeff;
    if (s2 > PG_SMALL_ANGLE) {
        double s = sqrt(s2), c = q[0];
        double two = 2.0 * (c < 0.0 ? atan2(-s, -c) : atan2(s, c));
        coeff = two / s;
    } else {
        coeff = 2.0;
    }
    th[0] = q[1] * coeff; th[1] = q[2] * coeff; th[2] = q[3] * coeff;
}
void pgo_so3_exp(const double th[3], double q[4]) { /* so3.rs:558-583 */
    double t2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    if (t2 > PG_SMALL_ANGLE) {
        double hx = th[0] / 2.0, hy = th[1] / 2.0, hz = th[2] / 2.0;
        double n = sqrt(hx * hx + hy * hy + hz * hz), s = sin(n) / n;
        q[0] = cos(n); q[1] = hx * s; q[2] = hy * s; q[3] = hz * s;
    } else {
        double w = 1.0, x = th[0] / 2.0, y = th[1] / 2.0, z = th[2] / 2.0;
        double n = sqrt(w * w + x * x + y * y + z * z);
        q[0] = w / n; q[1] = x / n; q[2] = y / n; q[3] = z / n;
    }
}
void pgo_so3_left_jacobian(const double th[3], double J[9]) { /* so3.rs:595-612 */
    double a = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    double K[9], K2[9];
    hat3(th, K);
    m3_mul(K, K, K2);
    double c1, c2;
    if (a <= PG_SMALL_ANGLE) { c1 = 0.5; c2 = 0.0; }
    else {
        double t = sqrt(a);
        c1 = (1.0 - cos(t)) / a;
        c2 = (t - sin(t)) / (a * t);
    }
    for (int i = 0; i < 9; ++i) J[i] = c1 * K[i] + c2 * K2[i];
    J[0] += 1.0; J[4] += 1.0; J[8] += 1.0;
}
void pgo_so3_left_jacobian_inv(const double th[3], double J[9]) { /* so3.rs:628-646 */
    double a = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    double K[9], K2[9];
    hat3(th, K);
    m3_mul(K, K, K2);
    double c2 = 0.0;
    if (a > PG_SMALL_ANGLE) {
        double t = sqrt(a);
        c2 = 1.0 / a - (1.0 + cos(t)) / (2.0 * t * sin(t));
    }
    for (int i = 0; i < 9; ++i) J[i] = -0.5 * K[i] + c2 * K2[i];
    J[0] += 1.0; J[4] += 1.0; J[8] += 1.0;
}

/* ------------------------------------------------------------------------------------------- */
/* SE3 (pose = t[3], q[4]; tangent = [rho, theta])                                               */
/* ------------------------------------------------------------------------------------------- */
static void se3_inverse(const double t[3], const double q[4], double ti[3], double qi[4]) { /* se3.rs:242-251 */
    double r[3];
    q_conj(q, qi);
    q_rot(qi, t, r);
    ti[0] = -r[0]; ti[1] = -r[1]; ti[2] = -r[2];
}
static void se3_compose(const double ta[3], const double qa[4], const double tb[3], const double qb[4], double t[3],
                        double q[4]) { /* se3.rs:272-293 */
    double r[3], qq[4];
    q_mul(qa, qb, qq);
    q_rot(qa, tb, r);
    t[0] = r[0] + ta[0]; t[1] = r[1] + ta[1]; t[2] = r[2] + ta[2];
    memcpy(q, qq, sizeof qq);
}
static void se3_adjoint(const double t[3], const double q[4], double A[36]) { /* se3.rs:347-369 */
    double R[9], T[9], TR[9];
    q_to_R(q, R);
    hat3(t, T);
    m3_mul(T, R, TR);
    memset(A, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[6 * i + j] = R[3 * i + j];
            A[6 * (i + 3) + (j + 3)] = R[3 * i + j];
            A[6 * i + (j + 3)] = TR[3 * i + j];
        }
}

/* Q(rho, theta) exactly as coded (se3.rs:520-558), including the d coefficient as written there */
void pgo_se3_q_block(const double rho[3], const double th[3], double Q[9]) {
    double Rk[9], Tk[9];
    hat3(rho, Rk);
    hat3(th, Tk);
    double t2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    double a = 0.5, b = 1.0 / 6.0 + 1.0 / 120.0 * t2, c = -1.0 / 24.0 + 1.0 / 720.0 * t2, d = -1.0 / 60.0;
    if (t2 > PG_SMALL_ANGLE) {
        double tn = sqrt(t2), tn3 = tn * t2, tn4 = t2 * t2, tn5 = tn3 * t2;
        double s = sin(tn), co = cos(tn);
        b = (tn - s) / tn3;
        c = (1.0 - t2 / 2.0 - co) / tn4;
        d = (c - 3.0) * (tn - s - tn3 / 6.0) / tn5;
    }
    double tr[9], rt[9], trt[9], rtt[9], rttT[9], trtt[9];
    m3_mul(Tk, Rk, tr);
    m3_mul(Rk, Tk, rt);
    m3_mul(tr, Tk, trt);
    m3_mul(rt, Tk, rtt);
    m3_T(rtt, rttT);
    m3_mul(trt, Tk, trtt);
    for (int i = 0; i < 9; ++i)
        Q[i] = Rk[i] * a + (tr[i] + rt[i] + trt[i]) * b - (rtt[i] - rttT[i] - trt[i] * 3.0) * c - trtt[i] * d;
}

static void blocks_to_6x6(const double D[9], const double TRb[9], double J[36]) {
    memset(J, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            J[6 * i + j] = D[3 * i + j];
            J[6 * (i + 3) + (j + 3)] = D[3 * i + j];
            J[6 * i + (j + 3)] = TRb[3 * i + j];
        }
}
void pgo_se3_right_jacobian(const double tau[6], double J[36]) { /* se3.rs:594-604 */
    double nth[3] = {-tau[3], -tau[4], -tau[5]}, nrho[3] = {-tau[0], -tau[1], -tau[2]};
    double Jl[9], D[9], Q[9];
    pgo_so3_left_jacobian(nth, Jl); /* SO3Tangent(-theta).right_jacobian() = Jl(-theta)^T */
    m3_T(Jl, D);
    pgo_se3_q_block(nrho, nth, Q);
    blocks_to_6x6(D, Q, J);
}
void pgo_se3_left_jacobian(const double tau[6], double J[36]) { /* se3.rs:615-627 */
    double D[9], Q[9];
    pgo_so3_left_jacobian(tau + 3, D);
    pgo_se3_q_block(tau, tau + 3, Q);
    blocks_to_6x6(D, Q, J);
}
void pgo_se3_right_jacobian_inv(const double tau[6], double J[36]) { /* se3.rs:652-666 */
    double nth[3] = {-tau[3], -tau[4], -tau[5]}, nrho[3] = {-tau[0], -tau[1], -tau[2]};
    double D[9], Q[9], T[9];
    pgo_so3_left_jacobian_inv(tau + 3, D); /* SO3Tangent(theta).left_jacobian_inv() */
    pgo_se3_q_block(nrho, nth, Q);
    m3_mul(D, Q, T);
    m3_mul(T, D, T);
    for (int i = 0; i < 9; ++i) T[i] = -1.0 * T[i];
    blocks_to_6x6(D, T, J);
}
void pgo_se3_left_jacobian_inv(const double tau[6], double J[36]) { /* se3.rs:687-700 */
    double D[9], Q[9], T[9];
    pgo_so3_left_jacobian_inv(tau + 3, D);
    pgo_se3_q_block(tau, tau + 3, Q);
    m3_mul(D, Q, T);
    m3_mul(T, D, T);
    for (int i = 0; i < 9; ++i) T[i] = -1.0 * T[i];
    blocks_to_6x6(D, T, J);
}

static void se3_log_tq(const double t[3], const double q[4], double tau[6]) { /* se3.rs:308-320 */
    double th[3], Ji[9], rho[3];
    pgo_so3_log(q, th);
    pgo_so3_left_jacobian_inv(th, Ji);
    m3_vec(Ji, t, rho);
    tau[0] = rho[0]; tau[1] = rho[1]; tau[2] = rho[2];
    tau[3] = th[0]; tau[4] = th[1]; tau[5] = th[2];
}
void pgo_se3_log(const double pose[7], double tau[6]) {
    double t[3], q[4];
    pgo_se3_from_vec(pose, t, q);
    se3_log_tq(t, q, tau);
}
void pgo_se3_exp(const double tau[6], double pose[7]) { /* se3.rs:569-583 */
    double q[4], Jl[9], t[3];
    pgo_so3_exp(tau + 3, q);
    pgo_so3_left_jacobian(tau + 3, Jl);
    m3_vec(Jl, tau, t);
    pose[0] = t[0]; pose[1] = t[1]; pose[2] = t[2];
    pose[3] = q[0]; pose[4] = q[1]; pose[5] = q[2]; pose[6] = q[3];
}
void pgo_se3_adjoint(const double pose[7], double A[36]) {
    double t[3], q[4];
    pgo_se3_from_vec(pose, t, q);
    se3_adjoint(t, q, A);
}
void pgo_se3_inverse(const double pose[7], double out[7]) {
    double t[3], q[4], ti[3], qi[4];
    pgo_se3_from_vec(pose, t, q);
    se3_inverse(t, q, ti, qi);
    out[0] = ti[0]; out[1] = ti[1]; out[2] = ti[2]; out[3] = qi[0]; out[4] = qi[1]; out[5] = qi[2]; out[6] = qi[3];
}
void pgo_se3_compose(const double a[7], const double b[7], double out[7]) {
    double ta[3], qa[4], tb[3], qb[4], t[3], q[4];
    pgo_se3_from_vec(a, ta, qa);
    pgo_se3_from_vec(b, tb, qb);
    se3_compose(ta, qa, tb, qb, t, q);
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; out[3] = q[0]; out[4] = q[1]; out[5] = q[2]; out[6] = q[3];
}
void pgo_se3_between(const double a[7], const double b[7], double out[7]) { /* lib.rs:401-419: a^-1 * b */
    double ai[7];
    pgo_se3_inverse(a, ai);
    pgo_se3_compose(ai, b, out);
}
/* right-plus retraction x (+) tau = x * Exp(tau) (lib.rs:269-292).  The variable is kept as an SE3 value
 * between iterations, so the stored quaternion is used as it is (never re-normalised) and the result is
 * stored un-normalised, exactly like the reference. */
void pgo_se3_plus(const double pose[7], const double tau[6], double out[7]) {
    double e[7], tn[3], qn[4];
    pgo_se3_exp(tau, e);
    se3_compose(pose, pose + 3, e, e + 3, tn, qn);
    out[0] = tn[0]; out[1] = tn[1]; out[2] = tn[2]; out[3] = qn[0]; out[4] = qn[1]; out[5] = qn[2]; out[6] = qn[3];
}

/* ------------------------------------------------------------------------------------------- */
/* BetweenFactor<SE3>::linearize (between_factor.rs:268-322)                                     */
/*   r = Log((k1^-1 k0) * meas) ; J = [dr/dk0 | dr/dk1] 6 x 12 row-major                         */
/* ------------------------------------------------------------------------------------------- */
void pgo_between_linearize(const double k0[7], const double k1[7], const double meas[7], double r[6], double *J /* 72 or NULL */) {
    double t0[3], q0[4], t1[3], q1[4], tm[3], qm[4];
    pgo_se3_from_vec(k0, t0, q0);
    pgo_se3_from_vec(k1, t1, q1);
    pgo_se3_from_vec(meas, tm, qm);
    /* step 1: k1.between(k0) = k1^-1 * k0 ; d/dk1 = -Adj(result^-1), d/dk0 = I */
    double t1i[3], q1i[4], tA[3], qA[4];
    se3_inverse(t1, q1, t1i, q1i);
    se3_compose(t1i, q1i, t0, q0, tA, qA);
    /* step 2: diff = A * meas ; d/dA = Adj(meas^-1) */
    double tD[3], qD[4];
    se3_compose(tA, qA, tm, qm, tD, qD);
    /* step 3: log ; d/ddiff = Jr^-1(r) */
    se3_log_tq(tD, qD, r);
    if (!J) return;
    double tAi[3], qAi[4], tmi[3], qmi[4];
    se3_inverse(tA, qA, tAi, qAi);
    se3_inverse(tm, qm, tmi, qmi);
    double j_k1[36], j_diff[36], j_log[36], d0[36], d1[36], J0[36], J1[36];
    se3_adjoint(tAi, qAi, j_k1);
    for (int i = 0; i < 36; ++i) j_k1[i] = -j_k1[i];
    se3_adjoint(tmi, qmi, j_diff);
    pgo_se3_right_jacobian_inv(r, j_log);
    memcpy(d0, j_diff, sizeof d0);   /* j_diff * I */
    m6_mul(j_diff, j_k1, d1);
    m6_mul(j_log, d0, J0);
    m6_mul(j_log, d1, J1);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            J[12 * i + j] = J0[6 * i + j];
            J[12 * i + 6 + j] = J1[6 * i + j];
        }
}

/* HuberLoss::evaluate + Corrector::new (loss_functions.rs:364-380, corrector.rs:143-181): for Huber
 * rho'' <= 0, so the correction is a plain scaling of r and J by sqrt(rho'). */
static double huber_scale(double delta, double s) {
    if (delta <= 0.0) return 1.0;
    if (s > delta * delta) return sqrt(delta / sqrt(s));
    return 1.0;
}

/* ------------------------------------------------------------------------------------------- */
/* problem                                                                                      */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t n_v, n_e, total_dof;
    int64_t *from, *to, *pose_col;
    double *meas, *poses;
    uint8_t *fix; /* [n_v][6] */
    double huber_delta;
    double *r, *J; /* last linearisation: [n_e][6], [n_e][72] */
    double *scaling; /* Jacobi column scaling (optimizer/mod.rs:749-763), caller's column order; NULL = off */
    /* PriorFactor blocks on SE3 variables (src/factors/prior_factor.rs:96-108; the gauge of the reference's pose-graph
     * integration test, tests/integration_tests.rs:98-118): r = to_vector(x) - data (7 rows), J = the 7 x 7 identity of
     * which the linearizer keeps the variable's 6 tangent columns (src/linearizer/cpu/sparse.rs:201-204), i.e. rows 0..5
     * are e_0..e_5 and row 6 (the quaternion's k component) has no Jacobian; each block has its own Huber loss. */
    int64_t n_prior;
    int64_t *prior_v;
    double *prior_data;  /* [n_prior][7] */
    double *prior_delta; /* [n_prior], <= 0: no loss */
    double *prior_r;     /* last evaluation, corrected: [n_prior][7] */
    double *prior_sc;    /* sqrt(rho') of the last evaluation */
} pgo_problem;

static void *dupmem(const void *src, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (src && bytes) memcpy(p, src, bytes); else if (bytes) memset(p, 0, bytes);
    return p;
}
pgo_problem *pgo_create(int64_t n_v, int64_t n_e, const int64_t *from, const int64_t *to, const double *meas,
                        const int64_t *pose_col, const uint8_t *fix, double huber_delta) {
    pgo_problem *p = (pgo_problem *)calloc(1, sizeof *p);
    p->n_v = n_v; p->n_e = n_e; p->total_dof = 6 * n_v;
    p->from = (int64_t *)dupmem(from, (size_t)n_e * 8);
    p->to = (int64_t *)dupmem(to, (size_t)n_e * 8);
    p->pose_col = (int64_t *)dupmem(pose_col, (size_t)n_v * 8);
    p->meas = (double *)dupmem(meas, (size_t)n_e * 56);
    p->poses = (double *)dupmem(NULL, (size_t)n_v * 56);
    p->fix = (uint8_t *)dupmem(fix, (size_t)n_v * 6);
    p->huber_delta = huber_delta;
    p->r = (double *)dupmem(NULL, (size_t)n_e * 48);
    p->J = (double *)dupmem(NULL, (size_t)n_e * 576);
    return p;
}
void pgo_destroy(pgo_problem *p) {
    if (!p) return;
    free(p->from); free(p->to); free(p->pose_col); free(p->meas); free(p->poses); free(p->fix); free(p->r); free(p->J);
    free(p->scaling);
    free(p->prior_v); free(p->prior_data); free(p->prior_delta); free(p->prior_r); free(p->prior_sc);
    free(p);
}
/* replaces the set of prior blocks (n = 0: none) */
void pgo_set_priors(pgo_problem *p, int64_t n, const int64_t *vertex, const double *data7, const double *huber_delta) {
    free(p->prior_v); free(p->prior_data); free(p->prior_delta); free(p->prior_r); free(p->prior_sc);
    p->n_prior = n;
    p->prior_v = (int64_t *)dupmem(vertex, (size_t)n * 8);
    p->prior_data = (double *)dupmem(data7, (size_t)n * 56);
    p->prior_delta = (double *)dupmem(huber_delta, (size_t)n * 8);
    p->prior_r = (double *)dupmem(NULL, (size_t)n * 56);
    p->prior_sc = (double *)dupmem(NULL, (size_t)n * 8);
}
/* the variable as the factors see it: SE3::from(DVector) normalises the quaternion, to_vector returns [t, w, i, j, k] */
static void pose_as_vector(const double *pose7, double out[7]) {
    double t[3], q[4];
    pgo_se3_from_vec(pose7, t, q);
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; out[3] = q[0]; out[4] = q[1]; out[5] = q[2]; out[6] = q[3];
}
/* evaluates every prior block (corrected residual kept in prior_r, scale in prior_sc); returns sum |r~|^2 */
static double eval_priors(pgo_problem *p) {
    double ss = 0.0;
    for (int64_t k = 0; k < p->n_prior; ++k) {
        double x[7], *r = p->prior_r + 7 * k, s = 0.0;
        pose_as_vector(p->poses + 7 * p->prior_v[k], x);
        for (int a = 0; a < 7; ++a) { r[a] = x[a] - p->prior_data[7 * k + a]; s += r[a] * r[a]; }
        const double sc = huber_scale(p->prior_delta[k], s);
        p->prior_sc[k] = sc;
        for (int a = 0; a < 7; ++a) { r[a] *= sc; ss += r[a] * r[a]; }
    }
    return ss;
}
void pgo_prior_residuals(const pgo_problem *p, double *r7_out) { memcpy(r7_out, p->prior_r, (size_t)p->n_prior * 56); }
void pgo_set_params(pgo_problem *p, const double *poses) { memcpy(p->poses, poses, (size_t)p->n_v * 56); }
void pgo_get_params(const pgo_problem *p, double *poses) { memcpy(poses, p->poses, (size_t)p->n_v * 56); }

/* compute_residual_sparse + compute_cost: 0.5 |r|^2 over the loss-corrected residuals */
double pgo_residuals(pgo_problem *p, double *r_out) {
    double ss = 0.0;
#pragma omp parallel for reduction(+ : ss) schedule(static)
    for (int64_t e = 0; e < p->n_e; ++e) {
        double r[6];
        pgo_between_linearize(p->poses + 7 * p->from[e], p->poses + 7 * p->to[e], p->meas + 7 * e, r, NULL);
        double s = 0.0;
        for (int a = 0; a < 6; ++a) s += r[a] * r[a];
        double sc = huber_scale(p->huber_delta, s);
        for (int a = 0; a < 6; ++a) {
            r[a] *= sc;
            ss += r[a] * r[a];
            if (r_out) r_out[6 * e + a] = r[a];
        }
    }
    ss += eval_priors(p);
    double nrm = sqrt(ss);
    return 0.5 * nrm * nrm;
}

double pgo_linearize(pgo_problem *p, double *r_out, double *J_out) {
    double ss = 0.0;
#pragma omp parallel for reduction(+ : ss) schedule(static)
    for (int64_t e = 0; e < p->n_e; ++e) {
        double *r = p->r + 6 * e, *J = p->J + 72 * e;
        pgo_between_linearize(p->poses + 7 * p->from[e], p->poses + 7 * p->to[e], p->meas + 7 * e, r, J);
        double s = 0.0;
        for (int a = 0; a < 6; ++a) s += r[a] * r[a];
        double sc = huber_scale(p->huber_delta, s);
        if (sc != 1.0) {
            for (int a = 0; a < 6; ++a) r[a] *= sc;
            for (int a = 0; a < 72; ++a) J[a] *= sc;
        }
        for (int a = 0; a < 6; ++a) ss += r[a] * r[a];
    }
    if (r_out) memcpy(r_out, p->r, (size_t)p->n_e * 48);
    if (J_out) memcpy(J_out, p->J, (size_t)p->n_e * 576);
    ss += eval_priors(p);
    double nrm = sqrt(ss);
    return 0.5 * nrm * nrm;
}

/* Dense H = J^T J (total_dof^2, row-major) and g = J^T r in the caller's column order. */
void pgo_normal_equations(const pgo_problem *p, double *H /* may be NULL */, double *g) {
    const int64_t n = p->total_dof;
    if (H) memset(H, 0, (size_t)n * (size_t)n * 8);
    memset(g, 0, (size_t)n * 8);
    for (int64_t e = 0; e < p->n_e; ++e) {
        const double *J = p->J + 72 * e, *r = p->r + 6 * e;
        const int64_t col[2] = {p->pose_col[p->from[e]], p->pose_col[p->to[e]]};
        for (int a = 0; a < 12; ++a) {
            const int64_t ca = col[a / 6] + a % 6;
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += J[12 * k + a] * r[k];
            g[ca] += s;
            if (!H) continue;
            for (int b = 0; b < 12; ++b) {
                const int64_t cb = col[b / 6] + b % 6;
                double h = 0.0;
                for (int k = 0; k < 6; ++k) h += J[12 * k + a] * J[12 * k + b];
                H[ca * n + cb] += h;
            }
        }
    }
    for (int64_t k = 0; k < p->n_prior; ++k) {   /* J~ = sc [I6; 0], r~ = sc r */
        const int64_t c0 = p->pose_col[p->prior_v[k]];
        const double sc = p->prior_sc[k];
        for (int a = 0; a < 6; ++a) {
            g[c0 + a] += sc * p->prior_r[7 * k + a];
            if (H) H[(c0 + a) * n + c0 + a] += sc * sc;
        }
    }
}

/* compute_column_norms (linearizer/mod.rs:229-239) of the last linearisation, caller's column order */
void pgo_column_norms(const pgo_problem *p, double *norms_out) {
    memset(norms_out, 0, (size_t)p->total_dof * 8);
    for (int64_t e = 0; e < p->n_e; ++e) {
        const double *J = p->J + 72 * e;
        const int64_t col[2] = {p->pose_col[p->from[e]], p->pose_col[p->to[e]]};
        for (int a = 0; a < 12; ++a)
            for (int k = 0; k < 6; ++k) norms_out[col[a / 6] + a % 6] += J[12 * k + a] * J[12 * k + a];
    }
    for (int64_t k = 0; k < p->n_prior; ++k)
        for (int a = 0; a < 6; ++a) norms_out[p->pose_col[p->prior_v[k]] + a] += p->prior_sc[k] * p->prior_sc[k];
    for (int64_t i = 0; i < p->total_dof; ++i) norms_out[i] = sqrt(norms_out[i]);
}

/* apply_column_scaling (linearizer/mod.rs:241-253) as a state of the problem: the following solves run on
 * J diag(scaling) and return the scaled step and gradient; NULL switches it off. */
void pgo_set_column_scaling(pgo_problem *p, const double *scaling) {
    free(p->scaling); p->scaling = NULL;
    if (scaling) p->scaling = (double *)dupmem(scaling, (size_t)p->total_dof * 8);
}

/* (J^T J + lambda I) dx = -J^T r by an envelope Cholesky in VERTEX order (cholesky.rs:159-230).
 * step/grad come back in the caller's column order.  PG_ERR_SINGULAR on a non-positive pivot. */
int pgo_solve_augmented(pgo_problem *p, double lambda, double *step_out, double *grad_out) {
    const int64_t n = p->total_dof, nv = p->n_v;
    /* envelope: first vertex connected to each vertex */
    int64_t *first = (int64_t *)malloc((size_t)nv * 8);
    for (int64_t v = 0; v < nv; ++v) first[v] = v;
    for (int64_t e = 0; e < p->n_e; ++e) {
        int64_t a = p->from[e], b = p->to[e];
        if (a < b) { if (a < first[b]) first[b] = a; } else { if (b < first[a]) first[a] = b; }
    }
    int64_t *rs = (int64_t *)malloc((size_t)(n + 1) * 8), *f = (int64_t *)malloc((size_t)n * 8);
    rs[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        f[i] = 6 * first[i / 6];
        rs[i + 1] = rs[i] + (i - f[i] + 1);
    }
    double *L = (double *)calloc((size_t)rs[n], 8), *g = (double *)calloc((size_t)n, 8);
#define ENV(i, j) L[rs[i] + ((j) - f[i])]
    for (int64_t e = 0; e < p->n_e; ++e) {
        const double *J0 = p->J + 72 * e, *r = p->r + 6 * e;
        const int64_t vv[2] = {p->from[e], p->to[e]};
        double Js[72];
        const double *J = J0;
        if (p->scaling) {
            for (int a = 0; a < 12; ++a) {
                const double sc = p->scaling[p->pose_col[vv[a / 6]] + a % 6];
                for (int k = 0; k < 6; ++k) Js[12 * k + a] = J0[12 * k + a] * sc;
            }
            J = Js;
        }
        for (int a = 0; a < 12; ++a) {
            const int64_t ia = 6 * vv[a / 6] + a % 6;
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += J[12 * k + a] * r[k];
            g[ia] += s;
            for (int b = 0; b < 12; ++b) {
                const int64_t ib = 6 * vv[b / 6] + b % 6;
                if (ib > ia) continue;
                double h = 0.0;
                for (int k = 0; k < 6; ++k) h += J[12 * k + a] * J[12 * k + b];
                ENV(ia, ib) += h;
            }
        }
    }
    for (int64_t k = 0; k < p->n_prior; ++k) {
        const int64_t v = p->prior_v[k];
        for (int a = 0; a < 6; ++a) {
            const double cs = p->scaling ? p->scaling[p->pose_col[v] + a] : 1.0, j = p->prior_sc[k] * cs;
            g[6 * v + a] += j * p->prior_r[7 * k + a];
            ENV(6 * v + a, 6 * v + a) += j * j;
        }
    }
    for (int64_t i = 0; i < n; ++i) ENV(i, i) += lambda;
    int rc = PG_OK;
    for (int64_t i = 0; i < n && rc == PG_OK; ++i) {
        for (int64_t j = f[i]; j <= i; ++j) {
            int64_t k0 = f[i] > f[j] ? f[i] : f[j];
            double s = ENV(i, j);
            const double *Li = &ENV(i, k0), *Lj = &ENV(j, k0);
            for (int64_t k = 0; k < j - k0; ++k) s -= Li[k] * Lj[k];
            if (j < i) ENV(i, j) = s / ENV(j, j);
            else {
                if (!(s > 0.0)) { rc = PG_ERR_SINGULAR; break; }
                ENV(i, i) = sqrt(s);
            }
        }
    }
    if (rc == PG_OK) {
        double *x = (double *)malloc((size_t)n * 8);
        for (int64_t i = 0; i < n; ++i) {
            double s = -g[i];
            for (int64_t k = f[i]; k < i; ++k) s -= ENV(i, k) * x[k];
            x[i] = s / ENV(i, i);
        }
        for (int64_t i = n - 1; i >= 0; --i) {
            x[i] /= ENV(i, i);
            for (int64_t k = f[i]; k < i; ++k) x[k] -= ENV(i, k) * x[i];
        }
        for (int64_t v = 0; v < nv; ++v)
            for (int a = 0; a < 6; ++a) {
                if (step_out) step_out[p->pose_col[v] + a] = x[6 * v + a];
                if (grad_out) grad_out[p->pose_col[v] + a] = g[6 * v + a];
            }
        free(x);
    }
#undef ENV
    free(L); free(g); free(rs); free(f); free(first);
    return rc;
}

/* SparseCholeskySolver::solve_augmented_equation on an arbitrary dense Jacobian (row-major n_rows x n_cols):
 * (J^T J + lambda I) dx = -J^T r by LL^T; PG_ERR_SINGULAR on a non-positive pivot ("Cholesky factorization
 * failed (matrix may be singular)", cholesky.rs:213-219).  Used to replay the reference's own solver unit tests
 * (cholesky.rs:266-470) on the same factorisation code path semantics. */
int pgo_solve_dense_jacobian(int64_t n_rows, int64_t n_cols, const double *J, const double *r, double lambda, double *dx,
                             double *grad_out) {
    const int64_t n = n_cols;
    if (n == 0) return PG_OK;
    double *H = (double *)calloc((size_t)n * (size_t)n, 8), *g = (double *)calloc((size_t)n, 8);
    for (int64_t k = 0; k < n_rows; ++k)
        for (int64_t a = 0; a < n; ++a) {
            const double ja = J[k * n + a];
            if (ja == 0.0) continue;
            g[a] += ja * r[k];
            for (int64_t b = 0; b <= a; ++b) H[a * n + b] += ja * J[k * n + b];
        }
    for (int64_t a = 0; a < n; ++a) H[a * n + a] += lambda;
    int rc = PG_OK;
    for (int64_t i = 0; i < n && rc == PG_OK; ++i)
        for (int64_t j = 0; j <= i; ++j) {
            double sacc = H[i * n + j];
            for (int64_t k = 0; k < j; ++k) sacc -= H[i * n + k] * H[j * n + k];
            if (j < i) H[i * n + j] = sacc / H[j * n + j];
            else {
                if (!(sacc > 0.0)) { rc = PG_ERR_SINGULAR; break; }
                H[i * n + i] = sqrt(sacc);
            }
        }
    if (rc == PG_OK) {
        for (int64_t i = 0; i < n; ++i) {
            double sacc = -g[i];
            for (int64_t k = 0; k < i; ++k) sacc -= H[i * n + k] * dx[k];
            dx[i] = sacc / H[i * n + i];
        }
        for (int64_t i = n - 1; i >= 0; --i) {
            dx[i] /= H[i * n + i];
            for (int64_t k = 0; k < i; ++k) dx[k] -= H[i * n + k] * dx[i];
        }
    }
    if (grad_out) memcpy(grad_out, g, (size_t)n * 8);
    free(H); free(g);
    return rc;
}

/* apply_parameter_step / apply_negative_parameter_step (optimizer/mod.rs:309-356): fixed DOF are
 * zeroed in the step first (problem.rs:185-197). */
void pgo_apply_step(pgo_problem *p, const double *step, double sign) {
    for (int64_t v = 0; v < p->n_v; ++v) {
        double d[6], out[7];
        for (int a = 0; a < 6; ++a) {
            d[a] = sign * step[p->pose_col[v] + a];
            if (p->fix[6 * v + a]) d[a] = 0.0;
        }
        pgo_se3_plus(p->poses + 7 * v, d, out);
        memcpy(p->poses + 7 * v, out, sizeof out);
    }
}
double pgo_parameter_norm(const pgo_problem *p) { /* optimizer/mod.rs:458-467 */
    double s = 0.0;
    for (int64_t i = 0; i < 7 * p->n_v; ++i) s += p->poses[i] * p->poses[i];
    return sqrt(s);
}

/* LM loop (levenberg_marquardt.rs:823-1031); same conventions as ba_oracle.c's ora_lm_optimize.
 * cfg: [max_iterations, cost_tol, param_tol, grad_tol, damping, damping_min, damping_max, nu,
 *       trust_region_radius, min_trust_region_radius, min_cost_threshold, use_jacobi_scaling]; hist rows of 8:
 * [cost_after, damping_after, rho, accepted, grad_norm, step_norm, predicted_reduction, trial_cost] */
int pgo_lm_optimize(pgo_problem *p, double *cfg, double *hist, int hist_rows, int *iterations_out,
                    double *initial_cost_out, double *final_cost_out, double *params_out /* optional: hist_rows x 7 n_v, params BEFORE each iteration */) {
    const int max_it = (int)cfg[0];
    const double cost_tol = cfg[1], param_tol = cfg[2], grad_tol = cfg[3], dmin = cfg[5], dmax = cfg[6];
    double lambda = cfg[4], nu = cfg[7];
    double cost = pgo_residuals(p, NULL);
    if (initial_cost_out) *initial_cost_out = cost;
    const int64_t n = p->total_dof;
    double *step = (double *)malloc((size_t)n * 8), *grad = (double *)malloc((size_t)n * 8);
    int iteration = 0, status = 1;
    for (;;) {
        if (params_out && iteration < hist_rows) memcpy(params_out + (size_t)iteration * 7 * p->n_v, p->poses, (size_t)p->n_v * 56);
        pgo_linearize(p, NULL, NULL);
        if (cfg[11] != 0.0 && iteration == 0) { /* process_jacobian_generic (optimizer/mod.rs:749-763) */
            pgo_column_norms(p, step);
            for (int64_t i = 0; i < n; ++i) step[i] = 1.0 / (1.0 + step[i]);
            pgo_set_column_scaling(p, step);
        }
        if (pgo_solve_augmented(p, lambda, step, grad) != PG_OK) { status = 100; break; }
        if (p->scaling) /* apply_inverse_scaling (levenberg_marquardt.rs:749-757); the gradient stays scaled */
            for (int64_t i = 0; i < n; ++i) step[i] *= p->scaling[i];
        double gn = 0.0, sn = 0.0, pred = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            gn += grad[i] * grad[i];
            sn += step[i] * step[i];
            pred += step[i] * (lambda * step[i] - grad[i]);
        }
        gn = sqrt(gn); sn = sqrt(sn); pred *= 0.5;
        pgo_apply_step(p, step, 1.0);
        double new_cost = pgo_residuals(p, NULL);
        double actual = cost - new_cost;
        double rho = fabs(pred) < 1e-15 ? (actual > 0.0 ? 1.0 : 0.0) : actual / pred;
        double cost_reduction = 0.0;
        int accepted;
        if (rho > 0.0) {
            double coff = 2.0 * rho - 1.0;
            lambda *= fmax(1.0 / 3.0, 1.0 - coff * coff * coff);
            lambda = fmax(lambda, dmin);
            nu = 2.0; accepted = 1;
            cost_reduction = cost - new_cost; cost = new_cost;
        } else {
            lambda *= nu; nu *= 2.0; lambda = fmin(lambda, dmax); accepted = 0;
            pgo_apply_step(p, step, -1.0);
        }
        if (hist && iteration < hist_rows) {
            double *h = hist + (size_t)iteration * 8;
            h[0] = cost; h[1] = lambda; h[2] = rho; h[3] = accepted; h[4] = gn; h[5] = sn; h[6] = pred; h[7] = new_cost;
        }
        double pnorm = pgo_parameter_norm(p);
        double cost_before = accepted ? cost + cost_reduction : cost;
        int st = -1;
        if (!isfinite(cost) || !isfinite(sn) || !isfinite(gn)) st = 11;
        else if (iteration >= max_it) st = 1;
        else if (accepted) {
            if (gn < grad_tol) st = 4;
            if (st < 0 && iteration > 0) {
                double rel = param_tol * (pnorm + param_tol);
                if (sn <= rel) st = 3;
                else if (fabs(cost_before - cost) / fmax(cost_before, 1e-10) < cost_tol) st = 2;
            }
            if (st < 0 && cfg[10] >= 0.0 && cost < cfg[10]) st = 9;
            if (st < 0 && cfg[8] < cfg[9]) st = 8;
        }
        if (st >= 0) { status = st; ++iteration; break; }
        ++iteration;
    }
    cfg[4] = lambda; cfg[7] = nu;
    if (iterations_out) *iterations_out = iteration;
    if (final_cost_out) *final_cost_out = cost;
    if (cfg[11] != 0.0) pgo_set_column_scaling(p, NULL);
    free(step); free(grad);
    return status;
}
