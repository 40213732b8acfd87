// Test-only: compiles apex-solver_amd/csrc/ba_device.hpp for the HOST so that the exact
// per-lane math of the HIP kernels can be compared with the oracle without a GPU.
#include "ba_device.hpp"
using namespace apex;
extern "C" {
int hh_linearize_obs(int dc, const double* pose, const double* intr, const double* pt, const double* uv,
                     double delta, double* r, double* Jc, double* Jl) {
    Cam c; load_cam(pose, intr, c);
    double jl[2][3]; bool ok;
    if (dc == 9) { double jc[2][9]; ok = linearize_obs<9>(c, pt, uv[0], uv[1], delta, r, jc, jl);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 9; ++j) Jc[9 * i + j] = jc[i][j]; }
    else { double jc[2][6]; ok = linearize_obs<6>(c, pt, uv[0], uv[1], delta, r, jc, jl);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 6; ++j) Jc[6 * i + j] = jc[i][j]; }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) Jl[3 * i + j] = jl[i][j];
    return ok;
}
int hh_residual_obs(const double* pose, const double* intr, const double* pt, const double* uv, double delta, double* r) {
    Cam c; load_cam(pose, intr, c);
    return residual_obs(c, pt, uv[0], uv[1], delta, r);
}
int hh_invert_block(const double* B, double* Binv) { return invert_landmark_block(B, Binv); }
void hh_se3_plus(const double* pose, const double* delta, double* out) { se3_plus(pose, delta, out); }
}

// ---- pose-graph device math (pg_device.hpp) ---------------------------------------------------
#include "pg_device.hpp"
extern "C" {
// r[6], J[6][12] = [dr/dk0 | dr/dk1] from raw (un-normalised) 7-vectors, loss-corrected like k_pg_edges
void hh_between_linearize(const double* k0, const double* k1, const double* meas, double delta, double* r, double* J) {
    double a[7], b[7], m[7];
    pose_normalise(k0, a); pose_normalise(k1, b); pose_normalise(meas, m);
    Jac6 Jx[2];
    between_linearize(a, b, m, r, Jx[0], Jx[1]);
    const double sc = pg_huber_scale(delta, r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3] + r[4] * r[4] + r[5] * r[5]);
    for (int i = 0; i < 6; ++i) r[i] *= sc;
    for (int w = 0; w < 2; ++w)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double* o = J + 6 * w;
                o[12 * i + j] = sc * Jx[w].P[3 * i + j];
                o[12 * i + 3 + j] = sc * Jx[w].T[3 * i + j];
                o[12 * (i + 3) + j] = 0.0;
                o[12 * (i + 3) + 3 + j] = sc * Jx[w].P[3 * i + j];
            }
}
// H = Ja^T Jb (6x6) and g = Ja^T r through the structured products the kernel uses
void hh_between_normal(const double* k0, const double* k1, const double* meas, double* H00, double* H11, double* H10, double* g0,
                       double* g1) {
    double a[7], b[7], m[7], r[6];
    pose_normalise(k0, a); pose_normalise(k1, b); pose_normalise(meas, m);
    Jac6 J0, J1;
    between_linearize(a, b, m, r, J0, J1);
    jtj(J0, J0, H00); jtj(J1, J1, H11); jtj(J1, J0, H10);
    jtr(J0, r, g0); jtr(J1, r, g1);
}
}
