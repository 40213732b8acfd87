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
