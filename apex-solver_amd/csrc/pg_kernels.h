// pg_kernels.h -- launchers of the SE3 pose-graph kernels (BASELINE.json configs[1]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ba_kernels.h"

namespace apex {

constexpr int kPoseStride = 8;      // doubles per prepared pose / measurement: t(3) q(4) pad
constexpr int kVertsPerTile = kNB / 6;  // 24 vertices per 144-row tile

// Read-only view of one parameter set + the edge list (internal vertex numbering).
struct PGView {
    int64_t n_v, n_e;
    const double* posep;     // [n_v][8] prepared poses (unit quaternion)
    const uint32_t* e_from;  // [n_e] k0 of BetweenFactor
    const uint32_t* e_to;    // [n_e] k1
    const double* meas;      // [n_e][8] prepared measurements
    double huber_delta;      // <= 0: no loss function
    // PriorFactor blocks (prior_factor.rs:96-108): r = to_vector(x_v) - data, 7 rows; J = the first six columns of I7
    int n_prior = 0;
    const uint32_t* prior_v = nullptr;   // [n_prior] vertex (device order)
    const double* prior_data = nullptr;  // [n_prior][8]: data (7) | the block's Huber delta (<= 0: none)
};

void launch_pg_prepare(int64_t n, const double* poses7, double* posep, hipStream_t s);
// H (tiles, lower triangle) += J^T J over all edges, g += J^T r  (tiles and g zeroed by the caller)
void launch_pg_edges(const PGView& v, const TileMap& tm, double* g, hipStream_t s);
// the prior blocks' J^T J (+= sc^2 on the six diagonal entries of the vertex) and J^T r; after launch_pg_edges
void launch_pg_priors(const PGView& v, const TileMap& tm, double* g, hipStream_t s);
// corrected prior residuals [n_prior][7]
void launch_pg_prior_export(const PGView& v, double* r7_out, hipStream_t s);
void launch_pg_cost(const PGView& v, double* partial, int n_partial, double* out_sumsq, hipStream_t s);
void launch_pg_retract(int64_t n_v, const double* poses, const double* d, double sign, const uint8_t* fix,
                       double* poses_out, hipStream_t s);
void launch_pg_negate(int64_t n, const double* x, double* y, hipStream_t s);
// corrected residuals [n_e][6] and Jacobians [n_e][6][12] in the kernel's edge order
void launch_pg_export(const PGView& v, double* r_out, double* j_out, hipStream_t s);

}  // namespace apex
