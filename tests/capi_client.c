/* A plain-C client of include/apexgpu.h: what a compiled host (the reference's Rust shim)
 * does through the FFI -- no Python, no torch types.  Reads a problem dumped by the test as raw little-endian arrays,
 * runs the device-resident LM loop and prints the result as one JSON line.
 *   capi_client ba <file>   |   capi_client pg <file>   |   capi_client ba-level1 <file>
 * "ba-level1" replays, call by call, what the Rust binding of rust/ makes out of the reference's UNCHANGED loop
 * (optimize_with_mode::<GpuBaMode>): the optimiser owns the variables on the host; every iteration GpuBaMode::assemble
 * uploads them (apexgpu_set_params), solve_augmented_equation is apexgpu_solve_augmented, the gradient / step norms and
 * the predicted reduction are host arithmetic on the returned vectors, apply_parameter_step retracts on the host (the
 * CPU path: ora_se3_plus of the oracle library stands in for apex-manifolds), and the trial cost is the residual
 * evaluation at the trial point (the reference calls Problem::compute_residual_sparse on the CPU; here the device's
 * cost kernel after an upload of the trial point, equal to 1e-13).  Must reproduce apexgpu_lm_optimize ("ba").
 * BA file: i64 n_cam, n_pt, n_obs, mode; u32 cam_idx[n_obs], pt_idx[n_obs]; f64 obs_uv[2 n_obs]; f64 poses[7 n_cam],
 *          intr[3 n_cam], points[3 n_pt]
 * PG file: i64 n_v, n_e; i64 ids[n_v]; u32 e_from[n_e], e_to[n_e]; f64 meas[7 n_e]; f64 poses[7 n_v]            */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "apexgpu.h"

static void* rd(FILE* f, size_t n, size_t sz) {
    void* p = malloc(n * sz + 8);
    if (fread(p, sz, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

static apexgpu_lm_config default_config(int max_iterations) {  /* LevenbergMarquardtConfig::default */
    apexgpu_lm_config c;
    memset(&c, 0, sizeof c);
    c.max_iterations = max_iterations; c.cost_tolerance = 1e-6; c.parameter_tolerance = 1e-8; c.gradient_tolerance = 1e-10;
    c.damping = 1e-3; c.damping_min = 1e-12; c.damping_max = 1e12; c.damping_nu = 2.0; c.trust_region_radius = 1e4;
    c.min_trust_region_radius = 1e-32; c.min_cost_threshold = -1.0; c.timeout_s = -1.0; c.variant = APEXGPU_VARIANT_SPARSE;
    return c;
}

/* oracle/ba_oracle.c (test infrastructure): SE3 right-plus as apply_tangent_step does it on the CPU */
void ora_se3_plus(const double pose[7], const double delta[6], double out[7]);

static double norm2(const double* x, int64_t n) { double s = 0; for (int64_t i = 0; i < n; ++i) s += x[i] * x[i]; return sqrt(s); }

/* apply_parameter_step / apply_negative_parameter_step (src/optimizer/mod.rs:309-356): fixed DOF zeroed first */
static void host_apply(int64_t n_cam, int64_t n_pt, const int64_t* ic, const int64_t* pc, const int64_t* lc, const uint8_t* fixp,
                       const double* step, double sign, int selfcal, double* poses, double* intr, double* pts) {
    for (int64_t c = 0; c < n_cam; ++c) {
        double d[6], o[7];
        for (int a = 0; a < 6; ++a) d[a] = fixp[6 * c + a] ? 0.0 : sign * step[pc[c] + a];
        ora_se3_plus(poses + 7 * c, d, o);
        memcpy(poses + 7 * c, o, sizeof o);
        if (selfcal) for (int a = 0; a < 3; ++a) intr[3 * c + a] += sign * step[ic[c] + a];
    }
    for (int64_t l = 0; l < n_pt; ++l)
        for (int a = 0; a < 3; ++a) pts[3 * l + a] += sign * step[lc[l] + a];
}

static int level1_sequence(apexgpu_solver* h, int64_t n_cam, int64_t n_pt, const int64_t* ic, const int64_t* pc, const int64_t* lc,
                           const uint8_t* fixp, int selfcal, double* poses, double* intr, double* pts, apexgpu_lm_config* cfg,
                           apexgpu_lm_result* res) {
    const int64_t total = 9 * n_cam + 3 * n_pt;
    double* step = malloc(8 * total); double* grad = malloc(8 * total);
    double lambda = cfg->damping, nu = cfg->damping_nu, cur = 0.0;
    int rc = apexgpu_set_params(h, poses, intr, pts);                 /* initialize_optimization_state: cost at the start */
    if (!rc) rc = apexgpu_cost(h, &cur);
    if (rc) return rc;
    memset(res, 0, sizeof *res);
    res->initial_cost = cur;
    int iteration = 0, status = 1 /* OptimizationStatus::MaxIterationsReached (src/optimizer/mod.rs:189-216) */;
    for (;;) {
        rc = apexgpu_set_params(h, poses, intr, pts);                  /* GpuBaMode::assemble: upload state.variables */
        if (!rc) rc = apexgpu_solve_augmented(h, lambda, cfg->variant, step, grad);   /* solve_augmented_equation */
        if (rc) return rc;
        const double gn = norm2(grad, total), sn = norm2(step, total);
        double pred = 0.0;                                               /* compute_predicted_reduction (:721-727) */
        for (int64_t i = 0; i < total; ++i) pred += step[i] * (lambda * step[i] - grad[i]);
        pred *= 0.5;
        host_apply(n_cam, n_pt, ic, pc, lc, fixp, step, 1.0, selfcal, poses, intr, pts);   /* apply_parameter_step */
        double trial = 0.0;
        rc = apexgpu_set_params(h, poses, intr, pts);                  /* compute_residual_sparse at the trial point */
        if (!rc) rc = apexgpu_cost(h, &trial);
        if (rc) return rc;
        const double actual = cur - trial;
        const double rho = (fabs(pred) < 1e-15) ? (actual > 0.0 ? 1.0 : 0.0) : actual / pred;
        int accepted; double reduction = 0.0;
        if (rho > 0.0) {
            const double coff = 2.0 * rho - 1.0, f = 1.0 - coff * coff * coff;
            lambda *= f > 1.0 / 3.0 ? f : 1.0 / 3.0; if (lambda < cfg->damping_min) lambda = cfg->damping_min;
            nu = 2.0; accepted = 1; reduction = cur - trial; cur = trial; res->successful_steps++;
        } else {
            lambda *= nu; nu *= 2.0; if (lambda > cfg->damping_max) lambda = cfg->damping_max;
            accepted = 0; res->unsuccessful_steps++;
            host_apply(n_cam, n_pt, ic, pc, lc, fixp, step, -1.0, selfcal, poses, intr, pts);  /* apply_negative_parameter_step */
        }
        double pn = 0.0;                                                /* compute_parameter_norm (:458-467) */
        { const double a = norm2(poses, 7 * n_cam), b = norm2(intr, 3 * n_cam), c = norm2(pts, 3 * n_pt); pn = sqrt(a * a + b * b + c * c); }
        const double before = accepted ? cur + reduction : cur;
        int st = -1;
        if (iteration >= cfg->max_iterations) st = 1;
        else if (accepted) {
            if (gn < cfg->gradient_tolerance) st = 4;   /* GradientToleranceReached */
            if (st < 0 && iteration > 0) {
                if (sn <= cfg->parameter_tolerance * (pn + cfg->parameter_tolerance)) st = 3;   /* ParameterToleranceReached */
                else if (fabs(before - cur) / (before > 1e-10 ? before : 1e-10) < cfg->cost_tolerance) st = 2;   /* CostToleranceReached */
            }
        }
        ++iteration;
        if (st >= 0) { status = st; break; }
    }
    res->status = status; res->iterations = iteration; res->final_cost = cur;
    free(step); free(grad);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 2;
    apexgpu_lm_result res;
    if (strcmp(argv[1], "ba") == 0 || strcmp(argv[1], "ba-level1") == 0) {
        int64_t hdr[4];
        if (fread(hdr, 8, 4, f) != 4) return 2;
        const int64_t n_cam = hdr[0], n_pt = hdr[1], n_obs = hdr[2];
        uint32_t* cam = rd(f, n_obs, 4); uint32_t* pt = rd(f, n_obs, 4);
        double* uv = rd(f, 2 * n_obs, 8); double* poses = rd(f, 7 * n_cam, 8); double* intr = rd(f, 3 * n_cam, 8);
        double* pts = rd(f, 3 * n_pt, 8);
        int64_t* ic = malloc(8 * n_cam); int64_t* pc = malloc(8 * n_cam); int64_t* lc = malloc(8 * n_pt);
        if (apexgpu_reference_columns(n_cam, n_pt, ic, pc, lc) != 0) return 3;
        uint8_t* fixp = calloc(6 * n_cam, 1);
        memset(fixp, 1, 6); /* pose_0000 fixed (bin/bundle_adjustment.rs) */
        apexgpu_solver* h = NULL;
        int rc = apexgpu_create(n_cam, n_pt, n_obs, (int)hdr[3], 0, &h);
        if (rc) { fprintf(stderr, "create: %d\n", rc); return 3; }
        rc = apexgpu_set_structure(h, cam, pt, uv, ic, pc, lc, fixp, NULL, NULL, 1.0);
        if (!rc) {   /* what the Rust shim logs once per structure: the variant that runs, the two predicted costs behind the choice */
            int used = -1; char why[256]; double costs[4] = {0, 0, -1, 0};
            rc = apexgpu_variant_info(h, APEXGPU_VARIANT_SPARSE, &used, why, (int)sizeof why);
            if (!rc) rc = apexgpu_variant_costs(h, costs);
            if (!rc && (used != APEXGPU_VARIANT_SPARSE || costs[2] != 0.0 || !(costs[0] > 0.0) || !(costs[1] > costs[0]))) {
                fprintf(stderr, "variant %d (%s), predicted %g / %g ms, choice %g\n", used, why, costs[0], costs[1], costs[2]);
                return 5;   /* this small banded problem runs the direct path */
            }
            if (apexgpu_host_cache_bytes() < 0) return 5;
        }
        apexgpu_lm_config cfg = default_config(20);
        if (!rc && strcmp(argv[1], "ba") == 0) {
            rc = apexgpu_set_params(h, poses, intr, pts);
            if (!rc) rc = apexgpu_lm_optimize(h, &cfg, &res, NULL, 0);
        } else if (!rc) {
            rc = level1_sequence(h, n_cam, n_pt, ic, pc, lc, fixp, hdr[3] == 1, poses, intr, pts, &cfg, &res);
        }
        if (rc) { fprintf(stderr, "error %d: %s\n", rc, apexgpu_last_error(h)); return 4; }
        apexgpu_destroy(h);
    } else {
        int64_t hdr[2];
        if (fread(hdr, 8, 2, f) != 2) return 2;
        const int64_t n_v = hdr[0], n_e = hdr[1];
        int64_t* ids = rd(f, n_v, 8); uint32_t* ef = rd(f, n_e, 4); uint32_t* et = rd(f, n_e, 4);
        double* meas = rd(f, 7 * n_e, 8); double* poses = rd(f, 7 * n_v, 8);
        int64_t* col = malloc(8 * n_v);
        if (apexgpu_pose_graph_columns(n_v, ids, col) != 0) return 3;
        uint8_t* fix = calloc(6 * n_v, 1);
        memset(fix, 1, 6);
        apexgpu_pg_solver* h = NULL;
        int rc = apexgpu_pg_create(n_v, n_e, 0, &h);
        if (rc) { fprintf(stderr, "create: %d\n", rc); return 3; }
        rc = apexgpu_pg_set_structure(h, ef, et, meas, col, fix, -1.0);
        if (!rc) rc = apexgpu_pg_set_params(h, poses);
        apexgpu_lm_config cfg = default_config(30);
        if (!rc) rc = apexgpu_pg_lm_optimize(h, &cfg, &res, NULL, 0);
        if (rc) { fprintf(stderr, "error %d: %s\n", rc, apexgpu_pg_last_error(h)); return 4; }
        apexgpu_pg_destroy(h);
    }
    printf("{\"status\": %d, \"iterations\": %d, \"initial_cost\": %.17g, \"final_cost\": %.17g}\n", res.status, res.iterations,
           res.initial_cost, res.final_cost);
    return 0;
}
