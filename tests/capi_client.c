/* A plain-C client of include/apexgpu.h: what a compiled host (the reference's Rust shim)
 * does through the FFI -- no Python, no torch types.  Reads a problem dumped by the test as raw little-endian arrays,
 * runs the device-resident LM loop and prints the result as one JSON line.
 *   capi_client ba <file>   |   capi_client pg <file>
 * BA file: i64 n_cam, n_pt, n_obs, mode; u32 cam_idx[n_obs], pt_idx[n_obs]; f64 obs_uv[2 n_obs]; f64 poses[7 n_cam],
 *          intr[3 n_cam], points[3 n_pt]
 * PG file: i64 n_v, n_e; i64 ids[n_v]; u32 e_from[n_e], e_to[n_e]; f64 meas[7 n_e]; f64 poses[7 n_v]            */
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

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 2;
    apexgpu_lm_result res;
    if (strcmp(argv[1], "ba") == 0) {
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
        if (!rc) rc = apexgpu_set_params(h, poses, intr, pts);
        apexgpu_lm_config cfg = default_config(20);
        if (!rc) rc = apexgpu_lm_optimize(h, &cfg, &res, NULL, 0);
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
