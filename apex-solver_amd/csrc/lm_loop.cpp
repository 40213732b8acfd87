// lm_loop.cpp -- see lm_loop.h
#include "lm_loop.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>

namespace apex {

int run_lm(LmBackend& b, LmConfig* cfg, LmResult* res, LmIterRecord* hist, int hist_cap) {
    const auto t0 = std::chrono::steady_clock::now();
    double lambda = cfg->damping, nu = cfg->damping_nu;
    double cur_cost = 0.0;
    int rc = b.cost(&cur_cost);  // initialize_optimization_state (optimizer/mod.rs:550-552)
    if (rc != kOk) return rc;
    memset(res, 0, sizeof *res);
    res->initial_cost = cur_cost;
    res->cost_evaluations = 1;
    // process_jacobian_generic (optimizer/mod.rs:749-763): the scaling is taken from the Jacobian of iteration 0
    // and lives in the optimizer for this optimize() only.
    struct ScalingGuard {
        LmBackend& b; bool on;
        ~ScalingGuard() { if (on) b.set_jacobi_scaling(false); }
    } scaling{b, false};
    if (cfg->use_jacobi_scaling) {
        rc = b.set_jacobi_scaling(true);
        if (rc != kOk) return rc;
        scaling.on = true;
    }
    int iteration = 0, status = kMaxIterationsReached;
    for (;;) {
        rc = b.solve_augmented(lambda, cfg->variant, nullptr, nullptr);  // assemble + compute_step (:861-883)
        res->jacobian_evaluations++;
        if (rc != kOk) { status = kLinearSolveFailed; break; }
        double st[3];
        rc = b.step_stats(st);
        if (rc != kOk) return rc;
        const double gn = st[0], sn = st[1], pred = st[2];
        double new_cost = 0.0;
        rc = b.eval_step(&new_cost);  // evaluate_and_apply_step (:770-817)
        if (rc != kOk) return rc;
        res->cost_evaluations++;
        const double actual = cur_cost - new_cost;  // compute_step_quality (optimizer/mod.rs:668-675)
        const double rho = (fabs(pred) < 1e-15) ? (actual > 0.0 ? 1.0 : 0.0) : actual / pred;
        int accepted;
        double cost_reduction = 0.0;
        if (rho > 0.0) {  // update_damping (:702-717)
            const double coff = 2.0 * rho - 1.0;
            lambda *= std::max(1.0 / 3.0, 1.0 - coff * coff * coff);
            lambda = std::max(lambda, cfg->damping_min);
            nu = 2.0;
            accepted = 1;
            cost_reduction = cur_cost - new_cost;
            cur_cost = new_cost;
            rc = b.commit_step();
            res->successful_steps++;
        } else {
            lambda *= nu;
            nu *= 2.0;
            lambda = std::min(lambda, cfg->damping_max);
            accepted = 0;
            rc = b.discard_step();
            res->unsuccessful_steps++;
        }
        if (rc != kOk) return rc;
        if (hist && iteration < hist_cap) {
            LmIterRecord& h = hist[iteration];
            h.cost = cur_cost; h.damping = lambda; h.rho = rho; h.accepted = accepted; h.gradient_norm = gn;
            h.step_norm = sn; h.predicted_reduction = pred; h.trial_cost = new_cost;
        }
        res->final_gradient_norm = gn;
        res->final_step_norm = sn;
        // check_convergence (optimizer/mod.rs:591-658)
        double pnorm = 0.0;
        rc = b.parameter_norm(&pnorm);
        if (rc != kOk) return rc;
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double cost_before = accepted ? cur_cost + cost_reduction : cur_cost;
        int stt = -1;
        if (!std::isfinite(cur_cost) || !std::isfinite(sn) || !std::isfinite(gn)) stt = kInvalidNumericalValues;
        else if (cfg->timeout_s > 0.0 && elapsed >= cfg->timeout_s) stt = kTimeout;
        else if (iteration >= cfg->max_iterations) stt = kMaxIterationsReached;
        else if (accepted) {
            if (gn < cfg->gradient_tolerance) stt = kGradientToleranceReached;
            if (stt < 0 && iteration > 0) {
                const double rel_step_tol = cfg->parameter_tolerance * (pnorm + cfg->parameter_tolerance);
                if (sn <= rel_step_tol) stt = kParameterToleranceReached;
                else {
                    const double cc = fabs(cost_before - cur_cost);
                    if (cc / std::max(cost_before, 1e-10) < cfg->cost_tolerance) stt = kCostToleranceReached;
                }
            }
            if (stt < 0 && cfg->min_cost_threshold >= 0.0 && cur_cost < cfg->min_cost_threshold) stt = kMinCostThresholdReached;
            if (stt < 0 && cfg->trust_region_radius < cfg->min_trust_region_radius) stt = kTrustRegionRadiusTooSmall;
        }
        if (stt >= 0) { status = stt; ++iteration; break; }
        ++iteration;
    }
    cfg->damping = lambda; cfg->damping_nu = nu;
    res->status = status;
    res->iterations = iteration;
    res->final_cost = cur_cost;
    res->elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return kOk;
}

}  // namespace apex
