// capi.cpp -- extern "C" surface declared in include/apexgpu.h.
#include "../../include/apexgpu.h"

#include <string.h>

#include <algorithm>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "ba_structure.h"
#include "pg_solver.h"
#include "solver.h"
#ifdef APEX_WITH_RCCL
#include <rccl/rccl.h>
#include <string.h>
#endif

struct apexgpu_solver {
    apex::Solver* s;
    std::string create_error;
};

static_assert(sizeof(apexgpu_lm_config) == sizeof(apex::LmConfig), "LmConfig layout");
static_assert(sizeof(apexgpu_lm_iter) == sizeof(apex::LmIterRecord), "LmIterRecord layout");
static_assert(sizeof(apexgpu_lm_result) == sizeof(apex::LmResult), "LmResult layout");
static_assert(APEXGPU_NUM_STAGES == apex::kNumStages, "stage count");
static_assert(APEXGPU_PG_NUM_STAGES == apex::kPgNumStages, "pose-graph stage count");

struct apexgpu_pg_solver {
    apex::PoseGraphSolver* s;
};

#define H_OR_FAIL            \
    if (!h || !h->s) return APEXGPU_ERR_INVALID_STATE

// No C++ exception may cross the C boundary (a caller in Rust / C / ctypes cannot unwind it): entry points that
// allocate run their body through this guard.  bad_alloc / length_error (e.g. sizes taken from an untrusted header)
// become APEXGPU_ERR_INVALID_INPUT.
template <typename F>
static int guarded(F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return APEXGPU_ERR_INVALID_INPUT;
    } catch (const std::exception&) {
        return APEXGPU_ERR_INVALID_INPUT;
    } catch (...) {
        return APEXGPU_ERR_INVALID_STATE;
    }
}

// APEX_SEGV_BACKTRACE=1 (debugging aid): the native stack of a crash on stderr (the GPU boxes have no debugger)
#include <execinfo.h>
#include <signal.h>
namespace {
void segv_backtrace(int sig) {
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    raise(sig);
}
void segv_hook() {   // (on an alternate stack: a stack overflow is a SIGSEGV too)
    static const bool on = [] { const char* e = getenv("APEX_SEGV_BACKTRACE"); return e && e[0] == '1'; }();
    if (!on) return;
    static thread_local char alt[1 << 16];
    stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0;
    (void)sigaltstack(&ss, nullptr);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = segv_backtrace; sa.sa_flags = SA_ONSTACK | SA_RESETHAND;
    (void)sigaction(SIGSEGV, &sa, nullptr);
}
}  // namespace

extern "C" {

const char* apexgpu_version(void) { return "apexgpu 0.1 (gfx950)"; }

int apexgpu_create(int64_t n_cam, int64_t n_pt, int64_t n_obs, int mode, int device, apexgpu_solver** out) {
    if (!out) return APEXGPU_ERR_INVALID_INPUT;
    *out = nullptr;
    if (mode < APEXGPU_MODE_BUNDLE_ADJUSTMENT || mode > APEXGPU_MODE_LANDMARKS_AND_INTRINSICS) return APEXGPU_ERR_INVALID_INPUT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return APEXGPU_ERR_DEVICE;
    apexgpu_solver* h = new (std::nothrow) apexgpu_solver();
    if (!h) return APEXGPU_ERR_INVALID_STATE;
    h->s = new (std::nothrow) apex::Solver(n_cam, n_pt, n_obs, mode, device);
    if (!h->s) { delete h; return APEXGPU_ERR_INVALID_STATE; }
    *out = h;
    return APEXGPU_OK;
}

void apexgpu_destroy(apexgpu_solver* h) {
    if (!h) return;
    delete h->s;
    delete h;
}

const char* apexgpu_last_error(const apexgpu_solver* h) { return (h && h->s) ? h->s->last_error() : "invalid handle"; }

int apexgpu_set_structure(apexgpu_solver* h, const uint32_t* cam_idx, const uint32_t* pt_idx, const double* obs_uv,
                          const int64_t* intr_col, const int64_t* pose_col, const int64_t* pt_col,
                          const uint8_t* fix_pose, const uint8_t* fix_intr, const uint8_t* fix_pt, double huber_delta) {
    H_OR_FAIL;
    if (!cam_idx || !pt_idx || !obs_uv || !intr_col || !pose_col || !pt_col) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&] { return h->s->set_structure(cam_idx, pt_idx, obs_uv, intr_col, pose_col, pt_col, fix_pose, fix_intr, fix_pt, huber_delta); });
}
int apexgpu_set_cg_params(apexgpu_solver* h, int max_iterations, double tolerance) {
    H_OR_FAIL;
    h->s->set_cg_params(max_iterations, tolerance);
    return APEXGPU_OK;
}
int apexgpu_set_params(apexgpu_solver* h, const double* poses, const double* intr, const double* points) {
    H_OR_FAIL;
    if (!poses || !intr || !points) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&] { return h->s->set_params(poses, intr, points); });
}
int apexgpu_get_params(apexgpu_solver* h, double* poses, double* intr, double* points) {
    H_OR_FAIL;
    if (!poses || !intr || !points) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&] { return h->s->get_params(poses, intr, points); });
}
int apexgpu_cost(apexgpu_solver* h, double* cost) { H_OR_FAIL; return guarded([&] { return h->s->cost(cost); }); }
int apexgpu_solve_augmented(apexgpu_solver* h, double lambda, int variant, double* step_out, double* grad_out) {
    H_OR_FAIL;
    if (variant != APEXGPU_VARIANT_SPARSE && variant != APEXGPU_VARIANT_ITERATIVE && variant != APEXGPU_VARIANT_IMPLICIT)
        return APEXGPU_ERR_INVALID_INPUT;
    segv_hook();
    return guarded([&] { return h->s->solve_augmented(lambda, variant, step_out, grad_out); });
}
int apexgpu_assemble(apexgpu_solver* h, double lambda) { H_OR_FAIL; return guarded([&] { return h->s->assemble_only(lambda); }); }
int apexgpu_step_stats(apexgpu_solver* h, double out3[3]) { H_OR_FAIL; return h->s->step_stats(out3); }
int apexgpu_eval_step(apexgpu_solver* h, double* trial_cost) { H_OR_FAIL; return h->s->eval_step(trial_cost); }
int apexgpu_commit_step(apexgpu_solver* h) { H_OR_FAIL; return h->s->commit_step(); }
int apexgpu_discard_step(apexgpu_solver* h) { H_OR_FAIL; return h->s->discard_step(); }
int apexgpu_parameter_norm(apexgpu_solver* h, double* out) { H_OR_FAIL; return h->s->parameter_norm(out); }

int apexgpu_column_norms(apexgpu_solver* h, double* norms_out) {
    H_OR_FAIL;
    if (!norms_out) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&] { return h->s->column_norms(norms_out); });
}
int apexgpu_set_column_scaling(apexgpu_solver* h, const double* scaling) { H_OR_FAIL; return guarded([&] { return h->s->set_column_scaling(scaling); }); }

int apexgpu_lm_optimize(apexgpu_solver* h, apexgpu_lm_config* cfg, apexgpu_lm_result* result, apexgpu_lm_iter* history,
                        int history_capacity) {
    H_OR_FAIL;
    if (!cfg || !result) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&] { return h->s->lm_optimize(reinterpret_cast<apex::LmConfig*>(cfg), reinterpret_cast<apex::LmResult*>(result),
                                                  reinterpret_cast<apex::LmIterRecord*>(history), history ? history_capacity : 0); });
}

int apexgpu_get_residual(apexgpu_solver* h, double* r_out) { H_OR_FAIL; return guarded([&] { return h->s->get_residual(r_out); }); }
int apexgpu_get_jacobian_blocks(apexgpu_solver* h, double* jc_out, double* jl_out) {
    H_OR_FAIL;
    return guarded([&] { return h->s->get_jacobian_blocks(jc_out, jl_out); });
}
int apexgpu_get_schur(apexgpu_solver* h, double* S_out, double* gred_out) { H_OR_FAIL; return guarded([&] { return h->s->get_schur(S_out, gred_out); }); }
int apexgpu_get_landmark_blocks(apexgpu_solver* h, double* hinv_out, double* gl_out) {
    H_OR_FAIL;
    return guarded([&] { return h->s->get_landmark_blocks(hinv_out, gl_out); });
}

int apexgpu_get_hessian_csc(apexgpu_solver* h, int64_t* nnz_out, int64_t* colptr_out, int64_t* rowidx_out, double* values_out) {
    H_OR_FAIL;
    return guarded([&] { return h->s->get_hessian_csc(nnz_out, colptr_out, rowidx_out, values_out); });
}

int apexgpu_schur_matvec(apexgpu_solver* h, double lambda, const double* x_in, double* y_explicit, double* y_implicit) {
    H_OR_FAIL;
    if (!x_in) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&] { return h->s->schur_matvec(lambda, x_in, y_explicit, y_implicit); });
}

// Test entry: the `n` handles are the ranks 0..n-1 of one sharded problem (apexgpu_set_shard(r, n) before
// set_structure, same parameters), all on this process's GPU.  Runs ONE distributed Cholesky solve in lockstep --
// every rank executes phase p, then this function plays the communicator for the exchange that follows it (sums in
// rank order, max for the failure flag) -- and leaves the step on every handle (apexgpu_export_step).
static int lockstep_solve_impl(apexgpu_solver** hs, int n, double lambda);
int apexgpu_debug_lockstep_solve(apexgpu_solver** hs, int n, double lambda) {
    return guarded([&] { return lockstep_solve_impl(hs, n, lambda); });
}
}  // extern "C"
static int lockstep_solve_impl(apexgpu_solver** hs, int n, double lambda) {
    if (!hs || n < 2) return APEXGPU_ERR_INVALID_INPUT;
    for (int r = 0; r < n; ++r) if (!hs[r] || !hs[r]->s) return APEXGPU_ERR_INVALID_INPUT;
    for (int phase = 0; phase <= 5; ++phase) {
        for (int r = 0; r < n; ++r) {
            const int rc = hs[r]->s->dist_phase(phase, lambda);
            if (rc != 0) return rc;
            // one rank at a time: the ranks share this GPU, and their stage timers then show what each would take alone
            if (hipDeviceSynchronize() != hipSuccess) return APEXGPU_ERR_DEVICE;
        }
        if (phase == 5) break;
        std::vector<std::vector<apex::Solver::DistBuf>> bufs(n);
        std::vector<int*> flags(n, nullptr);
        for (int r = 0; r < n; ++r) hs[r]->s->dist_buffers(phase, &bufs[r], &flags[r]);
        for (size_t b = 0; b < bufs[0].size(); ++b) {
            const size_t len = bufs[0][b].n;
            const int root = bufs[0][b].root;
            for (int r = 1; r < n; ++r)
                if (bufs[r].size() != bufs[0].size() || bufs[r][b].n != len || bufs[r][b].root != root) return APEXGPU_ERR_INVALID_STATE;
            if (len == 0) continue;
            const int dst = root >= 0 ? root : 0;   // ncclReduce(root) / ncclAllReduce: ranks summed in rank order
            for (int r = 0; r < n; ++r)
                if (r != dst) apex::launch_vec_add((int64_t)len, bufs[dst][b].ptr, bufs[r][b].ptr, bufs[dst][b].ptr, nullptr);
            if (root < 0)
                for (int r = 1; r < n; ++r)
                    if (hipMemcpyAsync(bufs[r][b].ptr, bufs[0][b].ptr, len * sizeof(double), hipMemcpyDeviceToDevice, nullptr) != hipSuccess) return APEXGPU_ERR_DEVICE;
        }
        if (flags[0]) {
            int mx = 0;
            for (int r = 0; r < n; ++r) { int f = 0; if (hipMemcpy(&f, flags[r], sizeof f, hipMemcpyDeviceToHost) != hipSuccess) return APEXGPU_ERR_DEVICE; mx = std::max(mx, f); }
            for (int r = 0; r < n; ++r) if (hipMemcpy(flags[r], &mx, sizeof mx, hipMemcpyHostToDevice) != hipSuccess) return APEXGPU_ERR_DEVICE;
        }
        if (hipDeviceSynchronize() != hipSuccess) return APEXGPU_ERR_DEVICE;
    }
    return APEXGPU_OK;
}
extern "C" {
// Host arithmetic only (no device is touched): the cut of the tile elimination tree that a distributed plan of `world`
// ranks makes for the lower-triangular tile structure `present` (nt x nt, row-major, I >= J).  owner_out[nt]: owning
// rank of every tile column, -1 for the shared top; returns the number of top columns (0: the plan stays replicated).
int apexgpu_debug_partition(int nt, const uint8_t* present, int world, int* owner_out) {
    if (nt <= 0 || !present || !owner_out || world < 1) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&]() -> int {
    apex::TilePlan tp;
    tp.set_partition(0, world);
    const std::vector<uint8_t> pr(present, present + (size_t)nt * nt);
    const std::vector<int> owner = tp.preview_owners(nt, pr);
    if (owner.empty()) { for (int i = 0; i < nt; ++i) owner_out[i] = 0; return 0; }
    int n_top = 0;
    for (int i = 0; i < nt; ++i) { owner_out[i] = owner[i]; n_top += owner[i] < 0; }
    return n_top;
    });
}

int apexgpu_debug_check_schedule(int nt, const uint8_t* present, int world, int rank, const int opts[8], int64_t out[8], char* msg, int msg_len) {
    if (nt <= 0 || !present || !opts || !out || world < 1 || rank < 0 || rank >= world) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&]() -> int {
    apex::TilePlan tp;
    if (world > 1) tp.set_partition(rank, world);
    tp.set_two_side(opts[0]);
    tp.enable_overlap(opts[1] != 0); if (opts[1] > 1) tp.set_overlap_min(opts[1]);
    tp.set_split_u1(opts[2]);
    tp.set_gate_min(opts[3]);
    tp.set_factor_flow(opts[4], opts[5]);
    tp.debug_skip_idle_level_wait(opts[6] != 0);
    const std::vector<uint8_t> pr(present, present + (size_t)nt * nt);
    const std::string e = tp.build_host_only(nt, pr);
    std::string first;
    if (!e.empty()) { if (msg && msg_len > 0) snprintf(msg, (size_t)msg_len, "%s", e.c_str()); return APEXGPU_ERR_INVALID_STATE; }
    for (int k = 0; k < 8; ++k) out[k] = 0;
    for (int ph = 0; ph < 2; ++ph) {
        std::vector<apex::SchedOp> ops = tp.schedule_trace(ph);
        if (opts[7] >= 0 && ph == 0) {   // drop the opts[7]-th stream wait of the sequence: the checker must notice when it mattered
            int seen = 0;
            for (size_t i = 0; i < ops.size(); ++i)
                if (ops[i].op == 2 && seen++ == opts[7]) { ops.erase(ops.begin() + (long)i); out[7] = 1; break; }
        }
        std::string why;
        const int bad = tp.check_schedule(ops, &why);
        out[0] += (int64_t)ops.size();
        for (const apex::SchedOp& o : ops) { out[1] += o.op == 0; out[6] += o.op == 2; }
        out[2 + ph] = bad;
        if (bad && first.empty()) first = why;
    }
    out[4] = tp.factor_flow_units(); out[5] = tp.factor_flow_groups();
    if (msg && msg_len > 0) snprintf(msg, (size_t)msg_len, "%s", first.c_str());
    return tp.n_levels();
    });
}

int apexgpu_owned_landmarks(apexgpu_solver* h, uint8_t* mask) {
    H_OR_FAIL;
    if (!mask) return APEXGPU_ERR_INVALID_INPUT;
    return h->s->owned_landmarks(mask);
}
int apexgpu_export_step(apexgpu_solver* h, double* step_out, double* grad_out) { H_OR_FAIL; return guarded([&] { return h->s->export_step(step_out, grad_out); }); }

int apexgpu_set_option(apexgpu_solver* h, const char* name, int value) {
    H_OR_FAIL;
    const std::string n = name ? name : "";
    // Switches that shape what set_structure builds (task lists, tile order, partition) or what the captured hipGraphs
    // hold are rejected once the structure exists: flipping them later would launch kernels over lists that were never built.
    static const char* const structural[] = {"schur_form", "hubs_last", "dist_factor", "tree_sharding", "dist_selftest", "nested_dissection", "update_overlap",
                                             "split_u1", "flood_gate", "two_side", "factor_flow", "factor_flow_rows", "device_pair_list", "matrix_free_only",
                                             "auto_variant", "variant_cost_permille", "max_tile_updates"};
    if (h->s->has_structure())
        for (const char* k : structural)
            if (n == k) return APEXGPU_ERR_INVALID_STATE;
    if (n == "schur_form") {   // 4 the queued pair list (default; nine-column cameras), 3 the pair list with one running block per wave
        if (value != 3 && value != 4) return APEXGPU_ERR_INVALID_INPUT;
        h->s->set_schur_form(value);
    }
    else if (n == "graphs") h->s->enable_graphs(value != 0);
    else if (n == "update_overlap") { h->s->enable_overlap(value != 0); if (value > 1) h->s->set_overlap_min(value); }
    else if (n == "tri_dataflow") h->s->enable_tri_flow(value != 0);
    else if (n == "split_u1") h->s->set_split_u1(value);
    else if (n == "flood_gate") h->s->set_gate_min(value);
    else if (n == "two_side") h->s->set_two_side(value);
    else if (n == "matrix_free_only") h->s->set_matrix_free_only(value != 0);
    else if (n == "auto_variant") h->s->set_auto_variant(value != 0);
    else if (n == "variant_cost_permille") h->s->set_variant_cost_permille(value);
    else if (n == "device_pair_list") h->s->set_device_pair_recs(value != 0);
    else if (n == "eager_step_eval") h->s->set_eager_step_eval(value != 0);
    else if (n == "one_wait") h->s->set_one_wait(value != 0);
    else if (n == "max_tile_updates") h->s->set_max_tile_updates(value);
    else if (n == "factor_flow") h->s->set_factor_flow(value, 0);          // max columns per level group inside the dataflow launch (0: off)
    else if (n == "factor_flow_rows") h->s->set_factor_flow(h->s->plan().factor_flow_cols(), value);
    else if (n == "debug_poison_sweep") h->s->debug_poison_next_solve(value);   /* tests: 1 / 2 = the next solve's forward / backward dataflow sweep times out */
    else if (n == "debug_poison_factor") h->s->debug_poison_next_factor();       /* tests: the next factorisation's dataflow launch times out */
    else if (n == "debug_occupy_cus") return h->s->debug_occupy_cus(value, 40000);   /* tests: block `value` CUs for 40 ms, starting now */
    else if (n == "hubs_last") h->s->set_hubs_last(value != 0);
    else if (n == "dist_factor") h->s->set_dist_factor(value != 0);
    else if (n == "tree_sharding") h->s->set_tree_sharding(value != 0);
    else if (n == "dist_selftest") h->s->set_dist_selftest(value);
    else if (n == "nested_dissection") h->s->set_nd(value != 0, value > 1 ? value : 0);  /* value > 1: leaf size */
    else return APEXGPU_ERR_INVALID_INPUT;
    return APEXGPU_OK;
}
int apexgpu_enable_stage_timing(apexgpu_solver* h, int on) {
    H_OR_FAIL;
    if (on > 1) h->s->enable_stage_timing_only((uint32_t)on >> 1);   // bit k + 1 of `on`: stage k alone is timed
    else h->s->enable_stage_timing(on != 0);
    return APEXGPU_OK;
}
int apexgpu_reset_stage_times(apexgpu_solver* h) { H_OR_FAIL; h->s->reset_stage_times(); return APEXGPU_OK; }
int apexgpu_stage_times(apexgpu_solver* h, double ms[APEXGPU_NUM_STAGES], int64_t calls[APEXGPU_NUM_STAGES]) {
    H_OR_FAIL;
    h->s->stage_times(ms, calls);
    return APEXGPU_OK;
}
int apexgpu_setup_times(apexgpu_solver* h, double seconds[6], double counts[4]) {
    H_OR_FAIL;
    if (!seconds) return APEXGPU_ERR_INVALID_INPUT;
    for (int k = 0; k < 6; ++k) seconds[k] = h->s->setup_seconds()[k];
    if (counts) { counts[0] = h->s->n_hubs(); counts[1] = h->s->pair_blocks(); counts[2] = h->s->pair_slots(); counts[3] = h->s->schur_form(); }
    return APEXGPU_OK;
}
int apexgpu_info(apexgpu_solver* h, double info[16]) {
    H_OR_FAIL;
    info[0] = h->s->n_tile_rows(); info[1] = (double)h->s->tile_count(); info[2] = h->s->schur_scatter_pairs();
    info[3] = (double)h->s->cam_dof_internal(); info[4] = h->s->last_reg(); info[5] = h->s->last_pcg_iters();
    info[6] = h->s->touched_tiles(); info[7] = h->s->local_obs();
    info[8] = h->s->n_levels();
    int64_t a = 0, b = 0, c = 0;
    h->s->plan().op_counts(&a, &b, &c);
    info[9] = (double)a; info[10] = (double)b; info[11] = (double)c;
    info[12] = h->s->dist_top_columns(); info[13] = h->s->dist_local_fraction();
    info[14] = h->s->tree_sharded() ? 1.0 : 0.0;
    info[15] = h->s->schur_form();
    return APEXGPU_OK;
}

int apexgpu_debug_get_pair_records(apexgpu_solver* h, uint32_t* recs4_out, int64_t cap_slots) {
    H_OR_FAIL;
    if (!recs4_out) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&] { return h->s->get_pair_records(recs4_out, cap_slots); });
}
int apexgpu_trim_host_cache(int64_t* released_bytes) {
    const size_t n = apex::HostBlockCache::get().trim();
    if (released_bytes) *released_bytes = (int64_t)n;
    return APEXGPU_OK;
}
int apexgpu_variant_costs(apexgpu_solver* h, double out[4]) {
    H_OR_FAIL;
    if (!out) return APEXGPU_ERR_INVALID_INPUT;
    h->s->variant_costs(out);
    return APEXGPU_OK;
}
int64_t apexgpu_host_cache_bytes(void) { return (int64_t)apex::HostBlockCache::get().kept_bytes(); }
int apexgpu_variant_info(apexgpu_solver* h, int asked_variant, int* used_variant, char* reason, int reason_len) {
    H_OR_FAIL;
    if (asked_variant != APEXGPU_VARIANT_SPARSE && asked_variant != APEXGPU_VARIANT_ITERATIVE && asked_variant != APEXGPU_VARIANT_IMPLICIT)
        return APEXGPU_ERR_INVALID_INPUT;
    if (used_variant) *used_variant = h->s->variant_used(asked_variant);
    if (reason && reason_len > 0) {
        const std::string& r = h->s->variant_reason();
        const size_t n = std::min<size_t>(r.size(), (size_t)reason_len - 1);
        memcpy(reason, r.data(), n);
        reason[n] = 0;
    }
    return APEXGPU_OK;
}

int apexgpu_counters(apexgpu_solver* h, int64_t out[4]) {
    H_OR_FAIL;
    if (!out) return APEXGPU_ERR_INVALID_INPUT;
    out[0] = h->s->sweep_timeouts(); out[1] = h->s->plan().tri_flow() ? 1 : 0; out[2] = h->s->factor_flow_timeouts(); out[3] = h->s->plan().factor_flow_groups();
    return APEXGPU_OK;
}

int apexgpu_get_unique_id(void* out128) {
#ifdef APEX_WITH_RCCL
    if (!out128) return APEXGPU_ERR_INVALID_INPUT;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return APEXGPU_ERR_DEVICE;
    memcpy(out128, &id, 128);
    return APEXGPU_OK;
#else
    (void)out128;
    return APEXGPU_ERR_INVALID_STATE;
#endif
}
int apexgpu_comm_init(apexgpu_solver* h, int world, int rank, const void* unique_id128) {
    H_OR_FAIL;
    if (!unique_id128) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&] { return h->s->comm_init(world, rank, unique_id128); });
}
int apexgpu_comm_init_shm(apexgpu_solver* h, int world, int rank, const char* name) {
    H_OR_FAIL;
    if (!name) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&] { return h->s->comm_init_shm(world, rank, name); });
}
int apexgpu_shard_range(int64_t n_pt, int64_t n_obs, const uint32_t* pt_idx, int rank, int world, int64_t* lo, int64_t* hi) {
    if (!pt_idx || !lo || !hi || world < 1 || rank < 0 || rank >= world || n_pt <= 0) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&]() -> int {
        std::vector<int64_t> ptr(n_pt + 1, 0);
        for (int64_t i = 0; i < n_obs; ++i) {
            if (pt_idx[i] >= (uint64_t)n_pt) return APEXGPU_ERR_INVALID_INPUT;
            ptr[pt_idx[i] + 1]++;
        }
        for (int64_t l = 0; l < n_pt; ++l) ptr[l + 1] += ptr[l];
        apex::shard_range(n_pt, ptr.data(), rank, world, lo, hi);
        return APEXGPU_OK;
    });
}
int apexgpu_set_shard(apexgpu_solver* h, int rank, int world) { H_OR_FAIL; return h->s->set_shard(rank, world); }

// Host arithmetic only (no device is touched): the sorted camera-pair lists k_schur_pairs consumes (csrc/schur_pairs.h)
// for the observation list (cam_idx, pt_idx), with the caller's camera order and a dense tile map (slot of tile (I, J),
// I >= J, = I (I + 1) / 2 + J).  Two-call pattern: with every output NULL the sizes come back in counts[4] = {slots,
// chunks, blocks, tasks}; o_index_out[n_obs] receives the landmark-major position -> caller's observation index map that
// the records' i / j refer to.  Lets CPU tests replay the kernel's bookkeeping (block boundaries, padding, flush points).
static int debug_pair_lists_impl(int64_t n_cam, int64_t n_pt, int64_t n_obs, int dc, const uint32_t* cam_idx, const uint32_t* pt_idx,
                                 int64_t counts[4], uint32_t* recs4_out, int32_t* chunks2_out, int64_t* blocks4_out,
                                 int32_t* tasks2_out, int32_t* o_index_out, bool queued, int64_t* qdesc3_out) {
    if (n_cam <= 0 || n_pt <= 0 || n_obs < 0 || (dc != 6 && dc != 9) || !cam_idx || !pt_idx || !counts) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&]() -> int {
        for (int64_t i = 0; i < n_obs; ++i)
            if (cam_idx[i] >= (uint64_t)n_cam || pt_idx[i] >= (uint64_t)n_pt) return APEXGPU_ERR_INVALID_INPUT;
        std::vector<int> pt_ptr(n_pt + 1, 0), order(n_obs);
        for (int64_t i = 0; i < n_obs; ++i) pt_ptr[pt_idx[i] + 1]++;
        for (int64_t l = 0; l < n_pt; ++l) pt_ptr[l + 1] += pt_ptr[l];
        { std::vector<int> fill(pt_ptr.begin(), pt_ptr.end() - 1); for (int64_t i = 0; i < n_obs; ++i) order[fill[pt_idx[i]]++] = (int)i; }
        for (int64_t l = 0; l < n_pt; ++l)
            std::stable_sort(order.begin() + pt_ptr[l], order.begin() + pt_ptr[l + 1], [&](int a, int b) { return cam_idx[a] < cam_idx[b]; });
        std::vector<uint32_t> o_cam(n_obs), o_pt(n_obs);
        for (int64_t k = 0; k < n_obs; ++k) { o_cam[k] = cam_idx[order[k]]; o_pt[k] = pt_idx[order[k]]; }
        std::vector<int> cam_ptr(n_cam + 1, 0), cam_obs(n_obs);
        for (int64_t k = 0; k < n_obs; ++k) cam_ptr[o_cam[k] + 1]++;
        for (int64_t c = 0; c < n_cam; ++c) cam_ptr[c + 1] += cam_ptr[c];
        { std::vector<int> fill(cam_ptr.begin(), cam_ptr.end() - 1); for (int64_t k = 0; k < n_obs; ++k) cam_obs[fill[o_cam[k]]++] = (int)k; }
        const int nt = (int)((n_cam * dc + apex::kNB - 1) / apex::kNB);
        std::vector<int> slot((size_t)nt * nt, -1);
        for (int I = 0; I < nt; ++I) for (int J = 0; J <= I; ++J) slot[(size_t)I * nt + J] = I * (I + 1) / 2 + J;
        std::vector<int> ext(n_cam);
        for (int64_t c = 0; c < n_cam; ++c) ext[c] = (int)c;
        apex::PairLists pl;
        apex::build_pair_lists(dc, nt, slot.data(), n_cam, ext.data(), o_cam.data(), o_pt.data(), pt_ptr.data(), cam_ptr.data(), cam_obs.data(), &pl, 0, queued);
        counts[0] = (int64_t)pl.recs.size(); counts[1] = (int64_t)pl.chunks.size(); counts[2] = (int64_t)pl.blocks.size(); counts[3] = (int64_t)pl.tasks.size();
        if (recs4_out) memcpy(recs4_out, pl.recs.data(), pl.recs.size() * sizeof(apex::PairRec));
        if (chunks2_out) memcpy(chunks2_out, pl.chunks.data(), pl.chunks.size() * sizeof(apex::PairChunk));
        if (blocks4_out)
            for (size_t b = 0; b < pl.blocks.size(); ++b) {
                blocks4_out[4 * b] = pl.blocks[b].dst; blocks4_out[4 * b + 1] = pl.blocks[b].ci; blocks4_out[4 * b + 2] = pl.blocks[b].cj;
                blocks4_out[4 * b + 3] = pl.blocks[b].flags;
            }
        if (tasks2_out) memcpy(tasks2_out, pl.tasks.data(), pl.tasks.size() * sizeof(apex::PairTask));
        if (o_index_out) for (int64_t k = 0; k < n_obs; ++k) o_index_out[k] = order[k];
        if (qdesc3_out)
            for (size_t q = 0; q < pl.qdesc.size(); ++q) { qdesc3_out[3 * q] = pl.qdesc[q].dst; qdesc3_out[3 * q + 1] = pl.qdesc[q].cj; qdesc3_out[3 * q + 2] = pl.qdesc[q].flags; }
        return APEXGPU_OK;
    });
}
int apexgpu_debug_pair_lists(int64_t n_cam, int64_t n_pt, int64_t n_obs, int dc, const uint32_t* cam_idx, const uint32_t* pt_idx,
                             int64_t counts[4], uint32_t* recs4_out, int32_t* chunks2_out, int64_t* blocks4_out,
                             int32_t* tasks2_out, int32_t* o_index_out) {
    return debug_pair_lists_impl(n_cam, n_pt, n_obs, dc, cam_idx, pt_idx, counts, recs4_out, chunks2_out, blocks4_out, tasks2_out, o_index_out, false, nullptr);
}
// The same in the QUEUED layout ("schur_form" 4, d_c = 9; csrc/schur_pairs.h): qdesc3_out[8 * chunks][3] = {dst, cj, flags} of
// every (chunk, queue) -- entry 7 of a chunk: cj = the row's camera --, chunks2_out[.][0] = the chunk's flush bits.
int apexgpu_debug_pair_lists_queued(int64_t n_cam, int64_t n_pt, int64_t n_obs, const uint32_t* cam_idx, const uint32_t* pt_idx,
                                    int64_t counts[4], uint32_t* recs4_out, int32_t* chunks2_out, int64_t* blocks4_out,
                                    int32_t* tasks2_out, int32_t* o_index_out, int64_t* qdesc3_out) {
    return debug_pair_lists_impl(n_cam, n_pt, n_obs, 9, cam_idx, pt_idx, counts, recs4_out, chunks2_out, blocks4_out, tasks2_out, o_index_out, true, qdesc3_out);
}
// ... and for either camera width (round 5): dc = 9 as above; dc = 6: sixteen queues of four pairs per chunk, qdesc3_out[17 * chunks][3],
// entry 16 of a chunk = the row's camera
int apexgpu_debug_pair_lists_queued_dc(int64_t n_cam, int64_t n_pt, int64_t n_obs, int dc, const uint32_t* cam_idx, const uint32_t* pt_idx,
                                       int64_t counts[4], uint32_t* recs4_out, int32_t* chunks2_out, int64_t* blocks4_out,
                                       int32_t* tasks2_out, int32_t* o_index_out, int64_t* qdesc3_out) {
    return debug_pair_lists_impl(n_cam, n_pt, n_obs, dc, cam_idx, pt_idx, counts, recs4_out, chunks2_out, blocks4_out, tasks2_out, o_index_out, true, qdesc3_out);
}

// Host arithmetic only: everything apexgpu_set_structure derives from the observation list before it touches the device
// (csrc/ba_structure.h) -- camera order with hub cameras last and nested dissection of the tile graph, symbolic fill,
// partition of the elimination tree, landmark sharding, the lists of the Schur reduction -- for rank `rank` of `world`.
// opts[5] = {nested_dissection (0 off, 1 on, > 1 leaf size), hubs_last, dist_factor, tree_sharding, schur_form}.
// stats_out[16] = {tile rows, hub cameras, border tiles, tiles S touches, tiles after fill, elimination-tree levels,
// shared top columns, tree sharded, seconds: order+structure, sharding+lists, (unused), Schur lists, (unused), total,
// pair contributions incl. self pairs, camera-pair blocks}.  cmap_out[n_cam] (caller's camera -> internal), owned_out[n_pt]
// (1: this rank assembles the landmark), tile_owner_out[tile rows] (-1: shared top / not distributed) may be NULL.
int apexgpu_debug_host_structure(int64_t n_cam, int64_t n_pt, int64_t n_obs, int mode, const uint32_t* cam_idx,
                                 const uint32_t* pt_idx, int rank, int world, const int opts[5], double stats_out[16],
                                 int32_t* cmap_out, uint8_t* owned_out, int32_t* tile_owner_out) {
    if (n_cam <= 0 || n_pt <= 0 || n_obs < 0 || !cam_idx || !pt_idx || !opts || !stats_out || world < 1 || rank < 0 || rank >= world)
        return APEXGPU_ERR_INVALID_INPUT;
    if (mode < APEXGPU_MODE_BUNDLE_ADJUSTMENT || mode > APEXGPU_MODE_LANDMARKS_AND_INTRINSICS) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&]() -> int {
        for (int64_t i = 0; i < n_obs; ++i)
            if (cam_idx[i] >= (uint64_t)n_cam || pt_idx[i] >= (uint64_t)n_pt) return APEXGPU_ERR_INVALID_INPUT;
        apex::BaStructOptions so;
        so.dc = (apex::mode_mask(mode) & 1) ? 9 : 6;
        so.use_nd = opts[0] != 0; if (opts[0] > 1) so.nd_leaf = opts[0];
        so.hubs_last = opts[1] != 0; so.dist_factor = opts[2] != 0; so.tree_sharding = opts[3] != 0; so.schur_form = opts[4];
        so.rank = rank; so.world = world;
        apex::TilePlan tp;
        apex::BaHostStructure hs;
        std::vector<double> uv(2 * (size_t)n_obs, 0.0);
        const std::string e = hs.build_lists(n_cam, n_pt, n_obs, cam_idx, pt_idx, uv.data(), so, tp);
        if (!e.empty()) return APEXGPU_ERR_INVALID_INPUT;
        tp.build_symbolic(hs.nt, hs.present);
        hs.build_schur_lists(so, tp.slot_host());
        const double t_all = hs.seconds[0] + hs.seconds[1] + hs.seconds[3];
        const double st[16] = {(double)hs.nt, (double)hs.n_hubs, (double)hs.n_border_tiles, (double)hs.n_present, (double)tp.n_slots(),
                               (double)tp.n_levels(), (double)tp.n_top_columns(), hs.tree_shard ? 1.0 : 0.0, hs.seconds[0], hs.seconds[1],
                               0.0, hs.seconds[3], 0.0, t_all, (double)hs.n_pairs, (double)hs.pl.n_blocks};
        for (int k = 0; k < 16; ++k) stats_out[k] = st[k];
        if (cmap_out) for (int64_t c = 0; c < n_cam; ++c) cmap_out[c] = hs.cmap[c];
        if (owned_out) for (int64_t l = 0; l < n_pt; ++l) owned_out[l] = (hs.lmap[l] >= hs.lm_lo && hs.lmap[l] < hs.lm_hi) ? 1 : 0;
        if (tile_owner_out) {
            const std::vector<int> ow = tp.preview_owners(hs.nt, hs.present);
            for (int t = 0; t < hs.nt; ++t) tile_owner_out[t] = ow.empty() ? -1 : ow[t];
        }
        return APEXGPU_OK;
    });
}

// Parity probe: the device's eigenvalue-gated 3x3 inverse (invert_landmark_blocks_with_lambda with lambda = 0,
// explicit_schur.rs:365-442) applied to n caller-supplied symmetric blocks on GPU `device`.
int apexgpu_debug_invert_blocks(int device, int64_t n, const double* blocks9, double* inv9_out, int32_t* ok_out) {
    if (n < 0 || (n > 0 && (!blocks9 || !inv9_out || !ok_out))) return APEXGPU_ERR_INVALID_INPUT;
    if (n == 0) return APEXGPU_OK;
    if (hipSetDevice(device) != hipSuccess) return APEXGPU_ERR_DEVICE;
    double *din = nullptr, *dout = nullptr;
    int* dok = nullptr;
    int rc = APEXGPU_ERR_DEVICE;
    if (hipMalloc((void**)&din, 9 * n * sizeof(double)) == hipSuccess && hipMalloc((void**)&dout, 9 * n * sizeof(double)) == hipSuccess &&
        hipMalloc((void**)&dok, n * sizeof(int)) == hipSuccess &&
        hipMemcpy(din, blocks9, 9 * n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess) {
        apex::launch_debug_invert_blocks(n, din, dout, dok, nullptr);
        if (hipMemcpy(inv9_out, dout, 9 * n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(ok_out, dok, n * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess)
            rc = APEXGPU_OK;
    }
    if (din) (void)hipFree(din);
    if (dout) (void)hipFree(dout);
    if (dok) (void)hipFree(dok);
    return rc;
}


/* ---- SE3 pose-graph backend --------------------------------------------------------------------- */
#define PG_OR_FAIL \
    if (!h || !h->s) return APEXGPU_ERR_INVALID_STATE

int apexgpu_pg_create(int64_t n_vertices, int64_t n_edges, int device, apexgpu_pg_solver** out) {
    if (!out) return APEXGPU_ERR_INVALID_INPUT;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return APEXGPU_ERR_DEVICE;
    apexgpu_pg_solver* h = new (std::nothrow) apexgpu_pg_solver();
    if (!h) return APEXGPU_ERR_INVALID_STATE;
    h->s = new (std::nothrow) apex::PoseGraphSolver(n_vertices, n_edges, device);
    if (!h->s) { delete h; return APEXGPU_ERR_INVALID_STATE; }
    *out = h;
    return APEXGPU_OK;
}
void apexgpu_pg_destroy(apexgpu_pg_solver* h) {
    if (!h) return;
    delete h->s;
    delete h;
}
const char* apexgpu_pg_last_error(const apexgpu_pg_solver* h) { return (h && h->s) ? h->s->last_error() : "invalid handle"; }
int apexgpu_pg_set_structure(apexgpu_pg_solver* h, const uint32_t* e_from, const uint32_t* e_to, const double* meas7,
                             const int64_t* pose_col, const uint8_t* fix6, double huber_delta) {
    PG_OR_FAIL;
    if (!e_from || !e_to || !meas7 || !pose_col) return APEXGPU_ERR_INVALID_INPUT;
    return guarded([&] { return h->s->set_structure(e_from, e_to, meas7, pose_col, fix6, huber_delta); });
}
int apexgpu_pg_set_params(apexgpu_pg_solver* h, const double* poses7) {
    PG_OR_FAIL;
    if (!poses7) return APEXGPU_ERR_INVALID_INPUT;
    return h->s->set_params(poses7);
}
int apexgpu_pg_get_params(apexgpu_pg_solver* h, double* poses7) {
    PG_OR_FAIL;
    if (!poses7) return APEXGPU_ERR_INVALID_INPUT;
    return h->s->get_params(poses7);
}
int apexgpu_pg_cost(apexgpu_pg_solver* h, double* cost) { PG_OR_FAIL; return h->s->cost(cost); }
int apexgpu_pg_solve_augmented(apexgpu_pg_solver* h, double lambda, double* step_out, double* grad_out) {
    PG_OR_FAIL;
    return guarded([&] { return h->s->solve_augmented(lambda, 0, step_out, grad_out); });
}
int apexgpu_pg_step_stats(apexgpu_pg_solver* h, double out3[3]) { PG_OR_FAIL; return h->s->step_stats(out3); }
int apexgpu_pg_eval_step(apexgpu_pg_solver* h, double* trial_cost) { PG_OR_FAIL; return h->s->eval_step(trial_cost); }
int apexgpu_pg_commit_step(apexgpu_pg_solver* h) { PG_OR_FAIL; return h->s->commit_step(); }
int apexgpu_pg_discard_step(apexgpu_pg_solver* h) { PG_OR_FAIL; return h->s->discard_step(); }
int apexgpu_pg_parameter_norm(apexgpu_pg_solver* h, double* out) { PG_OR_FAIL; return h->s->parameter_norm(out); }
int apexgpu_pg_column_norms(apexgpu_pg_solver* h, double* norms_out) {
    PG_OR_FAIL;
    if (!norms_out) return APEXGPU_ERR_INVALID_INPUT;
    return h->s->column_norms(norms_out);
}
int apexgpu_pg_set_column_scaling(apexgpu_pg_solver* h, const double* scaling) { PG_OR_FAIL; return h->s->set_column_scaling(scaling); }

int apexgpu_pg_lm_optimize(apexgpu_pg_solver* h, apexgpu_lm_config* cfg, apexgpu_lm_result* result, apexgpu_lm_iter* history,
                           int history_capacity) {
    PG_OR_FAIL;
    if (!cfg || !result) return APEXGPU_ERR_INVALID_INPUT;
    return h->s->lm_optimize(reinterpret_cast<apex::LmConfig*>(cfg), reinterpret_cast<apex::LmResult*>(result),
                             reinterpret_cast<apex::LmIterRecord*>(history), history ? history_capacity : 0);
}
int apexgpu_pg_get_residual(apexgpu_pg_solver* h, double* r_out) { PG_OR_FAIL; return h->s->get_residual(r_out); }
int apexgpu_pg_get_jacobian_blocks(apexgpu_pg_solver* h, double* j_out) { PG_OR_FAIL; return h->s->get_jacobian_blocks(j_out); }
int apexgpu_pg_set_priors(apexgpu_pg_solver* h, int64_t n, const uint32_t* vertex, const double* data7, const double* huber_delta) {
    PG_OR_FAIL;
    if (n > 0 && (!vertex || !data7)) return APEXGPU_ERR_INVALID_INPUT;
    return h->s->set_priors(n, vertex, data7, huber_delta);
}
int apexgpu_pg_get_prior_residual(apexgpu_pg_solver* h, double* r7_out) {
    PG_OR_FAIL;
    return h->s->get_prior_residual(r7_out);
}
int apexgpu_pg_get_hessian(apexgpu_pg_solver* h, double lambda, double* H_out, double* g_out) {
    PG_OR_FAIL;
    return h->s->get_hessian(lambda, H_out, g_out);
}
int apexgpu_pg_set_option(apexgpu_pg_solver* h, const char* name, int value) {
    PG_OR_FAIL;
    const std::string n = name ? name : "";
    if (n == "graphs") h->s->enable_graphs(value != 0);
    else if (n == "update_overlap") { h->s->enable_overlap(value != 0); if (value > 1) h->s->set_overlap_min(value); }
    else if (n == "tri_dataflow") h->s->enable_tri_flow(value != 0);
    else if (n == "two_side") h->s->set_two_side(value);
    else if (n == "factor_flow") h->s->set_factor_flow(value, 0);
    else if (n == "factor_flow_rows") h->s->set_factor_flow(h->s->plan().factor_flow_cols(), value);
    else if (n == "flood_gate") h->s->set_gate_min(value);
    else if (n == "split_u1") h->s->set_split_u1(value);
    else if (n == "one_wait") h->s->set_one_wait(value != 0);
    else if (n == "eager_step_eval") h->s->set_eager_step_eval(value != 0);
    else if (n == "nested_dissection") h->s->set_nd(value != 0, value > 1 ? value : 0);
    else if (n == "debug_poison_sweep") h->s->debug_poison_next_solve(value);
    else if (n == "debug_poison_factor") h->s->debug_poison_next_factor();
    else return APEXGPU_ERR_INVALID_INPUT;
    return APEXGPU_OK;
}
int apexgpu_pg_enable_stage_timing(apexgpu_pg_solver* h, int on) {
    PG_OR_FAIL;
    if (on > 1) h->s->enable_stage_timing_only((uint32_t)on >> 1);   // bit k + 1 of `on`: stage k alone is timed
    else h->s->enable_stage_timing(on != 0);
    return APEXGPU_OK;
}
int apexgpu_pg_reset_stage_times(apexgpu_pg_solver* h) { PG_OR_FAIL; h->s->reset_stage_times(); return APEXGPU_OK; }
int apexgpu_pg_stage_times(apexgpu_pg_solver* h, double ms[APEXGPU_PG_NUM_STAGES], int64_t calls[APEXGPU_PG_NUM_STAGES]) {
    PG_OR_FAIL;
    h->s->stage_times(ms, calls);
    return APEXGPU_OK;
}
int apexgpu_pg_counters(apexgpu_pg_solver* h, int64_t out[4]) {
    PG_OR_FAIL;
    if (!out) return APEXGPU_ERR_INVALID_INPUT;
    out[0] = h->s->sweep_timeouts(); out[1] = h->s->plan().tri_flow() ? 1 : 0; out[2] = h->s->factor_flow_timeouts(); out[3] = h->s->plan().factor_flow_groups();
    return APEXGPU_OK;
}
int apexgpu_pg_info(apexgpu_pg_solver* h, double info[8]) {
    PG_OR_FAIL;
    info[0] = h->s->n_tile_rows(); info[1] = (double)h->s->tile_count(); info[2] = (double)h->s->touched_tiles();
    info[3] = h->s->n_levels(); info[4] = 6.0 * (double)h->s->n_vertices();
    int64_t a = 0, b = 0, c = 0;
    h->s->plan().op_counts(&a, &b, &c);
    info[5] = (double)a; info[6] = (double)b; info[7] = (double)c;
    return APEXGPU_OK;
}

}  // extern "C"
