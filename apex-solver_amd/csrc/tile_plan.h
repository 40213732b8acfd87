// tile_plan.h -- tile-sparse symmetric positive definite system: structure, factorisation, solves.
//
// Shared by the bundle-adjustment backend (reduced camera matrix S) and the pose-graph backend
// (H = J^T J + lambda I).  A matrix of nt x nt tiles of 144 x 144 doubles, lower triangle only:
//   order()   nested-dissection order of the tile graph (the callers permute their variables by it)
//   build()   symbolic Cholesky fill, elimination-tree levels, slot map, batched task lists, uploads
//   factor() / solve()   level-scheduled tile Cholesky and triangular solves, replayed as hipGraphs
//   pcg()     Jacobi-preconditioned CG on the unfactored tiles (solve_with_pcg, explicit_schur.rs:639-756)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

#include "chol_kernels.h"

namespace apex {

class TilePlan {
   public:
    TilePlan() = default;
    ~TilePlan();
    TilePlan(const TilePlan&) = delete;
    TilePlan& operator=(const TilePlan&) = delete;

    // adj: symmetric nt x nt 0/1 adjacency in the CALLER's tile order.  Returns perm[old] = new.
    // The last tile (it may hold padding rows) stays last.
    static std::vector<int> order(int nt, const std::vector<uint8_t>& adj, bool nested_dissection, int leaf);

    // present: lower-triangular nt x nt 0/1 structure (I >= J) in the FINAL order.
    // Returns "" on success or an error message.
    std::string build(int nt, const std::vector<uint8_t>& present, hipStream_t stream);

    int nt() const { return nt_; }
    int64_t n_pad() const { return (int64_t)nt_ * kNB; }
    int64_t n_slots() const { return n_slots_; }
    int64_t n_touched_slots() const { return n_touched_; }  // tiles non-zero before fill come first
    int n_levels() const { return n_levels_; }
    // tile operations of one factorisation: potrf+inverse, panel products, trailing updates (each 2*144^3 flop for the last two)
    void op_counts(int64_t* potrf, int64_t* trsm, int64_t* upd) const { *potrf = n_potrf_; *trsm = n_trsm_; *upd = n_upd_; }
    double* tiles() const { return tiles_; }
    const int* slot_host() const { return slot_h_.data(); }
    int slot(int I, int J) const { return slot_h_[(size_t)I * nt_ + J]; }
    TileMap tilemap() const { return TileMap{tiles_, slot_, nt_}; }
    const int* diag_slot_dev() const { return diag_slot_; }
    void enable_graphs(bool on) { use_graphs_ = on; }
    void enable_overlap(bool on) { overlap_ = on; }  // before the first factor()
    void set_overlap_min(int n) { overlap_min_ = n; }

    hipError_t zero_tiles();                             // async on the plan's stream
    void add_diag(int n_valid, double add_valid, double pad_value);  // diagonal += / padding rows := value
    void diag(double* out) const;                        // out[n_pad] = diagonal
    void scale_sym(const double* scale);                 // A := D A D on the unfactored tiles, D = diag(scale[n_pad])
    // Cholesky in place; *failed_at = 0 or (tile column + 1) of the first non-positive pivot.  Syncs.
    // With rhs/work (2*n_pad doubles) the forward sweep L y = rhs rides along on a third stream; the next
    // solve(rhs, x, work) with the same pointers then only runs the backward sweep.
    hipError_t factor(int* failed_at, const double* rhs = nullptr, double* work = nullptr);
    void enable_fused_forward(bool on) { fuse_forward_ = on; }
    // x = (L L^T)^-1 rhs ; work: 2*n_pad doubles ; all on the plan's stream, no sync
    void solve(const double* rhs, double* x, double* work);
    // y = A x on the UNFACTORED tiles (deterministic two-pass symmetric product), no sync
    void sym_matvec(const double* x, double* y);
    // Jacobi-PCG on the UNFACTORED tiles; work: 6*n_pad doubles; syncs once per iteration
    hipError_t pcg(const double* rhs, double* x, double* work, int max_iter, double tol, int* iters);

   private:
    void enqueue_factor(const double* rhs, double* work);
    void enqueue_solve(const double* rhs, double* x, double* work, bool backward_only);
    bool run_graph(int which, const double* rhs, double* x, double* work);
    void release();

    int nt_ = 0, n_levels_ = 0;
    int64_t n_slots_ = 0, n_touched_ = 0;
    int64_t n_potrf_ = 0, n_trsm_ = 0, n_upd_ = 0;
    hipStream_t stream_ = nullptr;
    std::vector<int> slot_h_, diag_slot_h_;
    std::vector<int> lv_potrf_, lv_trsm_, lv_fwd_, lv_bwd_, lv_upd_round_, lv_upd_split_;
    hipStream_t side_ = nullptr;  // trailing updates that the next level does not need (enqueue_factor)
    std::vector<hipEvent_t> ev_t_, ev_u2_;
    std::vector<bool> u2_pending_;
    bool overlap_ = true;
    int overlap_min_ = 2;   // U2 batches smaller than this stay on the main stream (swept 1..1024: flat up to 64)
    std::vector<std::pair<int64_t, int64_t>> upd_rounds_;
    double *tiles_ = nullptr, *linv_ = nullptr;
    int *slot_ = nullptr, *diag_slot_ = nullptr, *flag_ = nullptr;
    PotrfTask* potrf_tasks_ = nullptr;
    GemmTask *trsm_tasks_ = nullptr, *upd_tasks_ = nullptr;
    TriTask *tri_fwd_ = nullptr, *tri_bwd_ = nullptr;
    SymTile* sym_tiles_ = nullptr;
    int n_sym_tiles_ = 0;
    int* sym_row_ptr_ = nullptr;
    SymEntry* sym_entries_ = nullptr;
    double *sym_part_ = nullptr, *row_dot_ = nullptr, *blk_part_ = nullptr, *scal_ = nullptr;
    hipGraphExec_t graph_exec_[3] = {nullptr, nullptr, nullptr};
    const double* graph_rhs_[3] = {nullptr, nullptr, nullptr};
    double *graph_x_[3] = {nullptr, nullptr, nullptr}, *graph_work_[3] = {nullptr, nullptr, nullptr};
    bool graph_failed_[3] = {false, false, false};
    hipStream_t fwd_ = nullptr;       // fused forward sweep
    hipEvent_t ev_fwd_ = nullptr;
    const double* fwd_rhs_ = nullptr;  // right-hand side whose forward sweep the last factor() carried
    double* fwd_work_ = nullptr;
    bool fuse_forward_ = false;  // measured: the extra cross-stream edges cost the factorisation more than the sweep saves (+0.3 ms)
    bool use_graphs_ = true;
};

}  // namespace apex
