// tile_plan.h -- tile-sparse symmetric positive definite system: structure, factorisation, solves.
//
// Shared by the bundle-adjustment backend (reduced camera matrix S) and the pose-graph backend
// (H = J^T J + lambda I).  A matrix of nt x nt tiles of 144 x 144 doubles, lower triangle only:
//   order()   nested-dissection order of the tile graph (the callers permute their variables by it)
//   build()   symbolic Cholesky fill, elimination-tree levels, slot map, batched task lists, uploads
//   factor() / solve()   level-scheduled tile Cholesky and triangular solves, replayed as hipGraphs
//   pcg()     Jacobi-preconditioned CG on the unfactored tiles (solve_with_pcg, explicit_schur.rs:639-756)
//
// Distributed factorisation (set_partition(rank, world) before build()): the elimination tree is cut below its top
// separators into `world` groups of independent subtrees.  A rank factorises the columns of ITS subtrees only (the
// "local" levels), the updates every rank adds to the shared top tiles are summed in ONE exchange, and the few top
// columns -- latency-bound, a small fraction of the flops -- are factorised redundantly by every rank.  The
// triangular solves follow the same pattern: local forward sweep, one n_pad-vector exchange for the top blocks,
// replicated top sweeps, local backward sweep, one n_pad-vector exchange that assembles x on every rank.
// factor()/solve() run the phases and call the communicator hooks in between; factor_phase()/solve_phase() expose the
// same phases so that several instances can be driven in lockstep inside one process (tests).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "chol_kernels.h"

namespace apex {

// One call of the factorisation's launch sequence, as TilePlan::schedule_trace records it instead of issuing it.
struct SchedOp {
    int op;             // 0 launch, 1 event record, 2 stream waits for event
    uintptr_t stream;   // the stream the call goes to
    uintptr_t event;    // record / wait: the event
    int list;           // launch: 0 potrf, 1 panel solves, 2 updates, 3 the dataflow launch
    int64_t first;      // launch: first task (unit) of its list
    int count;          // ... and how many
};

class TilePlan {
   public:
    TilePlan() = default;
    ~TilePlan();
    TilePlan(const TilePlan&) = delete;
    TilePlan& operator=(const TilePlan&) = delete;

    // adj: symmetric nt x nt 0/1 adjacency in the CALLER's tile order.  Returns perm[old] = new.
    // The last n_fixed_last tiles keep their places (the last tile may hold padding rows; a bundle-adjustment problem
    // also parks its hub cameras there): they are eliminated last and left out of the dissection.
    static std::vector<int> order(int nt, const std::vector<uint8_t>& adj, bool nested_dissection, int leaf, int n_fixed_last = 1);

    // present: lower-triangular nt x nt 0/1 structure (I >= J) in the FINAL order.
    // Returns "" on success or an error message.
    std::string build(int nt, const std::vector<uint8_t>& present, hipStream_t stream);
    // build() without a device (tests): the same lists and decisions on made-up addresses; inspect with schedule_trace /
    // check_schedule / flow_units_host.  Nothing on such a plan may be launched.
    std::string build_host_only(int nt, const std::vector<uint8_t>& present);
    // The launch sequence of one factorisation phase (0: the local level groups / everything, 1: the shared top of a
    // distributed plan) exactly as enqueue_factor issues it, recorded instead of issued.
    std::vector<SchedOp> schedule_trace(int phase);
    // Proves a recorded sequence race free: stream order + event edges give a happens-before relation; every two launches
    // that touch one tile, at least one of them writing it, must be ordered by it, and no two tasks of one launch may write
    // one tile (or one read what another writes).  Returns the number of violations (0 = proven) and describes the first.
    int check_schedule(const std::vector<SchedOp>& ops, std::string* first_violation) const;
    const std::vector<FactorUnit>& flow_units_host() const { return flow_units_h_; }
    void debug_skip_idle_level_wait(bool on) { debug_skip_idle_wait_ = on; }
    // The host half of build() alone: symbolic fill, partition, slot map, level count (slot_host(), n_slots(),
    // n_touched_slots(), n_levels(), op_counts() are valid afterwards; nothing is allocated on a device).
    void build_symbolic(int nt, const std::vector<uint8_t>& present);

    // ---- distributed factorisation ----
    struct Comm {  // in-place reductions over the ranks, enqueued on `stream`; false = the collective failed
        std::function<bool(double* buf, size_t n, hipStream_t stream)> sum;
        std::function<bool(int* buf, size_t n, hipStream_t stream)> max_int;
    };
    void set_partition(int rank, int world) { part_rank_ = rank; part_world_ = world; }  // before build()
    // self-test: cut the tree for `world` ranks but let THIS rank own every subtree -- the distributed schedule
    // (local levels, top levels, phased sweeps) then runs complete on one rank with no-op exchanges
    void set_own_all(bool on) { own_all_ = on; }
    void set_comm(Comm c) { comm_ = std::move(c); }
    std::vector<int> preview_owners(int nt, const std::vector<uint8_t>& present);  // after set_partition, before build
    bool distributed() const { return n_local_groups_ < n_levels_; }
    int n_top_columns() const { return n_top_cols_; }
    double local_work_fraction() const { return local_frac_; }  // this rank's share of the tile operations below the top
    // tiles every rank owns a copy of after the matrix all-reduce: [0, n_reduce_slots()); in a distributed plan the
    // top tiles are left out (they are summed after the local factorisation instead), otherwise = n_touched_slots()
    int64_t n_reduce_slots() const { return distributed() ? n_t_nt_ : n_touched_; }
    // ... and inside that range the tiles of rank o's columns are contiguous: [first, first + count).  The local phase
    // of rank o reads no other rank's columns, so a reduce to their owner (half the traffic of an all-reduce) is enough.
    int part_world() const { return part_world_; }
    std::pair<int64_t, int64_t> owner_slot_range(int o) const { return own_range_[o]; }
    // the slot ranges [first, count) summed after the local phase (touched top tiles, fill top tiles)
    void top_slot_ranges(std::pair<int64_t, int64_t> out[2]) const;
    void factor_phase(int phase);                  // 0: local levels, 1: top levels (after the top tiles were summed)
    int* flag_dev() const { return flag_; }       // first failed tile column + 1 (max over the ranks after the factorisation)
    // 0: local forward sweep, top blocks of the right-hand side packed into exch_buffer() [sum it over the ranks];
    // 1: top sweeps + local backward sweep, this rank's blocks of x packed into exch_buffer() [sum it]; 2: x := buffer
    void solve_phase(int phase, const double* rhs, double* x, double* work);
    double* exch_buffer() const { return exch_; }

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
    void set_two_side(int mode) { two_side_ = mode; }   // 0 off, 1 by plan size (default), 2 always (tests); before build()
    void set_gate_min(int n) { gate_min_ = n; }   // flood gate in front of U2 batches of at least n tasks (0: off); before the first factor()
    // The top of the elimination tree as one dataflow launch (k_factor_flow): the trailing level groups of a phase whose
    // groups have at most max_cols columns each, every column with at most max_rows off-diagonal tiles.  0 columns: off;
    // < 0 (default): where the launch starts is chosen by a cost model.  Before build().
    void set_factor_flow(int max_cols, int max_rows) { flow_cols_ = max_cols; if (max_rows > 0) flow_rows_ = max_rows; }
    // level groups inside the dataflow launches THAT RUN: none once a launch has timed out (flow_on_)
    int factor_flow_groups() const { return !flow_on_ ? 0 : (flow_g1_[0] - flow_g0_[0]) + (flow_g1_[1] - flow_g0_[1]); }
    int factor_flow_cols() const { return flow_cols_; }
    double factor_flow_sim_us() const { return flow_sim_us_[0] + flow_sim_us_[1]; }
    int factor_flow_units() const { return flow_n_[0] + flow_n_[1]; }
    // A dataflow factorisation whose waits ran into their spin limit leaves the tiles half updated: factor() reports it here
    // (once) and the plan goes back to the level launches for good; the caller re-assembles and factorises again.
    // tools/flow_bench: per-unit stamps of the next factorisations (dispatched, inputs ready, done; 100 MHz) + the unit list
    hipError_t enable_flow_trace();
    hipError_t read_flow_trace(std::vector<FactorUnit>* units, std::vector<unsigned long long>* stamps);
    bool refused_too_large() const { return refused_ == 1; }
    bool refused_no_memory() const { return refused_ == 2; }
    bool refused_by_cost() const { return refused_ == 3; }
    // Predicted milliseconds of one factorisation + both sweeps of a plan with these operation counts on one MI355X: the tile
    // products at the rate the factorisation sustains end to end on the headline shape (0.251 TFLOP in 6.5 ms = 38-40 TF/s, DESIGN
    // section 5; panel products count 45 / 81, a diagonal tile's Cholesky + inverse a third of a product), the sweeps at two
    // passes over the tiles of L at 4.2 TB/s, 30 us of dependent launches per elimination-tree level.  Host arithmetic on the
    // structure: every rank of a distributed plan arrives at the same number.
    static double predict_solve_ms(int64_t n_potrf, int64_t n_trsm, int64_t n_upd, int64_t n_tiles, int n_levels) {
        const double prod = (double)n_upd + (double)n_trsm * (45.0 / 81.0) + (double)n_potrf / 3.0;
        return prod * (2.0 * kNB * kNB * kNB) / 40e12 * 1e3 + 2.0 * (double)n_tiles * kNB * kNB * 8.0 / 4.2e12 * 1e3 + 0.03 * n_levels;
    }
    double predicted_ms() const { return predicted_ms_; }          // of the structure the last build() saw (also when it refused)
    void set_cost_limit_ms(double ms) { cost_limit_ms_ = ms; }     // before build(); <= 0: no limit
    void set_max_updates(int64_t n) { max_updates_ = n > 0 ? n : 80000000LL; }   // (tests lower it to force the refusal on a small problem)
    bool factor_flow_gave_up() { const bool g = flow_gave_up_; flow_gave_up_ = false; return g; }
    void set_split_u1(int min_tasks) { split_u1_ = min_tasks > 0; if (min_tasks > 0) split_u1_min_ = min_tasks; }   // before the first factor()
    hipError_t read_flags(int* failed_at);   // pivot flag of the last factorisation (syncs)
    void enable_tri_flow(bool on);   // triangular sweeps as one dataflow launch each (default) or level by level
    bool tri_flow() const { return tri_flow_; }
    // The dataflow sweeps bound their waits (chol_kernels.hip, flow_wait): a sweep that gave up leaves a WRONG x and raises
    // an error word, which solve() posts to pinned host memory behind the sweeps (in a distributed plan after a max over
    // the ranks, so that every rank takes the same decision).  Valid once the plan's stream has been synchronised behind
    // solve(); reading clears it.  The caller repeats that solve with enable_tri_flow(false).
    bool sweep_timed_out();
    bool sweep_timed_out_peek() const { return flow_err_host_ && flow_err_host_[0] != 0; }   // the same word, not cleared
    int sweep_timeouts() const { return n_sweep_timeouts_; }
    // tests only: the next solve()'s forward (1) / backward (2) dataflow sweep runs into its spin limit on purpose
    void debug_poison_next_solve(int which) { poison_ = which; }
    // tests only: the next factorisation's dataflow launch cannot finish (one version counter is made unreachable)
    void debug_poison_next_factor() { poison_factor_ = true; }
    // tests only: block n_cus compute units (all of their LDS) for `micros`, starting now, on a stream of their own;
    // returns once the blocking workgroups are resident (or after 200 ms)
    hipError_t debug_occupy_cus(int n_cus, int micros);

    // async on the plan's stream.  own_touched_only: (distributed plans) this rank adds to the tiles of its own columns and
    // of the shared top only -- tree-sharded landmarks; the other ranks' tiles are then left alone
    // skip_fill (round 5): the assembly is for the Cholesky factorisation of a single-GPU plan whose first writers are flagged
    // (first_writers_flagged()): the fill tiles are not cleared -- their first update does not read them
    hipError_t zero_tiles(bool own_touched_only = false, hipStream_t on = nullptr /* nullptr: the plan's stream */, bool skip_fill = false);
    bool first_writers_flagged() const { return first_ok_; }
    void add_diag(int n_valid, double add_valid, double pad_value);  // diagonal += / padding rows := value
    void diag(double* out) const;                        // out[n_pad] = diagonal
    void scale_sym(const double* scale);                 // A := D A D on the unfactored tiles, D = diag(scale[n_pad])
    // Cholesky in place; *failed_at = 0 or (tile column + 1) of the first non-positive pivot.  Syncs.
    // defer_flags: do not wait for the pivot flag (single-rank plans only): the caller enqueues the sweeps behind the
    // factorisation and calls read_flags() at its own synchronisation point (Solver::solve_augmented: one host wait per solve)
    hipError_t factor(int* failed_at, bool defer_flags = false);
    // x = (L L^T)^-1 rhs ; work: 2*n_pad doubles ; all on the plan's stream, no sync.  hipErrorUnknown: a collective of
    // the distributed sweeps failed (the communicator's own message is with the caller)
    hipError_t solve(const double* rhs, double* x, double* work);
    // y = A x on the UNFACTORED tiles (deterministic two-pass symmetric product), no sync
    void sym_matvec(const double* x, double* y);
    // Jacobi-PCG on the UNFACTORED tiles; work: 6*n_pad doubles; syncs once per iteration
    hipError_t pcg(const double* rhs, double* x, double* work, int max_iter, double tol, int* iters);

   private:
    std::vector<std::vector<int>> symbolic_slots(const std::vector<uint8_t>& present);
    void enqueue_factor(int g0, int g1);
    void enqueue_solve(const double* rhs, double* x, double* work);
    void launch_fwd_group(int lv, double* bvec, double* yvec, hipStream_t s);
    void enqueue_dist_solve(int phase, const double* rhs, double* x, double* work);
    bool run_graph(int which, const double* rhs, double* x, double* work);
    void release();
    void partition_columns(const std::vector<std::vector<int>>& col_rows);

    int nt_ = 0, n_levels_ = 0;   // n_levels_: number of level GROUPS (local groups first, then the top groups)
    int n_local_groups_ = 0, n_top_cols_ = 0;
    int part_rank_ = 0, part_world_ = 1;
    bool own_all_ = false;
    double local_frac_ = 1.0;
    std::vector<int> cls_h_;      // per tile column: 0 another rank's, 1 this rank's, 2 top (shared)
    std::vector<int> owner_h_;    // per tile column: owning rank, -1 top
    std::vector<std::pair<int64_t, int64_t>> own_range_;  // per rank: slots of the touched tiles of its columns
    std::vector<std::pair<int64_t, int64_t>> own_fill_;   // per rank: slots of the fill tiles of its columns
    int* cls_ = nullptr;
    double* exch_ = nullptr;
    Comm comm_;
    int64_t n_t_nt_ = 0, n_f_nt_ = 0;   // slot order: touched non-top | touched top | fill non-top | fill top
    int64_t n_slots_ = 0, n_touched_ = 0;
    int64_t n_potrf_ = 0, n_trsm_ = 0, n_upd_ = 0;
    hipStream_t stream_ = nullptr;
    std::vector<int> slot_h_, diag_slot_h_;
    std::vector<int> lv_potrf_, lv_trsm_, lv_fwd_, lv_bwd_, lv_upd_round_, lv_upd_split_, lv_upd_splitd_, lv_upd_splita_, lv_upd_splitb_;
    std::vector<std::vector<int>> fwd_cut_;  // per group: first forward task of each column that gets its own launch
    hipStream_t side_ = nullptr;  // trailing updates that the next level does not need (enqueue_factor)
    hipStream_t side2_ = nullptr; // U2b2: the bulk of U2 (targets four levels up and more)
    int two_side_ = 1;            // option; two_side_plan_: what build() decided for this plan
    bool two_side_plan_ = false;
    hipStream_t so_ = nullptr;    // U1o: updates of the next level's off-diagonal tiles, beside its potrf
    std::vector<hipEvent_t> ev_t_, ev_u2_, ev_o_, ev_b_, ev_b2_;   // ev_u2_: after U2a of the level; ev_b_: after its U2b
    std::vector<bool> u2_pending_, o_pending_;
    bool split_u1_ = true;
    int split_u1_min_ = 4;
    bool overlap_ = true;
    int gate_min_ = 256;  // U2 batches of at least this many tasks get the flood gate.  Before U2 was split into U2a / U2b the gate was worth 0.3-0.4 ms on
                          // final-13682 (8.3 -> 7.9, any threshold 2 .. 250); after the split it is neutral there (7.6-7.7 either way), +2-3 % on the
                          // dense fronts of ladybug / venice, -2 % on sphere2500's small batches: kept for the large batches only
    int* gate_cnt_ = nullptr;   // [levels + 1] potrf workgroups that have started, per level
    int overlap_min_ = 2;   // U2 batches smaller than this stay on the main stream (swept 1..1024: flat up to 64)
    std::vector<std::pair<int64_t, int64_t>> upd_rounds_;
    double *tiles_ = nullptr, *linv_ = nullptr;
    int *slot_ = nullptr, *diag_slot_ = nullptr, *flag_ = nullptr;
    PotrfTask* potrf_tasks_ = nullptr;
    GemmTask *trsm_tasks_ = nullptr, *upd_tasks_ = nullptr;
    TriTask *tri_fwd_ = nullptr, *tri_bwd_ = nullptr;
    FlowTask *flow_fwd_ = nullptr, *flow_bwd_ = nullptr;   // dataflow triangular sweeps (single-GPU plans)
    double* flow_part_ = nullptr;                          // one 144-vector per off-diagonal tile
    int* flow_flags_ = nullptr;                            // cnt[nt] | done[nt] | error word
    double* pcg_host_ = nullptr;                           // pinned: two slots of PCG scalars (pcg(): read one iteration behind)
    hipEvent_t pcg_ev_[2] = {nullptr, nullptr};
    int* flow_err_host_dev_ = nullptr;                     // the device address of flow_err_host_ (mapped pinned memory)
    int* flow_err_host_ = nullptr;                         // pinned: [0] the error word behind the last solve(), [1..2] debug_occupy_cus
    int n_sweep_timeouts_ = 0;
    int poison_ = 0;
    bool poison_factor_ = false;
    hipStream_t occ_stream_ = nullptr;
    bool post_sweep_status(bool reduce);   // false: the max-reduction over the ranks failed
    bool dry_run_ = false;
    int refused_ = 0;                       // why the last build() gave up: 1 update list beyond max_updates_, 2 tiles beyond the free memory, 3 predicted cost above cost_limit_ms_
    double predicted_ms_ = 0.0, cost_limit_ms_ = 0.0;
    int64_t max_updates_ = 80000000LL;      // tile products per factorisation a plan may hold (12.7 s at 45 TF/s)
    bool debug_skip_idle_wait_ = false;   // tests only: bring back the round-3 schedule bug (no wait after a level without side-stream work)
    std::vector<SchedOp>* sched_trace_ = nullptr;
    std::vector<PotrfTask> potrf_h_;
    std::vector<GemmTask> trsm_h_, upd_h_;
    std::vector<FactorUnit> flow_units_h_;
    FactorUnit* flow_units_ = nullptr;   // dataflow factorisation of the top groups: [phase 0 units | phase 1 units]
    int* flow_ver_ = nullptr;            // per tile slot: finished strips of in-launch writers
    unsigned long long* flow_trace_ = nullptr;
    int flow_cols_ = -1, flow_rows_ = 24;   // -1: the start of the launch is chosen by a cost model (build())
    int flow_g0_[2] = {0, 0}, flow_g1_[2] = {0, 0};   // per phase (local groups / top groups): the groups inside the launch
    int flow_first_[2] = {0, 0}, flow_n_[2] = {0, 0};
    double flow_sim_us_[2] = {0.0, 0.0};   // makespan of the list schedule that ordered the units (build())
    bool flow_on_ = true, flow_gave_up_ = false;
    // the dataflow triangular sweeps (enable_tri_flow): task counts of the forward / backward launch
    int n_flow_tasks_ = 0, n_flow_bwd_ = 0, n_flow_parts_ = 0;
    int n_flow_local_ = 0;   // distributed plans: the forward tasks of phase 0 (the rest: the top columns, phase 1)
    bool tri_flow_ = true;
    SymTile* sym_tiles_ = nullptr;
    int n_sym_tiles_ = 0;
    int* sym_row_ptr_ = nullptr;
    SymEntry* sym_entries_ = nullptr;
    double *sym_part_ = nullptr, *row_dot_ = nullptr, *blk_part_ = nullptr, *scal_ = nullptr;
    static constexpr int kGraphs = 6;  // 0 factor (local levels), 1 both sweeps, 3 factor (top levels), 4/5 distributed solve phases (2: unused)
    hipGraphExec_t graph_exec_[kGraphs] = {};
    const double* graph_rhs_[kGraphs] = {};
    double *graph_x_[kGraphs] = {}, *graph_work_[kGraphs] = {};
    bool graph_failed_[kGraphs] = {};
    // The FIRST update of every fill tile (a tile of L that is structurally zero in S) is flagged -- bit 0 of GemmTask::C in the
    // level lists, kFlowFirstWriter in the dataflow units -- and does not read its target (beta = 0): the 0.63 GB of fill tiles
    // of final-13682 are then neither cleared before a factorisation nor read by those updates (round 5).
    bool first_ok_ = false;   // (this plan qualifies: not distributed, has fill tiles, no dataflow launch over shared top groups)
    static constexpr int kTriInline = 8;       // (swept 0 / 4 / 8 / 16 / 32 / all: profiles/r05_sweep_tri_inline.txt) the dataflow sweeps: in levels of at most this many columns a block's solve task forms its last-arriving product itself (FlowTask::mat2)
    bool use_graphs_ = true;
};

}  // namespace apex
