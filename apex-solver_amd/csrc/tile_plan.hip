// tile_plan.hip -- see tile_plan.h
#include "tile_plan.h"
#include "host_parallel.h"

#include <chrono>
#include <thread>

#include <stdlib.h>

#include <math.h>

#include <algorithm>
#include <numeric>

namespace apex {

template <typename T>
static hipError_t dev_alloc(T** p, size_t n) {
    return hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
}
template <typename T>
static hipError_t upload(T** dptr, const std::vector<T>& hv) {
    if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
    hipError_t e = dev_alloc(dptr, hv.size());
    if (e != hipSuccess || hv.empty()) return e;
    return hipMemcpy(*dptr, hv.data(), hv.size() * sizeof(T), hipMemcpyHostToDevice);
}
static hipError_t alloc_zero(double** p, size_t n) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    hipError_t e = dev_alloc(p, n);
    if (e != hipSuccess) return e;
    return hipMemset(*p, 0, std::max<size_t>(n, 1) * sizeof(double));
}

// Nested-dissection order of the nodes of an undirected graph: recursive bisection by BFS level
// structures from a pseudo-peripheral node; the middle level is the separator and is ordered after
// both halves.  Sub-graphs of at most `leaf` nodes (or that a level structure cannot split, e.g. a
// clique) keep their natural order.  Deterministic.
static void nested_dissection(const std::vector<std::vector<int>>& adj, std::vector<int> nodes, std::vector<int>& out,
                              int leaf) {
    std::sort(nodes.begin(), nodes.end());
    if ((int)nodes.size() <= leaf) { out.insert(out.end(), nodes.begin(), nodes.end()); return; }
    const int n = (int)adj.size();
    std::vector<int> mark(n, -1), dist(n, -1);
    for (int v : nodes) mark[v] = 0;
    auto bfs = [&](int src, std::vector<int>& order) {
        for (int v : nodes) dist[v] = -1;
        order.clear();
        order.push_back(src); dist[src] = 0;
        for (size_t h = 0; h < order.size(); ++h)
            for (int w : adj[order[h]])
                if (mark[w] == 0 && dist[w] < 0) { dist[w] = dist[order[h]] + 1; order.push_back(w); }
    };
    std::vector<int> order;
    bfs(nodes[0], order);
    if (order.size() < nodes.size()) {  // disconnected: order the components independently
        std::vector<int> comp(order), rest;
        std::vector<char> in(n, 0);
        for (int v : comp) in[v] = 1;
        for (int v : nodes) if (!in[v]) rest.push_back(v);
        nested_dissection(adj, comp, out, leaf);
        nested_dissection(adj, rest, out, leaf);
        return;
    }
    bfs(order.back(), order);  // from a far node: long, thin level structure
    const int depth = dist[order.back()];
    if (depth < 2) { out.insert(out.end(), nodes.begin(), nodes.end()); return; }
    std::vector<int> cnt(depth + 1, 0);
    for (int v : nodes) cnt[dist[v]]++;
    int best = 1; long bestcost = -1; long below = cnt[0];
    for (int m = 1; m < depth; ++m) {
        const long above = (long)nodes.size() - below - cnt[m];
        const long cost = std::labs(below - above) + 2L * cnt[m];  // balance + separator size
        if (bestcost < 0 || cost < bestcost) { bestcost = cost; best = m; }
        below += cnt[m];
    }
    std::vector<int> A, B, S;
    for (int v : nodes) (dist[v] < best ? A : (dist[v] > best ? B : S)).push_back(v);
    nested_dissection(adj, A, out, leaf);
    nested_dissection(adj, B, out, leaf);
    std::sort(S.begin(), S.end());
    out.insert(out.end(), S.begin(), S.end());
}

std::vector<int> TilePlan::order(int nt, const std::vector<uint8_t>& adjm, bool nd, int leaf, int n_fixed_last) {
    std::vector<int> perm(nt);
    std::iota(perm.begin(), perm.end(), 0);
    const int nf = nt - std::max(1, std::min(n_fixed_last, nt));   // tiles that take part in the dissection
    if (!nd || nf < 23) return perm;
    std::vector<std::vector<int>> adj(nf);
    for (int a = 0; a < nf; ++a)
        for (int b = 0; b < nf; ++b)
            if (a != b && adjm[(size_t)a * nt + b]) adj[a].push_back(b);
    std::vector<int> nodes(nf), ord;
    std::iota(nodes.begin(), nodes.end(), 0);
    nested_dissection(adj, nodes, ord, leaf);
    for (int pos = 0; pos < (int)ord.size(); ++pos) perm[ord[pos]] = pos;
    return perm;
}

void TilePlan::release() {
    if (dry_run_) {   // a host-only plan owns no device memory, streams or events
        tiles_ = linv_ = sym_part_ = row_dot_ = blk_part_ = scal_ = exch_ = nullptr; flag_ = nullptr; gate_cnt_ = nullptr;
        side_ = side2_ = so_ = nullptr;
        ev_t_.clear(); ev_u2_.clear(); ev_o_.clear(); ev_b_.clear(); ev_b2_.clear();
        return;
    }
    void* ptrs[] = {tiles_, linv_, slot_, diag_slot_, flag_, potrf_tasks_, trsm_tasks_, upd_tasks_, tri_fwd_, tri_bwd_,
                    flow_fwd_, flow_bwd_, flow_part_, flow_flags_, flow_units_, flow_ver_, flow_trace_, sym_tiles_, sym_row_ptr_, sym_entries_, sym_part_, row_dot_, blk_part_, scal_, cls_, exch_, gate_cnt_};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    tiles_ = linv_ = sym_part_ = row_dot_ = blk_part_ = scal_ = exch_ = nullptr;
    slot_ = diag_slot_ = flag_ = sym_row_ptr_ = cls_ = nullptr;
    potrf_tasks_ = nullptr; trsm_tasks_ = upd_tasks_ = nullptr; tri_fwd_ = tri_bwd_ = nullptr;
    flow_fwd_ = flow_bwd_ = nullptr; flow_part_ = nullptr; flow_flags_ = nullptr; n_flow_tasks_ = 0;
    flow_units_ = nullptr; flow_ver_ = nullptr; flow_trace_ = nullptr; flow_n_[0] = flow_n_[1] = 0; flow_on_ = true; flow_gave_up_ = false;
    sym_tiles_ = nullptr; sym_entries_ = nullptr;
    gate_cnt_ = nullptr;
    for (int i = 0; i < kGraphs; ++i) {
        if (graph_exec_[i]) { (void)hipGraphExecDestroy(graph_exec_[i]); graph_exec_[i] = nullptr; }
        graph_failed_[i] = false;
    }
    if (flow_err_host_) { (void)hipHostFree(flow_err_host_); flow_err_host_ = nullptr; flow_err_host_dev_ = nullptr; }
    if (pcg_host_) { (void)hipHostFree(pcg_host_); pcg_host_ = nullptr; for (hipEvent_t& ev : pcg_ev_) { if (ev) (void)hipEventDestroy(ev); ev = nullptr; } }
    if (occ_stream_) { (void)hipStreamSynchronize(occ_stream_); (void)hipStreamDestroy(occ_stream_); occ_stream_ = nullptr; }
    for (hipEvent_t e : ev_t_) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_u2_) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_o_) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_b_) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_b2_) (void)hipEventDestroy(e);
    ev_t_.clear(); ev_u2_.clear(); ev_o_.clear(); ev_b_.clear(); ev_b2_.clear();
}

TilePlan::~TilePlan() {
    release();
    if (side_) (void)hipStreamDestroy(side_);
    if (side2_) { (void)hipStreamDestroy(side2_); side2_ = nullptr; }
    if (so_) { (void)hipStreamDestroy(so_); so_ = nullptr; }
}

// Cut the elimination tree into part_world_ groups of subtrees plus a shared top.  Deterministic: every rank
// computes the same cut.  Starting from the roots, the heaviest subtree is split (its root joins the top, its
// children become subtrees) until a longest-processing-time assignment of the subtrees balances within 8 %.
void TilePlan::partition_columns(const std::vector<std::vector<int>>& col_rows) {
    cls_h_.assign(nt_, 1);
    owner_h_.assign(nt_, 0);
    n_top_cols_ = 0; local_frac_ = 1.0;
    if (part_world_ <= 1) return;
    const int N = part_world_;
    std::vector<int> parent(nt_, -1);
    std::vector<std::vector<int>> children(nt_);
    std::vector<double> sub(nt_, 0.0);
    for (int K = 0; K < nt_; ++K) {
        const double m = (double)col_rows[K].size();
        sub[K] += 1.0 + m + 0.5 * m * (m + 1.0);   // potrf + panel products + trailing updates of column K
        if (!col_rows[K].empty()) {
            parent[K] = col_rows[K][0];
            children[parent[K]].push_back(K);
            sub[parent[K]] += sub[K];               // parent > K: its subtree sum is complete before it is read
        }
    }
    std::vector<int> S;
    for (int K = 0; K < nt_; ++K) if (parent[K] < 0) S.push_back(K);
    std::vector<char> top(nt_, 0);
    std::vector<int> owner_of_root;
    auto lpt = [&](const std::vector<int>& roots, std::vector<int>* assign) {
        std::vector<int> idx(roots.size());
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return sub[roots[a]] > sub[roots[b]]; });
        std::vector<double> load(N, 0.0);
        if (assign) assign->assign(roots.size(), 0);
        for (int i : idx) {
            const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            load[r] += sub[roots[i]];
            if (assign) (*assign)[i] = r;
        }
        return load;
    };
    // Walk down the tree (always splitting the heaviest subtree) and keep the cut with the smallest estimated
    // critical path: the most loaded rank's subtrees plus the replicated top, whose columns are latency-bound
    // (three dependent launches each, ~200 tile products' worth) and run at a fraction of the batched rate.
    std::vector<double> own_w(nt_);
    for (int K = 0; K < nt_; ++K) { const double m = (double)col_rows[K].size(); own_w[K] = 1.0 + m + 0.5 * m * (m + 1.0); }
    int n_top = 0;
    double top_cost = 0.0, best_cost = -1.0;
    std::vector<int> best_S;
    std::vector<char> best_top;
    int best_ntop = 0;
    for (;;) {
        if ((int)S.size() >= N && n_top > 0) {
            const std::vector<double> load = lpt(S, nullptr);
            const double cost = *std::max_element(load.begin(), load.end()) + top_cost;
            if (best_cost < 0.0 || cost < best_cost) { best_cost = cost; best_S = S; best_top = top; best_ntop = n_top; }
        }
        int best = -1;
        for (int i = 0; i < (int)S.size(); ++i)
            if (!children[S[i]].empty() && (best < 0 || sub[S[i]] > sub[S[best]])) best = i;
        if (best < 0 || n_top + 1 > nt_ / 2) break;
        const int R = S[best];
        top[R] = 1; ++n_top;
        top_cost += std::max(3.0 * own_w[R], 200.0);
        S.erase(S.begin() + best);
        S.insert(S.end(), children[R].begin(), children[R].end());
        std::sort(S.begin(), S.end());
    }
    if (best_cost < 0.0) return;  // nothing to share (a forest, or no cut with a subtree per rank): replicated factorisation
    S = best_S; top = best_top; n_top = best_ntop;
    if (n_top == 0) return;  // nothing shared (a forest that balances as it is): keep the replicated factorisation
    std::vector<int> assign;
    const std::vector<double> load = lpt(S, &assign);
    std::vector<int> owner(nt_, -1);
    for (size_t i = 0; i < S.size(); ++i) owner[S[i]] = assign[i];
    for (int K = nt_ - 1; K >= 0; --K)
        if (!top[K] && owner[K] < 0) owner[K] = owner[parent[K]];
    double sum = 0.0;
    for (double l : load) sum += l;
    local_frac_ = sum > 0.0 ? load[part_rank_] / sum : 0.0;
    const bool own_all = own_all_;  // self-test: one rank plays every owner (the two-phase schedule without exchanges)
    for (int K = 0; K < nt_; ++K) cls_h_[K] = top[K] ? 2 : ((owner[K] == part_rank_ || own_all) ? 1 : 0);
    for (int K = 0; K < nt_; ++K) owner_h_[K] = top[K] ? -1 : owner[K];
    n_top_cols_ = n_top;
}

// symbolic Cholesky at tile granularity: struct(L_K) \ {parent} merges into the parent column
static std::vector<std::vector<int>> symbolic_fill(int nt, const std::vector<uint8_t>& present) {
    std::vector<std::vector<int>> col_rows(nt);
    for (int K = 0; K < nt; ++K)
        for (int I = K + 1; I < nt; ++I)
            if (present[(size_t)I * nt + K]) col_rows[K].push_back(I);
    for (int K = 0; K < nt; ++K) {
        auto& rows = col_rows[K];
        if (rows.size() < 2) continue;
        const int parent = rows[0];
        std::vector<int> merged;
        std::set_union(col_rows[parent].begin(), col_rows[parent].end(), rows.begin() + 1, rows.end(),
                       std::back_inserter(merged));
        col_rows[parent].swap(merged);
    }
    return col_rows;
}

// The owner rank of every tile column (-1: shared top) that build() will arrive at for the same structure and
// partition; empty when the plan will not be distributed.  Host arithmetic only.
std::vector<int> TilePlan::preview_owners(int nt, const std::vector<uint8_t>& present) {
    const int keep = nt_;
    nt_ = nt;
    partition_columns(symbolic_fill(nt, present));
    nt_ = keep;
    return n_top_cols_ > 0 ? owner_h_ : std::vector<int>();
}

// host half of build(): symbolic fill, partition, slot map.  Returns the filled column structure.
std::vector<std::vector<int>> TilePlan::symbolic_slots(const std::vector<uint8_t>& present) {
    std::vector<std::vector<int>> col_rows = symbolic_fill(nt_, present);
    // slots: first every tile the matrix itself touches (diagonal + structural non-zeros), then the
    // tiles that exist only because of fill -- a multi-GPU all-reduce then moves the first group only
    // A distributed plan (partition_columns) keeps the tiles of the shared top columns at the end of either group:
    // touched non-top | touched top | fill non-top | fill top.
    partition_columns(col_rows);
    slot_h_.assign((size_t)nt_ * nt_, -1);
    diag_slot_h_.assign(nt_, 0);
    n_slots_ = 0;
    const int n_owner = n_top_cols_ > 0 ? part_world_ : 1;
    own_range_.assign(n_owner, {0, 0});
    for (int pass = 0; pass <= n_owner; ++pass) {   // owners 0..n_owner-1 (their columns contiguous), then the top
        const int64_t first = n_slots_;
        for (int K = 0; K < nt_; ++K) {
            const bool is_top = cls_h_[K] == 2;
            if (pass < n_owner ? (is_top || (n_top_cols_ > 0 && owner_h_[K] != pass)) : !is_top) continue;
            diag_slot_h_[K] = (int)n_slots_;
            slot_h_[(size_t)K * nt_ + K] = (int)n_slots_++;
            for (int I : col_rows[K])
                if (present[(size_t)I * nt_ + K]) slot_h_[(size_t)I * nt_ + K] = (int)n_slots_++;
        }
        if (pass < n_owner) own_range_[pass] = {first, n_slots_ - first};
        if (pass == n_owner - 1) n_t_nt_ = n_slots_;
    }
    n_touched_ = n_slots_;
    own_fill_.assign(n_owner, {0, 0});
    for (int pass = 0; pass <= n_owner; ++pass) {   // the fill tiles in the same order: owner by owner, then the top
        const int64_t first = n_slots_;
        for (int K = 0; K < nt_; ++K) {
            const bool is_top = cls_h_[K] == 2;
            if (pass < n_owner ? (is_top || (n_top_cols_ > 0 && owner_h_[K] != pass)) : !is_top) continue;
            for (int I : col_rows[K])
                if (!present[(size_t)I * nt_ + K]) slot_h_[(size_t)I * nt_ + K] = (int)n_slots_++;
        }
        if (pass < n_owner) own_fill_[pass] = {first, n_slots_ - first};
        if (pass == n_owner - 1) n_f_nt_ = n_slots_;
    }
    n_potrf_ = nt_; n_trsm_ = 0; n_upd_ = 0;
    for (int K = 0; K < nt_; ++K) {
        n_trsm_ += (int64_t)col_rows[K].size();
        n_upd_ += (int64_t)col_rows[K].size() * ((int64_t)col_rows[K].size() + 1) / 2;
    }
    return col_rows;
}

void TilePlan::build_symbolic(int nt, const std::vector<uint8_t>& present) {
    nt_ = nt;
    const std::vector<std::vector<int>> col_rows = symbolic_slots(present);
    std::vector<int> level(nt_, 0);
    for (int K = 0; K < nt_; ++K)
        if (!col_rows[K].empty()) level[col_rows[K][0]] = std::max(level[col_rows[K][0]], level[K] + 1);
    n_levels_ = 1 + *std::max_element(level.begin(), level.end());
    n_local_groups_ = n_levels_;
}

// Host-only twin of build() (tests: no device is touched): the same symbolic work, task lists, dataflow units and schedule
// decisions, with made-up tile addresses and stream / event handles.  The plan it leaves behind can only be inspected
// (schedule_trace, check_schedule, flow units): factor() / solve() on it are undefined.
std::string TilePlan::build_host_only(int nt, const std::vector<uint8_t>& present) {
    dry_run_ = true;
    const std::string e = build(nt, present, reinterpret_cast<hipStream_t>(uintptr_t(0x51)));
    return e;
}

std::string TilePlan::build(int nt, const std::vector<uint8_t>& present, hipStream_t stream) {
    if (dry_run_) { tiles_ = linv_ = nullptr; }   // (fake addresses: never freed)
    release();
    nt_ = nt;
    stream_ = stream;
    SetupTrace ptr_trace;
    const size_t tile_elems = (size_t)kNB * kNB;
    refused_ = 0;
    std::vector<std::vector<int>> col_rows = symbolic_slots(present);
    int64_t n_upd = 0;
    for (int K = 0; K < nt_; ++K) n_upd += (int64_t)col_rows[K].size() * (col_rows[K].size() + 1) / 2;
    // what this plan is predicted to cost per solve (reported whatever follows)
    {
        int n_lv = 1;
        std::vector<int> lvl(nt_, 0);
        for (int K = 0; K < nt_; ++K)
            if (!col_rows[K].empty()) { lvl[col_rows[K][0]] = std::max(lvl[col_rows[K][0]], lvl[K] + 1); n_lv = std::max(n_lv, lvl[col_rows[K][0]] + 1); }
        predicted_ms_ = predict_solve_ms(n_potrf_, n_trsm_, n_upd_, n_slots_, n_lv);
    }
    // (the size rule first: it is host arithmetic on the structure, so every rank of a distributed plan decides alike)
    if (n_upd > max_updates_) { refused_ = 1; return "tile update list too large (" + std::to_string(n_upd) + " tile products per factorisation, limit " + std::to_string(max_updates_) + ")"; }
    // (round 6) ... then the cost rule, the same kind of arithmetic: a caller that owns a cheaper way to the same step (the
    // matrix-free PCG, Solver::set_structure) hands in what that way costs, and a plan predicted to cost more is not built
    if (cost_limit_ms_ > 0.0 && predicted_ms_ > cost_limit_ms_) {
        refused_ = 3;
        char buf[160];
        snprintf(buf, sizeof buf, "predicted cost of the direct factorisation %.1f ms per solve, above the %.1f ms of the alternative", predicted_ms_, cost_limit_ms_);
        return buf;
    }
    if (!dry_run_) {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const double need = (double)(n_slots_ + nt_) * tile_elems * 8.0;
        if (need > 0.9 * (double)free_b) {
            refused_ = 2;
            return "the tile matrix needs " + std::to_string(need / 1e9) + " GB; only " + std::to_string(free_b / 1e9) + " GB free";
        }
    }

#define TP_TRY(expr) do { if (!dry_run_) { hipError_t _e = (expr); if (_e != hipSuccess) return std::string("HIP error in " #expr ": ") + hipGetErrorString(_e); } } while (0)
    ptr_trace.mark("plan: symbolic fill, slots");
    TP_TRY(alloc_zero(&tiles_, (size_t)n_slots_ * tile_elems));
    TP_TRY(alloc_zero(&linv_, (size_t)nt_ * tile_elems));
    ptr_trace.mark("plan: tiles allocated, cleared");
    if (dry_run_) {   // addresses that identify tiles, nothing more
        tiles_ = reinterpret_cast<double*>(uintptr_t(1) << 44);
        linv_ = reinterpret_cast<double*>(uintptr_t(1) << 45);
    }
    TP_TRY(upload(&slot_, slot_h_));
    TP_TRY(upload(&diag_slot_, diag_slot_h_));
    if (flag_ && !dry_run_) (void)hipFree(flag_);
    TP_TRY(dev_alloc(&flag_, 4));
    TP_TRY(hipMemset(flag_, 0, 4 * sizeof(int)));

    // ---- task lists scheduled by elimination-tree LEVEL ------------------------------------------------
    // parent(K) = first off-diagonal row of column K; level = height above the leaves.  Columns of one
    // level are independent: their potrf / panel solves / trailing updates run as ONE batched launch
    // each.  Two columns of a level may update the same ancestor tile: those updates are split into
    // conflict-free rounds (deterministic), one launch per round.
    auto tile_ptr = [&](int I, int J) { return tiles_ + (size_t)slot_h_[(size_t)I * nt_ + J] * tile_elems; };
    auto linv_ptr = [&](int K) { return linv_ + (size_t)K * tile_elems; };
    std::vector<int> level(nt_, 0);
    for (int K = 0; K < nt_; ++K)
        if (!col_rows[K].empty()) level[col_rows[K][0]] = std::max(level[col_rows[K][0]], level[K] + 1);
    // Level GROUPS in execution order: this rank's columns level by level, then the shared top columns level by
    // level (a plan that is not distributed has the first kind only); other ranks' columns get no tasks at all.
    const int n_true_levels = 1 + *std::max_element(level.begin(), level.end());
    std::vector<std::vector<int>> level_cols;
    std::vector<int> group_of(nt_, -1);
    n_local_groups_ = 0;
    for (int want = 1; want <= 2; ++want) {
        for (int lv = 0; lv < n_true_levels; ++lv) {
            std::vector<int> cols;
            for (int K = 0; K < nt_; ++K)
                if (level[K] == lv && cls_h_[K] == want) cols.push_back(K);
            if (cols.empty()) continue;
            for (int K : cols) group_of[K] = (int)level_cols.size();
            level_cols.push_back(std::move(cols));
        }
        if (want == 1) n_local_groups_ = (int)level_cols.size();
    }
    n_levels_ = (int)level_cols.size();
    std::vector<std::vector<int>> row_cols(nt_);
    for (int K = 0; K < nt_; ++K)
        if (cls_h_[K] != 0)
            for (int I : col_rows[K]) row_cols[I].push_back(K);
    std::vector<PotrfTask> potrf;
    std::vector<GemmTask> trsm, upd;
    std::vector<TriTask> tf, tb;
    lv_potrf_.assign(n_levels_ + 1, 0); lv_trsm_.assign(n_levels_ + 1, 0);
    lv_fwd_.assign(n_levels_ + 1, 0); lv_bwd_.assign(n_levels_ + 1, 0);
    fwd_cut_.assign(n_levels_, std::vector<int>());
    lv_upd_round_.assign(n_levels_ + 1, 0);
    lv_upd_split_.assign(n_levels_ + 1, 0);
    lv_upd_splita_.assign(n_levels_ + 1, 0);
    lv_upd_splitb_.assign(n_levels_ + 1, 0);
    lv_upd_splitd_.assign(n_levels_ + 1, 0);
    upd_rounds_.clear();
    upd.reserve(n_upd);
    for (int lv = 0; lv < n_levels_; ++lv) {
        struct U { int64_t key; int K; GemmTask t; };
        std::vector<U> us;
        for (int K : level_cols[lv]) {
            const auto& rows = col_rows[K];
            potrf.push_back({tile_ptr(K, K), linv_ptr(K), K});
            // The top columns of a distributed plan are swept by every rank, and the ranks' copies of the top solution
            // must be BITWISE equal (a rank's own blocks are back-substituted from its copy, the result takes rank 0's;
            // with cond(S) ~ 1e9 a last-bit difference shows up as a 1e-11 residual).  The forward step adds into shared
            // ancestor blocks with atomics, which is order-dependent when two columns of a level run in one launch:
            // top columns therefore get one launch each.
            if (cls_h_[K] == 2) fwd_cut_[lv].push_back((int)tf.size());
            tf.push_back({linv_ptr(K), nullptr, K, -1});
            for (int I : rows) {
                if (group_of[I] == lv + 1) trsm.push_back({tile_ptr(I, K), tile_ptr(I, K), linv_ptr(K)});   // (first: see below)
                tf.push_back({linv_ptr(K), tile_ptr(I, K), K, I});
            }
            for (size_t a = 0; a < rows.size(); ++a)
                for (size_t b = 0; b <= a; ++b)
                    us.push_back({(int64_t)rows[a] * nt_ + rows[b], K, {tile_ptr(rows[a], rows[b]), tile_ptr(rows[a], K), tile_ptr(rows[b], K)}});
        }
        // the panel solves of the level: first the tiles whose ROW belongs to the next level (all that U1d(lv) reads), then the
        // others; by column inside each part
        for (int K : level_cols[lv])
            for (int I : col_rows[K])
                if (group_of[I] != lv + 1) trsm.push_back({tile_ptr(I, K), tile_ptr(I, K), linv_ptr(K)});
        std::stable_sort(us.begin(), us.end(), [](const U& x, const U& y) { return x.key < y.key; });
        // U1d: targets = DIAGONAL tiles of the next level's columns (what its potrf needs);
        // U1o: the other tiles of the next level's columns (what its panel solves need) -- on a third stream, beside the
        //      next potrf;
        // U2: targets further up the tree -- these run on the side stream, overlapped with the next
        // level's potrf and panel solves (see enqueue_factor)
        // U2 itself in two parts: U2a = targets in the columns of level lv+2 -- the only ones the NEXT level's U1 updates also
        // write, so U1(lv+1) waits for U2a(lv) alone -- and U2b = everything higher, which then runs beside them.
        // ... and U2b in two: U2b1 = targets in level lv+3 (all that U2a of the NEXT level collides with), which stays on U2a's
        // stream, and U2b2 = level lv+4 and above, the bulk, on a stream of its own (enqueue_factor).
        for (int part = 0; part < 5; ++part) {
            std::vector<const U*> mine;
            for (const U& u : us) {
                const int tcol = (int)(u.key % nt_), trow = (int)(u.key / nt_);
                const int d = group_of[tcol] - lv;
                const int cls = d == 1 ? (trow == tcol ? 0 : 1) : (d == 2 ? 2 : (d == 3 ? 3 : 4));
                if (cls == part) mine.push_back(&u);
            }
            std::vector<int> round(mine.size(), 0);
            int n_rounds = 0;
            for (size_t i = 0; i < mine.size(); ++i) {
                round[i] = (i > 0 && mine[i]->key == mine[i - 1]->key) ? round[i - 1] + 1 : 0;
                n_rounds = std::max(n_rounds, round[i] + 1);
            }
            for (int r = 0; r < n_rounds; ++r) {
                // inside a round: by source column, so that tasks sharing operand tiles are neighbours
                std::vector<const U*> sel;
                for (size_t i = 0; i < mine.size(); ++i)
                    if (round[i] == r) sel.push_back(mine[i]);
                std::stable_sort(sel.begin(), sel.end(), [](const U* x, const U* y) { return x->K < y->K; });
                const int64_t off = (int64_t)upd.size();
                for (const U* u : sel) upd.push_back(u->t);
                upd_rounds_.push_back({off, (int64_t)upd.size() - off});
            }
            if (part == 0) lv_upd_splitd_[lv] = (int)upd_rounds_.size();
            if (part == 1) lv_upd_split_[lv] = (int)upd_rounds_.size();
            if (part == 2) lv_upd_splita_[lv] = (int)upd_rounds_.size();
            if (part == 3) lv_upd_splitb_[lv] = (int)upd_rounds_.size();
        }
        lv_potrf_[lv + 1] = (int)potrf.size();
        lv_trsm_[lv + 1] = (int)trsm.size();
        lv_fwd_[lv + 1] = (int)tf.size();
        lv_upd_round_[lv + 1] = (int)upd_rounds_.size();
    }
    for (int lv = n_levels_ - 1; lv >= 0; --lv) {  // backward sweep: levels from the root down
        for (int I : level_cols[lv]) {
            tb.push_back({linv_ptr(I), nullptr, I, -1});
            for (int J : row_cols[I]) tb.push_back({linv_ptr(I), tile_ptr(I, J), I, J});
        }
        lv_bwd_[n_levels_ - lv] = (int)tb.size();
    }
    // both sweeps as one dataflow launch each (k_tri_fwd_flow / k_tri_bwd_flow; plans that are not distributed): level
    // by level the solve tasks of the level's blocks, then the product tasks of the tiles those solutions multiply.
    // A block's products own consecutive slots of the partial array, in the order the solve task folds them.
    std::vector<FlowTask> ft, bt;
    n_flow_local_ = 0;
    {
        // forward: slots by block row.  In a distributed plan a shared top row takes products from this rank's columns
        // (phase 0: folded into the exchange vector, no solve) and from top columns (phase 1): the rank's sources get the
        // first slots of the row, the top sources the rest, each in column order -- so the fold of the top sources is
        // the same sequence of additions on every rank (the ranks' copies of the top solution must be bitwise equal).
        std::vector<int> first(nt_ + 1, 0), own_src(nt_, 0);
        std::vector<std::vector<int>> slot_of(nt_);
        for (int K = 0; K < nt_; ++K) {
            first[K + 1] = first[K] + (int)row_cols[K].size();
            for (int J : row_cols[K]) own_src[K] += cls_h_[J] == 1;
            int a = 0, b = own_src[K];
            slot_of[K].reserve(row_cols[K].size());
            for (int J : row_cols[K]) slot_of[K].push_back(cls_h_[J] == 1 ? a++ : b++);
        }
        // Single-GPU plans (kTriInline, round 5): the solve task of a block forms the product of its LAST-ARRIVING source itself
        // (FlowTask::mat2 / src2 / slot2: the source solved latest, i.e. of the highest level forward, of the lowest backward) --
        // the link of the dependency chain loses a flag hop and a trip through memory; that product task leaves the list.
        // Only in the NARROW levels (at most kTriInline columns): where a level is wide the sweeps are bound by HBM and the
        // second tile of a solve task only serialises two products (final-13682 with every block inlined: sweeps 0.71 -> 0.79 ms;
        // ladybug-1723, narrow everywhere: 0.35 -> 0.28).
        const bool inl = kTriInline > 0 && !distributed();
        std::vector<int> fwd_inl(nt_, -1), bwd_inl(nt_, -1);
        if (inl)
            for (int K = 0; K < nt_; ++K) {
                if ((int)level_cols[(size_t)group_of[K]].size() > kTriInline) continue;
                for (int J : row_cols[K]) if (fwd_inl[K] < 0 || group_of[J] >= group_of[fwd_inl[K]]) fwd_inl[K] = J;
                for (int I : col_rows[K]) if (bwd_inl[K] < 0 || group_of[I] < group_of[bwd_inl[K]]) bwd_inl[K] = I;
            }
        auto products_of = [&](int K) {
            for (int I : col_rows[K]) {
                if (fwd_inl[I] == K) continue;   // (formed by the solve task of block I)
                const auto& rc = row_cols[I];
                const int pos = (int)(std::lower_bound(rc.begin(), rc.end(), K) - rc.begin());
                ft.push_back({tile_ptr(I, K), K, I, first[I] + slot_of[I][pos], 0});
            }
        };
        auto fwd_solve = [&](int K) {
            FlowTask t{linv_ptr(K), -1, K, first[K], (int)row_cols[K].size()};
            if (fwd_inl[K] >= 0) {
                const auto& rc = row_cols[K];
                const int pos = (int)(std::lower_bound(rc.begin(), rc.end(), fwd_inl[K]) - rc.begin());
                t.mat2 = tile_ptr(K, fwd_inl[K]); t.src2 = fwd_inl[K]; t.slot2 = slot_of[K][pos];
            }
            return t;
        };
        if (!distributed()) {
            for (int lv = 0; lv < n_levels_; ++lv) {
                for (int K : level_cols[lv]) ft.push_back(fwd_solve(K));
                for (int K : level_cols[lv]) products_of(K);
            }
        } else {
            for (int lv = 0; lv < n_local_groups_; ++lv) {          // phase 0: this rank's columns ...
                for (int K : level_cols[lv]) ft.push_back({linv_ptr(K), -1, K, first[K], (int)row_cols[K].size()});
                for (int K : level_cols[lv]) products_of(K);
            }
            for (int lv = n_local_groups_; lv < n_levels_; ++lv)    // ... and what they add to the shared top blocks
                for (int K : level_cols[lv]) ft.push_back({linv_ptr(K), -2, K, first[K], own_src[K]});
            n_flow_local_ = (int)ft.size();
            for (int lv = n_local_groups_; lv < n_levels_; ++lv) {  // phase 1: the top columns, every rank alike
                for (int K : level_cols[lv])
                    ft.push_back({linv_ptr(K), -1, K, first[K] + own_src[K], (int)row_cols[K].size() - own_src[K]});
                for (int K : level_cols[lv]) products_of(K);
            }
        }
        if (!ft.empty()) {
            for (int K = 0; K < nt_; ++K) first[K + 1] = first[K] + (int)col_rows[K].size();      // backward: by block column
            for (int lv = n_levels_ - 1; lv >= 0; --lv) {
                for (int I : level_cols[lv]) {
                    FlowTask t{linv_ptr(I), -1, I, first[I], (int)col_rows[I].size()};
                    if (bwd_inl[I] >= 0) {
                        const auto& cr = col_rows[I];
                        t.mat2 = tile_ptr(bwd_inl[I], I); t.src2 = bwd_inl[I];
                        t.slot2 = (int)(std::lower_bound(cr.begin(), cr.end(), bwd_inl[I]) - cr.begin());
                    }
                    bt.push_back(t);
                }
                for (int I : level_cols[lv])
                    for (int J : row_cols[I]) {
                        if (bwd_inl[J] == I) continue;   // (formed by the solve task of block J)
                        const auto& cr = col_rows[J];
                        const int pos = (int)(std::lower_bound(cr.begin(), cr.end(), I) - cr.begin());
                        bt.push_back({tile_ptr(I, J), I, J, first[J] + pos, 0});
                    }
            }
        }
        int64_t a = 0, b = 0;
        for (int K = 0; K < nt_; ++K) { a += (int64_t)row_cols[K].size(); b += (int64_t)col_rows[K].size(); }
        n_flow_parts_ = (int)std::max(a, b);
    }
    n_flow_bwd_ = (int)bt.size();
    n_flow_tasks_ = (int)ft.size();
    // symmetric matvec of the PCG variant: only tiles that are non-zero before fill
    std::vector<int> sym_ptr(nt_ + 1, 0);
    std::vector<SymEntry> sym;
    std::vector<SymTile> symt;
    for (int I = 0; I < nt_; ++I) {
        for (int J = 0; J < I; ++J)
            if (present[(size_t)I * nt_ + J]) sym.push_back({slot_h_[(size_t)I * nt_ + J], J, 0});
        sym.push_back({diag_slot_h_[I], I, 2});
        for (int I2 = I + 1; I2 < nt_; ++I2)
            if (present[(size_t)I2 * nt_ + I]) sym.push_back({slot_h_[(size_t)I2 * nt_ + I], I2, 1});
        sym_ptr[I + 1] = (int)sym.size();
        for (int J = 0; J <= I; ++J)
            if (J == I || present[(size_t)I * nt_ + J]) symt.push_back({slot_h_[(size_t)I * nt_ + J], I, J});
    }
    n_potrf_ = (int64_t)potrf.size(); n_trsm_ = (int64_t)trsm.size(); n_upd_ = (int64_t)upd.size();
    // The second side stream (enqueue_factor), for the whole plan or not at all: it pays where a level carries a bulk worth
    // overlapping (final-13682: ~1,000 tile products per level, 7.95 -> 7.5 ms; synthetic-10k 6.4 -> 6.1) and costs where the
    // levels are small and the factorisation is its launch chain (the ladybug / venice shapes, ~100 products per level: one
    // more stream is one more edge per level, 3.1 -> 3.5 ms).
    two_side_plan_ = two_side_ == 2 || (two_side_ == 1 && n_upd_ >= 256 * (int64_t)n_levels_);
    // ---- the trailing level groups of each phase as ONE dataflow launch (k_factor_flow, chol_kernels.hip) ----------------
    // Units in left-looking order: per column of the region the updates into its tiles (per target in source order = the
    // order of the level launches), its potrf, its panel solves; last the updates into tiles whose column is outside
    // the launch (the local phase of a distributed plan adding to the shared top).  `running` replays the version
    // counters: a unit may only wait for what EARLIER units publish (the no-deadlock argument), checked here.
    auto make_flow_units = [&](int gf, int g1, std::vector<FactorUnit>& funits, double* sim_us) -> std::string {
        funits.clear();
        std::vector<int> cols;
        std::vector<char> in_reg(nt_, 0);
        for (int g = gf; g < g1; ++g)
            for (int K : level_cols[g]) { cols.push_back(K); in_reg[K] = 1; }
        auto slot_of = [&](int I, int J) { return slot_h_[(size_t)I * nt_ + J]; };
        std::vector<std::vector<int>> src_of((size_t)n_slots_);
        std::vector<std::pair<int, int>> outside;   // (J, I) of targets whose column is not in the launch
        for (int K : cols) {
            const auto& rows = col_rows[K];
            for (size_t a = 0; a < rows.size(); ++a)
                for (size_t b = 0; b <= a; ++b) {
                    std::vector<int>& v = src_of[(size_t)slot_of(rows[a], rows[b])];
                    if (v.empty() && !in_reg[rows[b]]) outside.push_back({rows[b], rows[a]});
                    v.push_back(K);
                }
        }
        std::sort(outside.begin(), outside.end());
        constexpr int W = kFlowUnitsPerTile;
        std::vector<int> running((size_t)n_slots_, 0);
        bool order_ok = true;
        auto emit = [&](FactorUnit u, int inc) {
            for (int q = 0; q < 3; ++q)
                if (u.wait_flag[q] >= 0 && running[(size_t)u.wait_flag[q]] < u.wait_val[q]) order_ok = false;
            funits.push_back(u);
            running[(size_t)u.pub] += inc;
        };
        auto n_upd_of = [&](int st) { return (int)src_of[(size_t)st].size(); };
        // An update whose target column lies TWO level groups or more above its source column is not on the chain
        // potrf -> panel solves -> updates of the next group's tiles -> potrf: it runs as ONE whole-tile unit (kind 3, the level
        // kernels' rate per CU) instead of nine 48 x 48 units made for latency (round 5; "factor_flow_tile" 0: nine everywhere).
        auto emit_updates = [&](int I, int J) {
            const int st = slot_of(I, J);
            for (int n = 0; n < n_upd_of(st); ++n) {
                const int K = src_of[(size_t)st][n], sa = slot_of(I, K), sb = slot_of(J, K);
                const bool whole = group_of[J] > group_of[K] + 1;
                if (whole) {
                    emit(FactorUnit{tile_ptr(I, J), tile_ptr(I, K), tile_ptr(J, K), {n > 0 ? st : -1, sa, sb},
                                    {W * n, W * (n_upd_of(sa) + 1), W * (n_upd_of(sb) + 1)}, st, 3, 0, 0}, W);
                    continue;
                }
                for (int sp = 0; sp < W; ++sp)
                    emit(FactorUnit{tile_ptr(I, J), tile_ptr(I, K), tile_ptr(J, K), {n > 0 ? st : -1, sa, sb},
                                    {W * n, W * (n_upd_of(sa) + 1), W * (n_upd_of(sb) + 1)}, st, 2, sp, 0}, 1);
            }
        };
        for (int J : cols) {
            const int sd = slot_of(J, J), nd = n_upd_of(sd);
            emit_updates(J, J);
            for (int I : col_rows[J]) emit_updates(I, J);
            emit(FactorUnit{tile_ptr(J, J), linv_ptr(J), nullptr, {nd > 0 ? sd : -1, -1, -1}, {W * nd, 0, 0}, sd, 0, J, 0}, W);
            for (int I : col_rows[J]) {
                const int st = slot_of(I, J), n = n_upd_of(st);
                for (int sp = 0; sp < W; ++sp)
                    emit(FactorUnit{tile_ptr(I, J), tile_ptr(I, J), linv_ptr(J), {n > 0 ? st : -1, -1, sd}, {W * n, 0, W * (nd + 1)}, st, 1, sp, 0}, 1);
            }
        }
        for (const auto& t : outside) emit_updates(t.second, t.first);
        if (!order_ok) return "internal error: a dataflow factorisation unit waits for a later one";
        // ---- dispatch order = the start order of a simulated list schedule -------------------------------------------------
        // Workgroups are dispatched in list order, one per CU: the launch works through a WINDOW of ~256 consecutive units.
        // In plain left-looking order that window fills up with units that wait for the current column while units further
        // down the list -- updates whose sources were finished long ago -- cannot start: the bulk ends up serialised behind
        // the critical chain, and the chain then waits for the bulk (measured: tools/flow_bench).  So the units are listed in
        // the order in which a 240-processor list schedule STARTS them (a unit becomes ready when the versions it waits for
        // are reached; among ready units the one with the longest remaining chain goes first).  A unit starts after its
        // producers finish, hence after they started: still a topological order, re-checked below.
        {
            const int base = 0, n = (int)funits.size();
            auto cost_of = [](const FactorUnit& u) { return u.kind == 0 ? 34.0 : (u.kind == 1 ? 10.0 : (u.kind == 3 ? 30.0 : 8.0)); };   // us, with the hop
            auto inc_of = [](const FactorUnit& u) { return (u.kind == 0 || u.kind == 3) ? W : 1; };
            std::vector<int> writer(n);           // which writer of its tile a unit belongs to
            {
                std::vector<int> cnt((size_t)n_slots_, 0);
                for (int x = 0; x < n; ++x) { const FactorUnit& u = funits[base + x]; writer[x] = cnt[(size_t)u.pub] / W; cnt[(size_t)u.pub] += inc_of(u); }
            }
            // remaining chain (bottom level) through the tile-version nodes (slot, writer)
            std::vector<int> node0((size_t)n_slots_ + 1, 0);
            for (int sl = 0; sl < n_slots_; ++sl) node0[(size_t)sl + 1] = node0[(size_t)sl] + n_upd_of(sl) + 1;
            std::vector<double> node_bl((size_t)node0[(size_t)n_slots_], 0.0), bl(n, 0.0);
            for (int x = n - 1; x >= 0; --x) {
                const FactorUnit& u = funits[base + x];
                bl[x] = cost_of(u) + node_bl[(size_t)node0[(size_t)u.pub] + writer[x]];
                for (int q = 0; q < 3; ++q)
                    if (u.wait_flag[q] >= 0) {
                        double& nb = node_bl[(size_t)node0[(size_t)u.wait_flag[q]] + u.wait_val[q] / W - 1];
                        nb = std::max(nb, bl[x]);
                    }
            }
            // event simulation
            std::vector<std::vector<std::pair<int, int>>> waiters((size_t)n_slots_);   // per flag: (value, unit)
            std::vector<int> pending(n, 0), ver_sim((size_t)n_slots_, 0), order;
            order.reserve(n);
            for (int x = 0; x < n; ++x) {
                const FactorUnit& u = funits[base + x];
                for (int q = 0; q < 3; ++q)
                    if (u.wait_flag[q] >= 0) { waiters[(size_t)u.wait_flag[q]].push_back({u.wait_val[q], x}); ++pending[x]; }
            }
            std::vector<size_t> woke((size_t)n_slots_, 0);
            for (auto& wl : waiters) std::sort(wl.begin(), wl.end());
            auto worse = [&](int a, int b) { return bl[a] != bl[b] ? bl[a] < bl[b] : a > b; };   // heap top = longest chain, then list order
            std::vector<int> ready;
            for (int x = 0; x < n; ++x) if (pending[x] == 0) ready.push_back(x);
            std::make_heap(ready.begin(), ready.end(), worse);
            std::vector<std::pair<double, int>> running_ev;   // min-heap of (finish time, unit)
            auto later = [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a > b; };
            int free_p = 240;
            double now = 0.0;
            while ((int)order.size() < n) {
                while (free_p > 0 && !ready.empty()) {
                    std::pop_heap(ready.begin(), ready.end(), worse);
                    const int x = ready.back(); ready.pop_back();
                    order.push_back(x); --free_p;
                    running_ev.push_back({now + cost_of(funits[base + x]), x});
                    std::push_heap(running_ev.begin(), running_ev.end(), later);
                }
                if (running_ev.empty()) return "internal error: the dataflow units do not form a schedule";
                std::pop_heap(running_ev.begin(), running_ev.end(), later);
                const std::pair<double, int> ev = running_ev.back(); running_ev.pop_back();
                now = ev.first; ++free_p;
                const FactorUnit& u = funits[base + ev.second];
                const size_t f = (size_t)u.pub;
                ver_sim[f] += inc_of(u);
                while (woke[f] < waiters[f].size() && waiters[f][woke[f]].first <= ver_sim[f]) {
                    const int x = waiters[f][woke[f]++].second;
                    if (--pending[x] == 0) { ready.push_back(x); std::push_heap(ready.begin(), ready.end(), worse); }
                }
            }
            *sim_us = now;
            std::vector<FactorUnit> sorted(n);
            for (int i = 0; i < n; ++i) sorted[i] = funits[base + order[i]];
            std::fill(running.begin(), running.end(), 0);
            for (int i = 0; i < n; ++i) {
                const FactorUnit& u = sorted[i];
                for (int q = 0; q < 3; ++q)
                    if (u.wait_flag[q] >= 0 && running[(size_t)u.wait_flag[q]] < u.wait_val[q]) order_ok = false;
                running[(size_t)u.pub] += inc_of(u);
                funits[base + i] = u;
            }
            if (!order_ok) return "internal error: the scheduled dataflow order is not topological";
        }
        return "";
    };
    // Where the launch starts.  "factor_flow" > 0: the trailing groups with at most that many columns (and "factor_flow_rows"
    // off-diagonal tiles per column).  < 0 (default): by a model -- the level launches cost max(80 us of launch chain,
    // 0.14 us per tile product) per group, the dataflow launch what its list schedule says (it runs a tile product on one
    // CU at a time and reads every operand past the L2: ~0.22 us per product with all CUs busy, but a level costs it ~55 us
    // of chain instead of 80); the start with the smallest sum wins, no launch if none beats the level launches.
    std::vector<FactorUnit> funits;
    for (int ph = 0; ph < 2; ++ph) {
        const int g0 = ph == 0 ? 0 : n_local_groups_, g1 = ph == 0 ? n_local_groups_ : n_levels_;
        flow_g0_[ph] = flow_g1_[ph] = g1; flow_first_[ph] = (int)funits.size(); flow_n_[ph] = 0; flow_sim_us_[ph] = 0.0;
        if (flow_cols_ == 0 || g1 - g0 < 2) continue;
        std::vector<FactorUnit> best_units;
        double best_sim = 0.0;
        int best_gf = g1;
        if (flow_cols_ > 0) {
            int gf = g1;
            while (gf > g0) {
                const std::vector<int>& cols = level_cols[gf - 1];
                bool ok = (int)cols.size() <= flow_cols_;
                for (int K : cols) ok = ok && (int)col_rows[K].size() <= flow_rows_;
                if (!ok) break;
                --gf;
            }
            if (g1 - gf < 2) continue;   // a single group has nothing to chain
            const std::string e = make_flow_units(gf, g1, best_units, &best_sim);
            if (!e.empty()) return e;
            best_gf = gf;
        } else {
            auto level_us = [&](int g) {
                double prod = 0.0;
                for (int K : level_cols[g]) { const double m = (double)col_rows[K].size(); prod += m + 0.5 * m * (m + 1.0); }
                // (round 5: by the timeline a middle level of final-13682 really takes 140-250 us, ~90 + 0.11 prod -- but the launch's
                // own simulated time is as optimistic there, and the starts this pair of models picks ARE the measured optima:
                // profiles/r05_flow_dyn_sweep.txt.  Both left as they are.)
                return std::max(80.0, 0.14 * prod);
            };
            double level_tail = 0.0, best_total = 0.0;   // cost of the groups [gf, g1) by level launches; best (level head dropped: common)
            int64_t units = 0;
            // the candidate starts, from the top down, with what the level launches would cost from there
            std::vector<std::pair<int, double>> cands;
            for (int gf = g1 - 1; gf >= g0; --gf) {
                bool ok = (int)level_cols[gf].size() <= 64;
                for (int K : level_cols[gf]) {
                    const int64_t m = (int64_t)col_rows[K].size();
                    ok = ok && m <= 96;
                    units += 1 + kFlowUnitsPerTile * (m + m * (m + 1) / 2);
                }
                if (!ok || units > 400000) break;   // (the model is evaluated per candidate start: keep plan building in the milliseconds)
                level_tail += level_us(gf);
                if (g1 - gf >= 2) cands.push_back({gf, level_tail});
            }
            // The model of every candidate (its units in list-scheduled order, simulated) was half of the plan's build time on
            // final-13682 -- 50 of 95 ms, evaluated one after the other.  They are independent: a batch at a time on the host
            // pool, the choice replayed over the batch in the old order (same rule, same start), the winner's units built once
            // more at the end (round 5).
            const int batch = std::max(1, std::min<int>(8, (int)host_threads()));
            bool past = false;
            std::string err;
            for (size_t c0 = 0; c0 < cands.size() && !past && err.empty(); c0 += (size_t)batch) {
                const size_t c1 = std::min(cands.size(), c0 + (size_t)batch);
                std::vector<double> sims(c1 - c0, 0.0);
                std::vector<std::string> errs(c1 - c0);
                parallel_rows((int64_t)(c1 - c0), [&](int64_t i) {
                    std::vector<FactorUnit> scratch;
                    errs[(size_t)i] = make_flow_units(cands[c0 + (size_t)i].first, g1, scratch, &sims[(size_t)i]);
                }, 1);
                for (size_t i = 0; i < c1 - c0 && !past; ++i) {
                    if (!errs[i].empty()) { err = errs[i]; break; }
                    // gain of starting the launch at gf = what the level launches would have cost from there - the launch
                    const double gain = cands[c0 + i].second - (sims[i] + 15.0);
                    if (gain > best_total) { best_total = gain; best_gf = cands[c0 + i].first; best_sim = sims[i]; }
                    else if (gain < best_total - 300.0) past = true;   // past the optimum: the launch is swallowing throughput-bound levels
                }
            }
            if (!err.empty()) return err;
            if (best_gf == g1) continue;
            { double sim = 0.0; const std::string e = make_flow_units(best_gf, g1, best_units, &sim); if (!e.empty()) return e; }
        }
        flow_g0_[ph] = best_gf;
        flow_n_[ph] = (int)best_units.size();
        flow_sim_us_[ph] = best_sim;
        funits.insert(funits.end(), best_units.begin(), best_units.end());
    }
    // ---- first writers of the fill tiles (tile_plan.h, first_ok_) ------------------------------------------------------------
    // Two execution orders exist: the level launches alone (the lists of every level, in list order) and the level launches of
    // the levels below a dataflow launch followed by its units (in unit order: the writers of a tile are chained in that order).
    // A fill tile's first writer is flagged in both; touched tiles hold S and are never "first written".
    first_ok_ = false;
    if (!distributed() && n_slots_ > n_touched_ && flow_n_[1] == 0) {
        const size_t te = tile_elems;
        auto slot_of_ptr = [&](const double* c) { return (int64_t)((c - tiles_) / (ptrdiff_t)te); };
        std::vector<char> seen_a((size_t)n_slots_, 0);
        for (int64_t sl = 0; sl < n_touched_; ++sl) seen_a[(size_t)sl] = 1;
        std::vector<char> seen_b(seen_a);
        int64_t upd_before_flow = (int64_t)upd.size();   // the level lists that run in front of the dataflow launch
        if (flow_n_[0] > 0) {
            const int r = lv_upd_round_[(size_t)flow_g0_[0]];
            if (r < (int)upd_rounds_.size()) upd_before_flow = upd_rounds_[(size_t)r].first;
        }
        for (size_t q = 0; q < upd.size(); ++q) {
            const int64_t sl = slot_of_ptr(upd[q].C);
            if ((int64_t)q < upd_before_flow) seen_b[(size_t)sl] = 1;
            if (!seen_a[(size_t)sl]) { seen_a[(size_t)sl] = 1; upd[q].C = reinterpret_cast<double*>(reinterpret_cast<uintptr_t>(upd[q].C) | 1); }
        }
        // the dataflow units: the first (tile, writer) of a tile not written below the launch; its nine block units share C, A, B
        std::vector<const double*> first_a((size_t)n_slots_, nullptr), first_b((size_t)n_slots_, nullptr);
        for (FactorUnit& u : funits) {
            if (u.kind != 2 && u.kind != 3) continue;
            const int64_t sl = slot_of_ptr(u.C);
            if (!seen_b[(size_t)sl]) { seen_b[(size_t)sl] = 1; first_a[(size_t)sl] = u.A; first_b[(size_t)sl] = u.B; }
            if (first_a[(size_t)sl] == u.A && first_b[(size_t)sl] == u.B && first_a[(size_t)sl] != nullptr) u.kind |= kFlowFirstWriter;
        }
        bool all = true;
        for (int64_t sl = n_touched_; sl < n_slots_; ++sl) all = all && seen_a[(size_t)sl] && (flow_n_[0] == 0 || seen_b[(size_t)sl]);
        if (all) first_ok_ = true;
        else {   // (a fill tile without an update: cannot be -- take the flags back and clear everything as before)
            for (GemmTask& t : upd) t.C = reinterpret_cast<double*>(reinterpret_cast<uintptr_t>(t.C) & ~uintptr_t(7));
            for (FactorUnit& u : funits) u.kind &= 15;
        }
    }
    ptr_trace.mark("plan: task lists, dataflow units");
    potrf_h_ = potrf; trsm_h_ = trsm; upd_h_ = upd; flow_units_h_ = funits;   // (kept for check_schedule / the tools: small)
    TP_TRY(upload(&flow_units_, funits));
    if (flow_ver_ && !dry_run_) { (void)hipFree(flow_ver_); flow_ver_ = nullptr; }
    TP_TRY(dev_alloc(&flow_ver_, (size_t)n_slots_));
    TP_TRY(hipMemset(flow_ver_, 0, (size_t)std::max<int64_t>(n_slots_, 1) * sizeof(int)));
    n_sym_tiles_ = (int)symt.size();
    TP_TRY(upload(&sym_tiles_, symt));
    TP_TRY(alloc_zero(&sym_part_, (size_t)n_slots_ * 2 * kNB));
    TP_TRY(alloc_zero(&row_dot_, (size_t)nt_));
    TP_TRY(alloc_zero(&blk_part_, 2 * (size_t)((n_pad() + 255) / 256)));
    TP_TRY(alloc_zero(&scal_, 8));
    TP_TRY(upload(&tri_fwd_, tf));
    TP_TRY(upload(&tri_bwd_, tb));
    TP_TRY(upload(&flow_fwd_, ft));
    TP_TRY(upload(&flow_bwd_, bt));
    TP_TRY(alloc_zero(&flow_part_, (size_t)std::max(n_flow_parts_, 1) * kNB));
    if (flow_flags_ && !dry_run_) { (void)hipFree(flow_flags_); flow_flags_ = nullptr; }
    TP_TRY(dev_alloc(&flow_flags_, (size_t)2 * nt_ + 1));   // cnt[nt] | done[nt] | error word of the dataflow sweeps
    TP_TRY(hipMemset(flow_flags_, 0, ((size_t)2 * nt_ + 1) * sizeof(int)));
    if (!flow_err_host_ && !dry_run_) {
        TP_TRY(hipHostMalloc(reinterpret_cast<void**>(&flow_err_host_), 4 * sizeof(int), hipHostMallocDefault));
        flow_err_host_[0] = flow_err_host_[1] = flow_err_host_[2] = flow_err_host_[3] = 0;
        void* dp = nullptr;   // (pinned host memory is mapped: the kernels that post a word write it through this address)
        flow_err_host_dev_ = hipHostGetDevicePointer(&dp, flow_err_host_, 0) == hipSuccess ? static_cast<int*>(dp) : nullptr;
        (void)hipGetLastError();
    }
    TP_TRY(upload(&potrf_tasks_, potrf));
    TP_TRY(upload(&trsm_tasks_, trsm));
    TP_TRY(upload(&upd_tasks_, upd));
    TP_TRY(upload(&sym_row_ptr_, sym_ptr));
    TP_TRY(upload(&cls_, cls_h_));
    TP_TRY(alloc_zero(&exch_, (size_t)n_pad()));
    TP_TRY(upload(&sym_entries_, sym));
    // (a lowest-priority side stream was tried: no gain without graphs, +2.7 ms with them)
    // (and so was a CU-masked one that leaves 1 CU in 8 / 4 / 2 to the critical path: the same, either way)
    if (!side_) TP_TRY(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
    if (!so_) TP_TRY(hipStreamCreateWithFlags(&so_, hipStreamNonBlocking));
    if (!side2_) TP_TRY(hipStreamCreateWithFlags(&side2_, hipStreamNonBlocking));
    ev_t_.resize(n_levels_); ev_u2_.resize(n_levels_); ev_o_.resize(n_levels_); ev_b_.resize(n_levels_); ev_b2_.resize(n_levels_);
    if (dry_run_) {   // handles that identify streams and events in a schedule trace
        side_ = reinterpret_cast<hipStream_t>(uintptr_t(0x52)); side2_ = reinterpret_cast<hipStream_t>(uintptr_t(0x53));
        so_ = reinterpret_cast<hipStream_t>(uintptr_t(0x54));
        for (int i = 0; i < n_levels_; ++i) {
            ev_t_[i] = reinterpret_cast<hipEvent_t>(uintptr_t(0x10000 + 8 * i)); ev_u2_[i] = reinterpret_cast<hipEvent_t>(uintptr_t(0x10001 + 8 * i));
            ev_o_[i] = reinterpret_cast<hipEvent_t>(uintptr_t(0x10002 + 8 * i)); ev_b_[i] = reinterpret_cast<hipEvent_t>(uintptr_t(0x10003 + 8 * i));
            ev_b2_[i] = reinterpret_cast<hipEvent_t>(uintptr_t(0x10004 + 8 * i));
        }
        gate_cnt_ = reinterpret_cast<int*>(uintptr_t(1) << 46);
    }
    u2_pending_.assign(n_levels_, false);
    o_pending_.assign(n_levels_, false);
    for (int i = 0; i < n_levels_ && !dry_run_; ++i) {
        TP_TRY(hipEventCreateWithFlags(&ev_t_[i], hipEventDisableTiming));
        TP_TRY(hipEventCreateWithFlags(&ev_u2_[i], hipEventDisableTiming));
        TP_TRY(hipEventCreateWithFlags(&ev_o_[i], hipEventDisableTiming));
        TP_TRY(hipEventCreateWithFlags(&ev_b_[i], hipEventDisableTiming));
        TP_TRY(hipEventCreateWithFlags(&ev_b2_[i], hipEventDisableTiming));
    }
    TP_TRY(hipMalloc(&gate_cnt_, (size_t)(n_levels_ + 1) * sizeof(int)));
    TP_TRY(hipDeviceSynchronize());  // the null-stream memsets above precede any work on the stream
    ptr_trace.mark("plan: uploads, streams, events");
#undef TP_TRY
    return "";
}

hipError_t TilePlan::zero_tiles(bool own_touched_only, hipStream_t on, bool skip_fill) {
    const hipStream_t zs = on ? on : stream_;
    const size_t te = (size_t)kNB * kNB * sizeof(double);
    hipError_t e = hipSuccess;
    auto clear = [&](int64_t first, int64_t count) {
        if (e == hipSuccess && count > 0) e = hipMemsetAsync(tiles_ + (size_t)first * kNB * kNB, 0, (size_t)count * te, zs);
    };
    if (distributed() && !own_all_ && part_rank_ < (int)own_range_.size()) {
        // a rank of a distributed plan factorises its own columns and the shared top: the fill tiles of the other ranks'
        // columns are never touched; their touched tiles only when this rank's landmarks may add to them (range sharding)
        if (own_touched_only) { clear(own_range_[part_rank_].first, own_range_[part_rank_].second); clear(n_t_nt_, n_touched_ - n_t_nt_); }
        else clear(0, n_touched_);
        clear(own_fill_[part_rank_].first, own_fill_[part_rank_].second);
        clear(n_f_nt_, n_slots_ - n_f_nt_);
    } else {
        clear(0, (skip_fill && first_ok_) ? n_touched_ : n_slots_);
    }
    if (e != hipSuccess) return e;
    if (dry_run_) return hipSuccess;
    launch_clear_i32(flag_, 4, zs);
    return hipGetLastError();
}

void TilePlan::add_diag(int n_valid, double add_valid, double pad_value) {
    launch_tile_add_diag(tiles_, diag_slot_, n_valid, (int)n_pad(), add_valid, pad_value, stream_);
}

void TilePlan::scale_sym(const double* scale) { launch_tile_scale_sym(sym_tiles_, n_sym_tiles_, tiles_, scale, stream_); }

void TilePlan::diag(double* out) const { launch_tile_diag(tiles_, diag_slot_, nt_, out, stream_); }

// forward step of level group lv: one launch, or one per column where the plan asks for it (fwd_cut_)
void TilePlan::launch_fwd_group(int lv, double* bvec, double* yvec, hipStream_t s) {
    const std::vector<int>& cut = fwd_cut_[lv];
    if (cut.size() < 2) {
        launch_tri_step(false, tri_fwd_ + lv_fwd_[lv], lv_fwd_[lv + 1] - lv_fwd_[lv], bvec, yvec, s);
        return;
    }
    for (size_t i = 0; i < cut.size(); ++i) {
        const int b = cut[i], e = i + 1 < cut.size() ? cut[i + 1] : lv_fwd_[lv + 1];
        launch_tri_step(false, tri_fwd_ + b, e - b, bvec, yvec, s);
    }
}

// The factorisation and the triangular solves are static launch sequences for a given structure:
// they are captured once into hipGraphs (a few hundred dependent launches would otherwise be paced by
// host launch overhead) and replayed every iteration.
void TilePlan::enqueue_factor(int g0, int g1) {
    // Three streams.  Main: potrf(lv), panel solves(lv), U1d(lv) = the updates of the next level's DIAGONAL tiles (all
    // its potrf needs).  Third: U1o(lv) = the updates of the other tiles of the next level's columns, beside that
    // potrf; the next panel solves wait for them.  Side: U2(lv) = every other update of level lv, overlapped with
    // potrf / panel solves of level lv+1 (one workgroup resp. a few dozen: they leave the chip nearly empty).
    // Ordering that keeps every tile's read-modify-write sequence race free:
    //   U2(lv) after the panel solves of lv;  U1d(lv), U1o(lv) after U2a(lv-1) -- the part of U2(lv-1) whose targets lie in
    //   the columns of level lv+1, and with it (side-stream order) every older side-stream update; U2b(lv-1), targets in
    //   level lv+2 and above, runs on beside them (round 3: the wait for the whole of U2(lv-1) had become the critical chain
    //   once the flood gate let the potrf start on time);
    //   potrf(lv) after U1d(lv-1) [stream order] and whatever U1d(lv-1) waited for;
    //   panel(lv) after U1o(lv-1) [event];  U1o(lv) and U2(lv) hit different columns (level lv+1 / above);
    //   a U2 too small for the side stream runs on the main stream after the side stream's last U2b [ev_b_].
    // Every call below goes through these shadows: with a trace attached (schedule_trace: tests, host-only plans) the call
    // is recorded instead of issued -- what check_schedule() then proves is this very sequence.
    std::vector<SchedOp>* const tr = sched_trace_;
    auto hipStreamWaitEvent = [&](hipStream_t s, hipEvent_t e, unsigned) { if (tr) { tr->push_back({2, (uintptr_t)s, (uintptr_t)e, -1, 0, 0}); return hipSuccess; } return ::hipStreamWaitEvent(s, e, 0); };
    auto hipEventRecord = [&](hipEvent_t e, hipStream_t s) { if (tr) { tr->push_back({1, (uintptr_t)s, (uintptr_t)e, -1, 0, 0}); return hipSuccess; } return ::hipEventRecord(e, s); };
    auto launch_potrf_inv = [&](const PotrfTask* t, int n, int* fail, hipStream_t s, int* arrived) {
        if (tr) { if (n > 0) tr->push_back({0, (uintptr_t)s, 0, 0, (int64_t)(t - potrf_tasks_), n}); return; }
        apex::launch_potrf_inv(t, n, fail, s, arrived);
    };
    auto launch_tile_gemm_nt = [&](const GemmTask* t, int n, double alpha, double beta, hipStream_t s) {
        if (tr) {
            const bool panel = beta == 0.0;
            if (n > 0) tr->push_back({0, (uintptr_t)s, 0, panel ? 1 : 2, (int64_t)(t - (panel ? trsm_tasks_ : upd_tasks_)), n});
            return;
        }
        apex::launch_tile_gemm_nt(t, n, alpha, beta, s, /*tri_b=*/beta == 0.0);   // the panel solves multiply by Linv
    };
    auto launch_gate = [&](const int* a, int expected, int us, hipStream_t s) { if (!tr) apex::launch_gate(a, expected, us, s); };
    auto hipMemsetAsync = [&](void* p, int v, size_t n, hipStream_t s) { if (tr) return hipSuccess; return ::hipMemsetAsync(p, v, n, s); };
    auto launch_clear_i32 = [&](int* p, int64_t n, hipStream_t s) { if (!tr) apex::launch_clear_i32(p, n, s); };
    auto launch_factor_flow = [&](const FactorUnit* u, int n, int* ver, int* fail, int* err, hipStream_t s, unsigned long long* trace) {
        if (tr) { tr->push_back({0, (uintptr_t)s, 0, 3, (int64_t)(u - flow_units_), n}); return; }
        apex::launch_factor_flow(u, n, ver, fail, err, s, trace);
    };
    const bool two = overlap_ && side_ != nullptr && n_levels_ > 2;
    // the trailing groups [gf, g1) of this phase run as one dataflow launch behind the level launches (build())
    const int ph = (g0 == n_local_groups_ && g1 == n_levels_ && n_local_groups_ < n_levels_) ? 1 : 0;
    const int g_end = g1;
    if (flow_on_ && flow_n_[ph] > 0 && flow_g0_[ph] >= g0 && flow_g1_[ph] == g1) g1 = flow_g0_[ph];
    if (gate_min_ > 0 && gate_cnt_) launch_clear_i32(gate_cnt_, n_levels_ + 1, stream_);
    int last_a = -1, last_b = -1;   // last levels with work on the side streams A / B that the main stream has not waited for
    std::vector<int> lastb((size_t)std::max(g1 - g0, 1), -1);   // lastb[lv - g0]: the last level <= lv with U2b2 work on stream B
    int b2_pending = -1, a_waited = -1;
    auto a_wait_upto = [&](int lvb) {   // stream A waits for stream B up to level lvb's U2b2 (B runs in order)
        if (lvb > a_waited) { (void)hipStreamWaitEvent(side_, ev_b2_[lvb], 0); a_waited = lvb; }
    };
    for (int lv = g0; lv < g1; ++lv) {
        launch_potrf_inv(potrf_tasks_ + lv_potrf_[lv], lv_potrf_[lv + 1] - lv_potrf_[lv], flag_, stream_,
                         gate_min_ > 0 && gate_cnt_ ? gate_cnt_ + lv : nullptr);
        // the panel solves work on the off-diagonal tiles of this level's columns: U1o of the level below must be in
        if (lv > g0 && o_pending_[lv - 1]) (void)hipStreamWaitEvent(stream_, ev_o_[lv - 1], 0);
        const int r0 = lv_upd_round_[lv], rd = lv_upd_splitd_[lv], rs = lv_upd_split_[lv], r1 = lv_upd_round_[lv + 1];
        int64_t n_u2 = 0, n_o = 0;
        for (int r = rs; r < r1; ++r) n_u2 += upd_rounds_[r].second;
        for (int r = rd; r < rs; ++r) n_o += upd_rounds_[r].second;
        // a cross-stream edge costs a few microseconds in the graph: only worth it when the batch is a real one
        const bool has_u2 = two && n_u2 >= overlap_min_;
        const bool has_o = two && so_ != nullptr && split_u1_ && n_o >= split_u1_min_;
        launch_tile_gemm_nt(trsm_tasks_ + lv_trsm_[lv], lv_trsm_[lv + 1] - lv_trsm_[lv], 1.0, 0.0, stream_);
        if (has_u2 || has_o) (void)hipEventRecord(ev_t_[lv], stream_);
        if (has_u2) (void)hipStreamWaitEvent(side_, ev_t_[lv], 0);
        if (has_o) (void)hipStreamWaitEvent(so_, ev_t_[lv], 0);
        if (two && lv > g0 && u2_pending_[lv - 1]) {
            (void)hipStreamWaitEvent(stream_, ev_u2_[lv - 1], 0);
            if (has_o) (void)hipStreamWaitEvent(so_, ev_u2_[lv - 1], 0);
        } else if (two && lv > g0 && !debug_skip_idle_wait_) {
            // Level lv-1 put nothing on the side streams, so there is no ev_u2_[lv-1] to carry "every older side-stream update
            // precedes U1(lv)": U2b1(lv-2) [targets in level lv+1, stream A] and the U2b2 of levels <= lv-3 [stream B] may
            // still be at work on the tiles U1(lv) is about to update (and that potrf(lv+1) then reads).  Levels are assigned
            // by height, so a chain can pass through such a level.  Wait for both side streams outright.
            if (last_a >= 0) {
                (void)hipStreamWaitEvent(stream_, ev_b_[last_a], 0);
                if (has_o) (void)hipStreamWaitEvent(so_, ev_b_[last_a], 0);
                last_a = -1;
            }
            if (last_b >= 0) {
                (void)hipStreamWaitEvent(stream_, ev_b2_[last_b], 0);
                if (has_o) (void)hipStreamWaitEvent(so_, ev_b2_[last_b], 0);
                last_b = -1;
            }
        }
        for (int r = r0; r < rd; ++r)   // U1d: what the next potrf needs
            launch_tile_gemm_nt(upd_tasks_ + upd_rounds_[r].first, (int)upd_rounds_[r].second, -1.0, 1.0, stream_);
        hipStream_t s1 = has_o ? so_ : stream_;
        for (int r = rd; r < rs; ++r)   // U1o: what the next panel solves need, beside the next potrf
            launch_tile_gemm_nt(upd_tasks_ + upd_rounds_[r].first, (int)upd_rounds_[r].second, -1.0, 1.0, s1);
        o_pending_[lv] = has_o;
        if (has_o) (void)hipEventRecord(ev_o_[lv], so_);
        // U2 on two streams of its own.  A (side_): U2a(lv) [targets in level lv+2: what U1(lv+1) waits for], then U2b1(lv)
        // [level lv+3].  B (side2_): U2b2(lv) [level lv+4 and above: the bulk].  Writers of one target level t, in time:
        // U2b2(<= t-4) -> U2b1(t-3) -> U2a(t-2) -> U1(t-1); B orders the first among themselves, U2b1(lv) waits for
        // U2b2(lv-1) [ev_b2_], the rest is stream order on A and ev_u2_.  U2a(lv+1) thus waits for U2b1(lv) only, not for the
        // bulk of level lv (on one stream it did, and through it U1d(lv+2) and the potrf behind it).
        // (only when there is such work: a stream that joins the capture must come back to it with an event)
        const bool b2_side = has_u2 && side2_ != nullptr && two_side_plan_ && r1 > lv_upd_splitb_[lv];
        // flood gate: the bulk updates of a big level start when the next level's potrf workgroups sit on their CUs (they
        // follow U1d on the main stream) -- otherwise the update's grid takes every CU first and the potrf, 124 KB of LDS per
        // workgroup, waits for it to drain
        const bool gated = has_u2 && gate_min_ > 0 && gate_cnt_ && n_u2 >= gate_min_ && lv + 1 < g1;
        if (gated) launch_gate(gate_cnt_ + lv + 1, lv_potrf_[lv + 2] - lv_potrf_[lv + 1], 150, side_);
        // a small U2 stays on the main stream: earlier levels' U2b may still be at work on the same targets over there
        if (!has_u2 && r1 > rs) {
            if (last_a >= 0) { (void)hipStreamWaitEvent(stream_, ev_b_[last_a], 0); last_a = -1; }
            if (last_b >= 0) { (void)hipStreamWaitEvent(stream_, ev_b2_[last_b], 0); last_b = -1; }
        }
        const int ra = lv_upd_splita_[lv], rb = lv_upd_splitb_[lv];
        hipStream_t sa = has_u2 ? side_ : stream_, sb = b2_side ? side2_ : sa;
        // U2a(lv) [level lv+2] follows every U2b2 of levels <= lv-2 [their targets start at level lv+2] ...
        if (has_u2 && lv - 2 >= g0) a_wait_upto(lastb[lv - 2 - g0]);
        for (int r = rs; r < ra; ++r)   // U2a
            launch_tile_gemm_nt(upd_tasks_ + upd_rounds_[r].first, (int)upd_rounds_[r].second, -1.0, 1.0, sa);
        u2_pending_[lv] = has_u2;
        if (has_u2) (void)hipEventRecord(ev_u2_[lv], side_);   // ... and, in stream order, every earlier update on A
        if (b2_side) {
            (void)hipStreamWaitEvent(side2_, ev_t_[lv], 0);
            if (gated) launch_gate(gate_cnt_ + lv + 1, lv_potrf_[lv + 2] - lv_potrf_[lv + 1], 150, side2_);
        }
        if (has_u2 && lv - 1 >= g0) a_wait_upto(lastb[lv - 1 - g0]);   // ... and U2b1(lv) [level lv+3] every U2b2 of levels <= lv-1
        for (int r = ra; r < rb; ++r)   // U2b1
            launch_tile_gemm_nt(upd_tasks_ + upd_rounds_[r].first, (int)upd_rounds_[r].second, -1.0, 1.0, sa);
        for (int r = rb; r < r1; ++r)   // U2b2
            launch_tile_gemm_nt(upd_tasks_ + upd_rounds_[r].first, (int)upd_rounds_[r].second, -1.0, 1.0, sb);
        if (has_u2) { (void)hipEventRecord(ev_b_[lv], side_); last_a = lv; }
        if (b2_side) { (void)hipEventRecord(ev_b2_[lv], side2_); last_b = lv; }
        lastb[lv - g0] = b2_pending = b2_side ? lv : b2_pending;
    }
    if (g1 > g0 && o_pending_[g1 - 1]) (void)hipStreamWaitEvent(stream_, ev_o_[g1 - 1], 0);
    // join: the last side-stream work precedes whatever follows on the main stream
    if (last_a >= 0) (void)hipStreamWaitEvent(stream_, ev_b_[last_a], 0);
    if (last_b >= 0) (void)hipStreamWaitEvent(stream_, ev_b2_[last_b], 0);
    if (g1 < g_end) {   // every update the level launches add to the region's tiles is in: the joins above
        launch_clear_i32(flow_ver_, n_slots_, stream_);
        if (poison_factor_ && !tr)   // (tests: the version of the first unit's tile starts hugely negative and is never reached)
            (void)hipMemsetAsync(flow_ver_ + flow_units_h_[(size_t)flow_first_[ph]].pub, 0x80, sizeof(int), stream_);
        launch_factor_flow(flow_units_ + flow_first_[ph], flow_n_[ph], flow_ver_, flag_, flag_ + 1, stream_,
                           flow_trace_ ? flow_trace_ + 3 * (size_t)flow_first_[ph] : nullptr);
    }
}

void TilePlan::enqueue_solve(const double* rhs, double* x, double* work) {
    // L y = rhs (work vector bvec), then L^T x = y (work vector yvec); level by level
    double* bvec = work;
    double* yvec = work + n_pad();
    const bool flow = tri_flow_ && n_flow_tasks_ > 0;
    if (flow) {
        launch_tri_flow(false, flow_fwd_, n_flow_tasks_, rhs, yvec, flow_part_, flow_flags_, nt_, stream_, nullptr, nullptr,
                        poison_ == 1 ? nt_ - 1 : -1);
    } else {
        (void)hipMemcpyAsync(bvec, rhs, n_pad() * sizeof(double), hipMemcpyDeviceToDevice, stream_);
        for (int lv = 0; lv < n_levels_; ++lv)
            launch_fwd_group(lv, bvec, yvec, stream_);
    }
    if (flow) {
        launch_tri_flow(true, flow_bwd_, n_flow_bwd_, yvec, x, flow_part_, flow_flags_, nt_, stream_, nullptr, nullptr,
                        poison_ == 2 ? nt_ - 1 : -1);
        return;
    }
    for (int s = 0; s < n_levels_; ++s)
        launch_tri_step(true, tri_bwd_ + lv_bwd_[s], lv_bwd_[s + 1] - lv_bwd_[s], yvec, x, stream_);
}

// The distributed triangular solves (see tile_plan.h).  bvec/yvec as in enqueue_solve; masks: bit (1 << class).
void TilePlan::enqueue_dist_solve(int phase, const double* rhs, double* x, double* work) {
    double* bvec = work;
    double* yvec = work + n_pad();
    const int n = (int)n_pad();
    const int L1 = n_local_groups_;
    const bool flow = tri_flow_ && n_flow_local_ > 0;
    if (phase == 0 && flow) {
        // dataflow form: this rank's columns in one launch; its contributions to the shared top blocks are folded
        // straight into the exchange vector (the top blocks of the right-hand side enter the sum once, on rank 0)
        (void)hipMemsetAsync(exch_, 0, n_pad() * sizeof(double), stream_);
        launch_tri_flow(false, flow_fwd_, n_flow_local_, rhs, yvec, flow_part_, flow_flags_, nt_, stream_,
                        part_rank_ == 0 ? rhs : nullptr, exch_);
    } else if (phase == 1 && flow) {
        // the top columns forward (right-hand side = the summed exchange vector), then everything backward, top first.
        // Pull form, fixed fold order: the ranks' copies of the top solution are bitwise equal by construction.
        launch_tri_flow(false, flow_fwd_ + n_flow_local_, n_flow_tasks_ - n_flow_local_, exch_, yvec, flow_part_, flow_flags_, nt_,
                        stream_, nullptr, nullptr);
        launch_tri_flow(true, flow_bwd_, n_flow_bwd_, yvec, x, flow_part_, flow_flags_, nt_, stream_, nullptr, nullptr);
        launch_vec_select(n, x, cls_, part_rank_ == 0 ? 6 : 2, exch_, stream_);
    } else if (phase == 0) {
        // the top blocks of the right-hand side enter the sum once (rank 0); every rank adds its columns' updates
        launch_vec_select(n, rhs, cls_, part_rank_ == 0 ? 7 : 3, bvec, stream_);
        for (int lv = 0; lv < L1; ++lv)
            launch_fwd_group(lv, bvec, yvec, stream_);
        launch_vec_select(n, bvec, cls_, 4, exch_, stream_);
    } else if (phase == 1) {
        launch_vec_merge(n, exch_, cls_, 4, bvec, stream_);
        for (int lv = L1; lv < n_levels_; ++lv)
            launch_fwd_group(lv, bvec, yvec, stream_);
        for (int s = 0; s < n_levels_; ++s)  // top groups first, then this rank's
            launch_tri_step(true, tri_bwd_ + lv_bwd_[s], lv_bwd_[s + 1] - lv_bwd_[s], yvec, x, stream_);
        launch_vec_select(n, x, cls_, part_rank_ == 0 ? 6 : 2, exch_, stream_);
    } else {
        (void)hipMemcpyAsync(x, exch_, n_pad() * sizeof(double), hipMemcpyDeviceToDevice, stream_);
    }
}

// graph 0: factorisation of the local levels (all levels, + fused forward sweep when rhs/work are given, in a plan
// that is not distributed), 1: both sweeps, 2: backward sweep only, 3: factorisation of the top levels,
// 4/5: phases 0/1 of the distributed solve
bool TilePlan::run_graph(int which, const double* rhs, double* x, double* work) {
    if (!use_graphs_) return false;
    if (graph_exec_[which] && (rhs != graph_rhs_[which] || x != graph_x_[which] || work != graph_work_[which])) {
        (void)hipGraphExecDestroy(graph_exec_[which]);  // the captured pointers changed
        graph_exec_[which] = nullptr;
    }
    if (!graph_exec_[which]) {
        if (graph_failed_[which]) return false;
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal) != hipSuccess) { graph_failed_[which] = true; return false; }
        if (which == 0) enqueue_factor(0, n_local_groups_);
        else if (which == 3) enqueue_factor(n_local_groups_, n_levels_);
        else if (which == 4 || which == 5) enqueue_dist_solve(which - 4, rhs, x, work);
        else enqueue_solve(rhs, x, work);
        if (hipStreamEndCapture(stream_, &g) != hipSuccess || !g) { graph_failed_[which] = true; (void)hipGetLastError(); return false; }
        hipGraphExec_t ex = nullptr;
        if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(g); graph_failed_[which] = true; (void)hipGetLastError(); return false; }
        (void)hipGraphDestroy(g);
        graph_exec_[which] = ex;
        graph_rhs_[which] = rhs; graph_x_[which] = x; graph_work_[which] = work;
    }
    return hipGraphLaunch(graph_exec_[which], stream_) == hipSuccess;
}

void TilePlan::enable_tri_flow(bool on) {
    if (on == tri_flow_) return;
    tri_flow_ = on;
    for (int which : {1, 4, 5})   // the captured sweeps change
        if (graph_exec_[which]) { (void)hipGraphExecDestroy(graph_exec_[which]); graph_exec_[which] = nullptr; }
}

hipError_t TilePlan::enable_flow_trace() {
    if (flow_trace_) return hipSuccess;
    const size_t n = 3 * (size_t)std::max(flow_n_[0] + flow_n_[1], 1);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&flow_trace_), n * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    for (int which : {0, 3})   // the captured launches hold the old (null) pointer
        if (graph_exec_[which]) { (void)hipGraphExecDestroy(graph_exec_[which]); graph_exec_[which] = nullptr; }
    return hipMemset(flow_trace_, 0, n * sizeof(unsigned long long));
}

hipError_t TilePlan::read_flow_trace(std::vector<FactorUnit>* units, std::vector<unsigned long long>* stamps) {
    const size_t n = (size_t)(flow_n_[0] + flow_n_[1]);
    units->resize(n); stamps->resize(3 * n);
    if (n == 0 || !flow_trace_) return hipErrorNotInitialized;
    hipError_t e = hipMemcpy(units->data(), flow_units_, n * sizeof(FactorUnit), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    return hipMemcpy(stamps->data(), flow_trace_, 3 * n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}

std::vector<SchedOp> TilePlan::schedule_trace(int phase) {
    std::vector<SchedOp> ops;
    sched_trace_ = &ops;
    if (phase == 0) enqueue_factor(0, n_local_groups_);
    else enqueue_factor(n_local_groups_, n_levels_);
    sched_trace_ = nullptr;
    return ops;
}

int TilePlan::check_schedule(const std::vector<SchedOp>& ops, std::string* first_violation) const {
    // vector clocks over the streams that appear: clock[s] = how many launches of stream s happen before this point
    std::vector<uintptr_t> streams;
    auto sid = [&](uintptr_t s) { for (size_t i = 0; i < streams.size(); ++i) if (streams[i] == s) return (int)i; streams.push_back(s); return (int)streams.size() - 1; };
    for (const SchedOp& o : ops) (void)sid(o.stream);
    const int S = (int)streams.size();
    typedef std::vector<int> Clock;
    std::vector<Clock> now((size_t)S, Clock((size_t)S, 0));     // per stream: what precedes its next call
    std::vector<std::pair<uintptr_t, Clock>> events;             // last record of each event
    struct Access { int launch; bool write; };
    struct Launch { int stream, pos; Clock before; const SchedOp* op; };
    std::vector<Launch> launches;
    std::vector<std::pair<const double*, Access>> acc;
    int bad = 0;
    auto complain = [&](const std::string& m) { if (bad++ == 0 && first_violation) *first_violation = m; };
    auto describe = [&](const Launch& l) {
        static const char* const names[] = {"potrf", "panel solves", "updates", "dataflow launch"};
        return std::string(names[l.op->list]) + " [" + std::to_string(l.op->first) + ", +" + std::to_string(l.op->count) + ") on stream " + std::to_string(l.stream);
    };
    for (const SchedOp& o : ops) {
        const int s = sid(o.stream);
        if (o.op == 1) {
            bool found = false;
            for (auto& e : events) if (e.first == o.event) { e.second = now[(size_t)s]; found = true; }
            if (!found) events.push_back({o.event, now[(size_t)s]});
        } else if (o.op == 2) {
            bool found = false;
            for (const auto& e : events)
                if (e.first == o.event) { for (int k = 0; k < S; ++k) now[(size_t)s][(size_t)k] = std::max(now[(size_t)s][(size_t)k], e.second[(size_t)k]); found = true; }
            if (!found) complain("a stream waits for an event that was never recorded");
        } else {
            const int li = (int)launches.size();
            launches.push_back({s, now[(size_t)s][(size_t)s] + 1, now[(size_t)s], &o});
            now[(size_t)s][(size_t)s] += 1;
            // the tiles the launch touches; inside one launch no tile may be written twice or read and written by two tasks
            std::vector<std::pair<const double*, int>> local;   // (tile, +1 write / 0 read) of this launch
            auto touch = [&](const double* t, bool w) { acc.push_back({t, {li, w}}); local.push_back({t, w ? 1 : 0}); };
            for (int64_t q = o.first; q < o.first + o.count; ++q) {
                if (o.list == 0) { touch(potrf_h_[(size_t)q].A, true); touch(potrf_h_[(size_t)q].Linv, true); }
                else if (o.list == 1) { touch(trsm_h_[(size_t)q].C, true); touch(trsm_h_[(size_t)q].B, false); }
                else if (o.list == 2) { touch(reinterpret_cast<const double*>(reinterpret_cast<uintptr_t>(upd_h_[(size_t)q].C) & ~uintptr_t(7)), true); touch(upd_h_[(size_t)q].A, false); touch(upd_h_[(size_t)q].B, false); }
                else {   // the dataflow launch orders its own units (version counters): one writer of everything it touches
                    const FactorUnit& u = flow_units_h_[(size_t)q];
                    touch(u.C, true);
                    if ((u.kind & 15) == 0) touch(u.A, true);
                }
            }
            if (o.list != 3) {
                std::sort(local.begin(), local.end());
                for (size_t i = 0; i < local.size();) {
                    size_t j = i; int writes = 0;
                    while (j < local.size() && local[j].first == local[i].first) writes += local[j++].second;
                    if (writes >= 1 && j - i >= 2) { complain("two tasks of one launch touch a tile that one of them writes: " + describe(launches.back())); break; }
                    i = j;
                }
            }
        }
    }
    // every pair of launches on one tile with a writer among them must be ordered
    std::sort(acc.begin(), acc.end(), [](const std::pair<const double*, Access>& a, const std::pair<const double*, Access>& b) {
        return a.first != b.first ? a.first < b.first : a.second.launch < b.second.launch; });
    for (size_t i = 0; i < acc.size();) {
        size_t j = i;
        while (j < acc.size() && acc[j].first == acc[i].first) ++j;
        for (size_t a = i; a < j; ++a)
            for (size_t b = a + 1; b < j; ++b) {
                const Access &x = acc[a].second, &y = acc[b].second;
                if (x.launch == y.launch || (!x.write && !y.write)) continue;
                const Launch &lx = launches[(size_t)x.launch], &ly = launches[(size_t)y.launch];   // lx was issued first
                if (ly.before[(size_t)lx.stream] < lx.pos)
                    complain("unordered accesses to one tile: " + describe(lx) + " and " + describe(ly));
            }
        i = j;
    }
    return bad;
}

void TilePlan::top_slot_ranges(std::pair<int64_t, int64_t> out[2]) const {
    out[0] = {n_t_nt_, n_touched_ - n_t_nt_};
    out[1] = {n_f_nt_, n_slots_ - n_f_nt_};
}

void TilePlan::factor_phase(int phase) {
    if (phase == 0) { if (!run_graph(0, nullptr, nullptr, nullptr)) enqueue_factor(0, n_local_groups_); }
    else if (!run_graph(3, nullptr, nullptr, nullptr)) enqueue_factor(n_local_groups_, n_levels_);
}

void TilePlan::solve_phase(int phase, const double* rhs, double* x, double* work) {
    if (phase == 2 || !run_graph(4 + phase, rhs, x, work)) enqueue_dist_solve(phase, rhs, x, work);
}

hipError_t TilePlan::factor(int* failed_at, bool defer_flags) {
    if (distributed()) {
        if (!comm_.sum || !comm_.max_int) return hipErrorNotInitialized;  // a distributed plan needs its communicator
        factor_phase(0);
        std::pair<int64_t, int64_t> rg[2];
        top_slot_ranges(rg);
        const size_t te = (size_t)kNB * kNB;
        for (int i = 0; i < 2; ++i)
            if (rg[i].second > 0 && !comm_.sum(tiles_ + (size_t)rg[i].first * te, (size_t)rg[i].second * te, stream_)) return hipErrorUnknown;
        factor_phase(1);
        if (!comm_.max_int(flag_, 2, stream_)) return hipErrorUnknown;  // a failed pivot (or a dataflow time-out) anywhere fails the factorisation everywhere
        return read_flags(failed_at);
    }
    if (poison_factor_) {   // (tests: the poisoned launch is not part of the captured graphs)
        enqueue_factor(0, n_levels_);
        poison_factor_ = false;
    } else if (!run_graph(0, nullptr, nullptr, nullptr)) enqueue_factor(0, n_levels_);
    if (defer_flags) { *failed_at = 0; return hipGetLastError(); }
    return read_flags(failed_at);
}

// The pivot flag of this factorisation.  (The error word of the dataflow sweeps belongs to the SOLVE that ran them:
// post_sweep_status / sweep_timed_out.)
hipError_t TilePlan::read_flags(int* failed_at) {
    int f[2] = {0, 0};   // [0] first failed tile column + 1, [1] error word of the dataflow factorisation (k_factor_flow)
    hipError_t e = hipMemcpyAsync(f, flag_, sizeof f, hipMemcpyDeviceToHost, stream_);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream_);
    *failed_at = f[0];
    if (e == hipSuccess && f[1] != 0) {
        // a unit gave up waiting (flow_wait's spin limit): the tiles are half updated.  Back to the level launches for the
        // rest of the plan's life; the caller re-assembles and factorises again (factor_flow_gave_up()).
        flow_gave_up_ = true;
        flow_on_ = false;
        for (int which : {0, 3})
            if (graph_exec_[which]) { (void)hipGraphExecDestroy(graph_exec_[which]); graph_exec_[which] = nullptr; }
        (void)hipMemsetAsync(flag_ + 1, 0, sizeof(int), stream_);
    }
    return e;
}

// Behind the sweeps of a solve: the error word goes to pinned host memory (no synchronisation here: the caller's next
// one covers it) and is cleared for the next solve.  Distributed plans take the max over the ranks first -- a rank whose
// sweep gave up must not be the only one that repeats the solve, the others would be waiting in its collectives.
bool TilePlan::post_sweep_status(bool reduce) {
    if (!flow_flags_ || !flow_err_host_) return true;
    int* err = flow_flags_ + 2 * (size_t)nt_;
    if (reduce && comm_.max_int && !comm_.max_int(err, 1, stream_)) return false;   // (the communicator keeps its message)
    if (!reduce && flow_err_host_dev_) { launch_post_word(err, flow_err_host_dev_, stream_); return true; }   // (one launch, no copy engine)
    (void)hipMemcpyAsync(flow_err_host_, err, sizeof(int), hipMemcpyDeviceToHost, stream_);
    (void)hipMemsetAsync(err, 0, sizeof(int), stream_);
    return true;
}

bool TilePlan::sweep_timed_out() {
    if (!flow_err_host_ || flow_err_host_[0] == 0) return false;
    flow_err_host_[0] = 0;
    ++n_sweep_timeouts_;
    return true;
}

hipError_t TilePlan::debug_occupy_cus(int n_cus, int micros) {
    if (!flow_err_host_) return hipErrorNotInitialized;
    if (!occ_stream_) { const hipError_t e = hipStreamCreateWithFlags(&occ_stream_, hipStreamNonBlocking); if (e != hipSuccess) return e; }
    volatile int* started = flow_err_host_ + 1;
    started[0] = 0; started[1] = 0;
    launch_occupy_cus(n_cus, micros, flow_err_host_ + 1, occ_stream_);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    for (int spin = 0; spin < 2000 && started[0] < n_cus; ++spin) std::this_thread::sleep_for(std::chrono::microseconds(100));
    return started[0] >= n_cus ? hipSuccess : hipErrorNotReady;
}

hipError_t TilePlan::solve(const double* rhs, double* x, double* work) {
    if (distributed()) {
        if (!comm_.sum) return hipErrorNotInitialized;
        solve_phase(0, rhs, x, work);
        if (!comm_.sum(exch_, (size_t)n_pad(), stream_)) return hipErrorUnknown;
        solve_phase(1, rhs, x, work);
        if (!comm_.sum(exch_, (size_t)n_pad(), stream_)) return hipErrorUnknown;
        solve_phase(2, rhs, x, work);
        // (the collective is entered by every rank or by none: tri_flow_ is a plan-wide setting, and a distributed plan has
        // dataflow tasks on every rank -- at least the fold tasks of the shared top columns)
        if (tri_flow_ && !post_sweep_status(true)) return hipErrorUnknown;
        return hipGetLastError();
    }
    if (poison_ != 0) {   // (tests: the poisoned launch is not part of the captured graphs)
        enqueue_solve(rhs, x, work);
        poison_ = 0;
    } else if (!run_graph(1, rhs, x, work)) enqueue_solve(rhs, x, work);
    if (tri_flow_ && n_flow_tasks_ > 0) (void)post_sweep_status(false);
    return hipGetLastError();
}

void TilePlan::sym_matvec(const double* x, double* y) {
    launch_sym_tile_products(sym_tiles_, n_sym_tiles_, tiles_, x, sym_part_, stream_);
    launch_sym_tile_gather(nt_, sym_row_ptr_, sym_entries_, sym_part_, x, y, row_dot_, stream_);
}

// solve_with_pcg (explicit_schur.rs:639-756).  Per iteration: one pass over the non-zero tiles
// (k_sym_tile_products + k_sym_tile_gather, which also yields p.Ap), two fused vector kernels that keep
// alpha/beta on the device, and ONE host read-back of {p.Ap, r.r, r.z} for the reference's three
// termination tests -- read ONE ITERATION BEHIND (round 5): iteration k + 1 is enqueued before the host waits for the
// scalars of iteration k, so the device never idles through a host round trip (25 us of a 185-us iteration).  The tests are
// also made on the device (k_pcg_close_iteration): the speculative iteration behind a met test changes nothing, and x, the
// iteration count and every scalar are those of the loop that waited every time.
hipError_t TilePlan::pcg(const double* rhs, double* x, double* work, int max_iter, double tol, int* iters) {
    const int n = (int)n_pad();
    double *dg = work, *pre = work + n, *r = work + 2 * (size_t)n, *z = work + 3 * (size_t)n, *p = work + 4 * (size_t)n,
           *ap = work + 5 * (size_t)n;
    double* sc = scal_;  // [0] rz_old  [1] p.Ap  [2] r.r  [3] r.z  [4] frozen
    hipError_t e;
    if (!pcg_host_) {
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&pcg_host_), 16 * sizeof(double), hipHostMallocDefault)) != hipSuccess) return e;
        for (hipEvent_t& ev : pcg_ev_) if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return e;
    }
    launch_tile_diag(tiles_, diag_slot_, nt_, dg, stream_);
    launch_pcg_init(n, dg, rhs, pre, x, r, z, p, stream_);
    if ((e = hipMemsetAsync(sc, 0, 8 * sizeof(double), stream_)) != hipSuccess) return e;
    launch_dot(n, r, z, sc, stream_);
    launch_dot(n, r, r, sc + 2, stream_);
    double* h = pcg_host_;
    if ((e = hipMemcpyAsync(h, sc, 4 * sizeof(double), hipMemcpyDeviceToHost, stream_)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream_)) != hipSuccess) return e;
    const double abs_tol = tol * std::max(sqrt(h[2]), 1.0);
    auto enqueue_iteration = [&](int slot) -> hipError_t {
        launch_sym_tile_products(sym_tiles_, n_sym_tiles_, tiles_, p, sym_part_, stream_);
        launch_sym_tile_gather(nt_, sym_row_ptr_, sym_entries_, sym_part_, p, ap, row_dot_, stream_);
        launch_pcg_step1(n, nt_, sc, row_dot_, p, ap, pre, x, r, blk_part_, sc + 1, stream_);
        launch_pcg_step2(n, sc, blk_part_, pre, r, p, sc + 2, abs_tol, stream_);
        const hipError_t ce = hipMemcpyAsync(pcg_host_ + 8 * slot, sc, 5 * sizeof(double), hipMemcpyDeviceToHost, stream_);
        return ce != hipSuccess ? ce : hipEventRecord(pcg_ev_[slot], stream_);
    };
    int it = 0;
    if (max_iter > 0 && (e = enqueue_iteration(0)) != hipSuccess) return e;
    for (; it < max_iter; ++it) {
        if (it + 1 < max_iter && (e = enqueue_iteration((it + 1) & 1)) != hipSuccess) return e;   // on speculation
        if ((e = hipEventSynchronize(pcg_ev_[it & 1])) != hipSuccess) return e;
        h = pcg_host_ + 8 * (it & 1);
        // the device's verdict (h[4], k_pcg_close_iteration) decides -- the speculative iteration obeys the same word
        if (fabs(h[1]) < 1e-30) break;                       // p.Ap (:703-705); x was left untouched
        if (h[4] != 0.0) { ++it; break; }                    // |r| < tol (:726-728) or rz_old ~ 0 (:741-743)
    }
    *iters = it;
    return hipStreamSynchronize(stream_);   // (the speculative iteration, if any, has drained: x is final)
}

}  // namespace apex
