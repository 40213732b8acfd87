// ba_structure.h -- everything Solver::set_structure derives from the observation list on the HOST (no device call):
// internal camera order (hub cameras last, nested dissection of the tile graph), tile structure of S, landmark
// sharding, landmark-major / camera-major observation lists, and the task lists of the Schur reduction.
// Replaces, for this backend, StructureAware::initialize_structure + build_block_structure
// (src/linalg/sparse/explicit_schur.rs:1038-1062, 244-323) and build_symbolic_structure (src/linearizer/cpu/sparse.rs:54-105).
// Kept apart from the device code so that CPU tests can run it (capi: apexgpu_debug_host_structure).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "ba_kernels.h"
#include "host_parallel.h"
#include "schur_pairs.h"
#include "tile_plan.h"

namespace apex {

struct BaStructOptions {
    int dc = 9;
    bool use_nd = true;
    int nd_leaf = 16;
    double strong_edge_frac = 0.02;   // tile-graph edges lighter than this share of the weaker end's heaviest edge are left out of the ORDERING
    bool hubs_last = true;     // a vertex cover of the camera pairs that share landmarks across weakly connected tiles (hub cameras,
                               // accidental long-range matches) is ordered last: a dense border instead of dense rows everywhere
    int rank = 0, world = 1;
    bool dist_factor = true, tree_sharding = true;
    int dist_selftest = 0;
    int schur_form = 3;        // 3 sorted pair list, 4 the same pairs in the queued layout (d_c = 9; schur_pairs.h); < 0: none (matrix-free only)
    bool device_gathers = false;   // the measurement lists (o_uv, co_uv) and co_pt are gathered on the device from the caller's array and the
                                   // index lists (Solver::set_structure, single rank): the host does not build them
};

struct BaHostStructure {
    int64_t n_cam = 0, n_pt = 0, n_obs = 0;
    int dc = 9, nt = 0;
    int64_t n_c = 0, n_c_pad = 0;
    std::vector<int> cmap, cinv;       // caller's camera -> internal camera and back
    int n_hubs = 0, n_border_tiles = 1;   // border cameras (ordered last) and the tiles they occupy
    std::vector<uint8_t> present;      // nt x nt lower-triangular tile structure, final order
    std::vector<int> lmap;             // caller's landmark -> internal landmark (identity unless tree sharded)
    int64_t lm_lo = 0, lm_hi = 0;      // internal landmark range of this rank
    bool tree_shard = false;
    int pad_rank = 0;
    std::vector<uint8_t> lam_mask;     // tree sharding: cameras whose diagonal block gets lambda on this rank
    // local observation lists (this rank's landmarks)
    raw_vector<uint32_t> o_cam, o_pt;     // (raw_vector: filled from parallel loops, never value-initialised)
    raw_vector<double> o_uv;
    raw_vector<int> o_orig, cam_obs;
    std::vector<int> pt_ptr, cam_ptr;
    raw_vector<uint32_t> co_pt;
    raw_vector<double> co_uv;
    raw_vector<int> co_rank;
    int64_t n_pairs = 0, n_present = 0;
    PairLists pl;                      // the Schur reduction's task lists (schur_pairs.h)
    double seconds[6] = {0, 0, 0, 0, 0, 0};  // order + tile structure | sharding + lists | tile plan | Schur lists | uploads | total

    // Step 1: camera order, tile structure, sharding, observation lists.  Applies the partition settings to `tp`
    // (host calls only).  Returns "" or an error message.
    std::string build_lists(int64_t n_cam, int64_t n_pt, int64_t n_obs, const uint32_t* cam_idx, const uint32_t* pt_idx,
                            const double* obs_uv, const BaStructOptions& o, TilePlan& tp);
    // Step 1 in its two halves (round 5): build_order leaves the camera order and the tile structure (`present`) -- all the tile
    // plan needs, so TilePlan::build may run on another thread beside build_obs_lists, which reads `present` only through
    // tp.preview_owners and only when needs_owner_preview() (a distributed plan with tree sharding: no overlap then).
    void build_order(int64_t n_cam, int64_t n_pt, int64_t n_obs, const uint32_t* cam_idx, const uint32_t* pt_idx,
                     const BaStructOptions& o, TilePlan& tp);
    std::string build_obs_lists(const uint32_t* cam_idx, const uint32_t* pt_idx, const double* obs_uv, const BaStructOptions& o, TilePlan& tp);
    static bool needs_owner_preview(const BaStructOptions& o) { return o.world > 1 && o.dist_factor && o.tree_sharding; }
    // Step 2, after tp.build() / tp.build_symbolic(): the task lists of the selected Schur form
    // dev_tables != NULL (queued layout only): the pair RECORDS are left to the device (PairDeviceTables, schur_pairs.h)
    void build_schur_lists(const BaStructOptions& o, const int* slot_host, PairDeviceTables* dev_tables = nullptr);
    void release_scratch();   // the full-problem lists step 2 needed

   private:
    std::vector<int64_t> lp_;                    // landmark buckets of the caller's list (build_order -> build_obs_lists)
    raw_vector<int> lobs_;
    raw_vector<uint32_t> cam_i_, pt_i_;          // internal camera / landmark of every observation (caller's order)
    std::vector<int64_t> full_ptr_;
    raw_vector<int> full_obs_;
};

}  // namespace apex
