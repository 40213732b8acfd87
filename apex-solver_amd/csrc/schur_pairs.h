// schur_pairs.h -- the Schur reduction as a camera-pair list sorted by destination block, reduced over the lanes of a wave.
//
// Off-diagonal part of compute_schur_complement (src/linalg/sparse/explicit_schur.rs:771-925):
//     S(ci, cj) -= sum over landmarks l seen by both cameras of  W_i Hll^-1 W_j^T ,   W = Jc^T Jl  (d_c x 3).
// With N_i = Jl_i Hll^-1 (2 x 3) and M_ij = -N_i Jl_j^T (2 x 2) every pair contributes a RANK-2 update
//     -W_i Hll^-1 W_j^T = (Jc_i^T M_ij) Jc_j = U_ij V_j ,   U (d_c x 2), V (2 x d_c),
// so a block S(ci, cj) is the product of a d_c x 2P and a 2P x d_c matrix, P = pairs of the block: a tiny GEMM whose K
// dimension runs over the pairs.  All pairs of the problem are sorted by block once per structure; a wave takes 64
// consecutive pair slots, one pair per lane, rebuilds both Jacobians from the 32-byte projection records k_landmark_reduce
// writes and the staged cameras, and parks U and V in LDS; then the lanes turn into block owners: lane (g, bi, bj) keeps the
// 3 x 3 sub-block (bi, bj) of a block and adds 18 FMA per pair.  Two layouts of the list decide WHICH block:
//   form 3  the lane groups split the pairs of ONE running block (g, g + NG, ...) and are folded at a block boundary;
//   form 4  (round 4, default for nine-column cameras) every lane group owns a block of its own -- see PairQDesc below.
// A block is owned by one wave and stored ONCE with plain stores: no atomics, no LDS accumulators, no neighbour chunking
// (blocks longer than a piece and pairs on the diagonal of S add atomically).
// (v_mfma_f64_16x16x4_f64 was the first design: it occupies the fp64 datapath for 64 cycles whatever the tile holds,
// a 9 x 9 block fills 32 % of it, and it cannot overlap the linearisation's vector work: DESIGN.md section 4.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "ba_kernels.h"
#include "host_parallel.h"

namespace apex {

constexpr uint32_t kPairPad = 0xFFFFFFFFu;

struct PairRec {      // one pair slot (16 bytes, one coalesced dwordx4 per lane)
    uint32_t i, j;    // landmark-major observation indices; cam(j) <= cam(i); i == kPairPad: padding slot (U = V = 0)
    uint32_t l;       // landmark
    uint32_t blk;     // block of the pair, local to its chunk (0..31)
};
struct PairChunk {    // 64 consecutive slots = 32 K-steps of the MFMA reduction
    uint32_t mask;    // bit s: a new block starts at K-step s (slots 2s, 2s+1)
    int32_t first_block;   // global index of the block active at slot 0
};
constexpr uint32_t kPairBlockAtomic = 1;   // the block is split over several waves: flush with atomic adds
constexpr uint32_t kPairBlockDiag = 2;     // ci == cj (one camera sees a landmark twice): B + B^T into the lower triangle
struct PairBlock {
    int64_t dst;      // offset (doubles) of S(ci, cj)[0][0] inside the tile storage
    uint32_t ci, cj;  // internal camera indices
    uint32_t flags;
    uint32_t pad;
};
struct PairTask {     // the work of one wave: whole blocks, a whole number of chunks
    int32_t chunk0, nchunks;
};
// QUEUED layout (round 4, "schur_form" 4).  The seven lane groups of the product phase each own a block of their own instead
// of a seventh of the pairs of one block: a finished block is complete in the nine lanes of its group and is stored without a
// fold over the groups.  A task is the blocks of (part of) ONE row ci, each padded to whole NONETS of nine slots; the task's
// nonets, in block order, are cut into seven contiguous queues of `nchunks` nonets, and chunk q holds nonet q of every queue:
// slot g + 7 t of the chunk is pair t of queue g's nonet, slot 63 is padding.  Inside a chunk a queue therefore has exactly one
// block -- one descriptor per (chunk, queue), flush decisions per chunk, 1 + 7 cameras per chunk.
struct PairQDesc {    // queue g < 7 of a chunk: the block its nonet belongs to; entry 7: cj = the row's camera ci
    int64_t dst;      // as PairBlock::dst
    uint32_t cj;      // the block's partner camera (a valid camera index also in an empty queue)
    uint32_t flags;   // kPairBlock* | kPairQFlush: the block (piece) ends with this nonet -- store / add it after this chunk
};
// (Round 5 built the same layout for SIX-column cameras -- sixteen queues of quartets -- and measured it slower than form 3 there,
// 3.52 against 2.90 ms, profiles/r05_bench_final13682_ba6.json: the kernel side was removed in round 6; the host builder and the
// replay tests keep the (4, 16) parameters of pair_queue_len / pair_queues as a second shape of the same code.)
constexpr int pair_queue_len(int dc) { return dc == 9 ? 9 : 4; }
constexpr int pair_queues(int dc) { return dc == 9 ? 7 : 16; }
constexpr uint32_t kPairQFlush = 4;
// A block cut between the tail of queue g and the head of queue g + 1 (the same wave) is still stored once: the head part is
// CARRIED in registers from the chunk it ends in to the end of the task, where queue g JOINS it to its tail part.
constexpr uint32_t kPairQCarry = 8;     // this flush keeps the sum in the lanes (head part of a cut block)
constexpr uint32_t kPairQJoin = 16;     // this flush (last chunk of the task) adds the next queue's carried sum first
constexpr int kPairQPiecePairs = 576;   // a block with more pairs is cut into pieces of 64 nonets (atomic flush); a task has at
                                        // most 64 chunks >= its longest piece, so a piece spans at most two queues

struct PairLists {
    raw_vector<PairRec> recs;     // written once, in parallel (1.5 GB on final-13682)
    raw_vector<PairChunk> chunks; // (raw: 12 MB / 190 MB on final-13682, initialised by the parallel passes that fill them)
    std::vector<PairBlock> blocks;
    std::vector<PairTask> tasks;
    raw_vector<PairQDesc> qdesc;  // queued layout only: 8 per chunk (PairChunk::mask then holds the flush bits of its queues)
    bool queued = false;
    int64_t n_pairs = 0;      // real pairs (without padding)
    int64_t n_blocks = 0;     // camera-pair blocks that receive contributions
};

// Round 5: the RECORDS of the queued layout (97 M slots = 1.56 GB on final-13682: 0.1-0.18 s of host time plus 35 ms of upload,
// the largest single piece of apexgpu_set_structure) can be written by the DEVICE instead: the host keeps what is small and
// serial -- the blocks of every row, pieces, tasks, descriptors -- and hands over these tables; k_build_pair_recs_q walks the
// observation lists that are on the device anyway.  The order of a block's pairs is defined by that kernel (64 observations of
// the row camera at a time, their partner slots in lockstep) and the host builder follows it: the two lists are the same list.
struct PairDeviceTables {
    std::vector<int> rows;          // [n_cam] internal camera of row r (rows in the caller's camera order)
    std::vector<int> run_ptr;       // [n_cam + 1] the row's runs (= blocks of S with this row camera)
    std::vector<uint32_t> run_cj;   // partner camera, ascending inside a row
    std::vector<int> run_piece0;    // first piece of the run (a run longer than kPairQPiecePairs has several, consecutive)
    std::vector<int2> piece;        // (task, first nonet inside the task)
    std::vector<int2> task;         // (first chunk, chunks)
    int64_t n_slots = 0;
};

// Host, once per structure.  o_cam / o_pt / pt_ptr: the LOCAL landmark-major observation arrays (observations of one
// landmark sorted by internal camera index); cam_ptr / cam_obs: their camera-major view; cam_ext[ci]: the caller's index
// of internal camera ci (rows are processed in the caller's order: consecutive rows then share landmarks);
// slot: tile slot map (nt x nt, lower).
void build_pair_lists(int dc, int nt, const int* slot, int64_t n_cam, const int* cam_ext, const uint32_t* o_cam,
                      const uint32_t* o_pt, const int* pt_ptr, const int* cam_ptr, const int* cam_obs, PairLists* out,
                      int task_slots = 0 /* 0: default */, bool queued = false,
                      PairDeviceTables* dev_tables = nullptr /* queued only: leave out->recs empty and fill these instead */);
// the records of the queued layout from those tables (all pointers device memory; recs is cleared to padding first)
hipError_t launch_build_pair_recs_q(int64_t n_rows, const int* rows, const int* run_ptr, const uint32_t* run_cj, const int* run_piece0,
                                    const int2* piece, const int2* task, const int* cam_ptr, const int* cam_obs, const uint32_t* o_pt,
                                    const int* pt_ptr, const uint32_t* o_cam, PairRec* recs, int64_t n_slots, hipStream_t s);

// The pair kernel (record form: J rebuilt from the 32-byte projection records k_landmark_reduce writes, orec).
void launch_schur_pairs(int dc, const BAView& v, double* tiles, const PairTask* tasks, int n_tasks, const PairChunk* chunks,
                        const PairBlock* blocks, const PairRec* recs, const double* lmrec, hipStream_t s, const double* orec,
                        const PairQDesc* qdesc = nullptr /* the queued layout's descriptors */);

}  // namespace apex
