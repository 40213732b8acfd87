// schur_pairs.hip -- see schur_pairs.h
#include "schur_pairs.h"

#include <algorithm>
#include <type_traits>

#include "ba_device.hpp"
#include "host_parallel.h"

namespace apex {


constexpr int kPairTaskSlots = 1536;     // a wave's task is closed once it holds this many slots (24 chunks: the pipeline's
                                         // prologue -- three dependent loads -- is paid once per task)
constexpr int kPairQTaskSlots = 2016;    // queued layout: slots of a task (224 nonets = 7 queues x 32 chunks: the longest block of the
                                         // headline shape is 32 nonets, and a task has at least as many chunks as its longest piece;
                                         // 2016 / 2592 / 4032: 3.18 / 3.24 / 3.36 ms, smaller tasks pad their queues: 1152 3.66 ms)
constexpr int kPairMaxBlockSlots = 4096; // a block with more slots is split over several waves (atomic flush).  With 2 kTask <= 4096
                                         // a task never has more than 64 chunks: the record kernel keeps a task's chunk descriptors
                                         // one per lane (k_schur_pairs_r)

// ------------------------------------------------------------------------------------------------------------------
// host: the sorted pair list
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct Run { uint32_t cj; int len; int piece0; };   // the pairs of one row with one partner camera = one block of S
}  // namespace

void build_pair_lists(int dc, int nt, const int* slot, int64_t n_cam, const int* cam_ext, const uint32_t* o_cam,
                      const uint32_t* o_pt, const int* pt_ptr, const int* cam_ptr, const int* cam_obs, PairLists* out,
                      int task_slots, bool queued, PairDeviceTables* dev_tables) {
    SetupTrace tr;
    const int cpt = kNB / dc;
    const int kTask = std::min(task_slots > 0 ? (task_slots + 63) / 64 * 64 : kPairTaskSlots, kPairMaxBlockSlots / 2);
    // rows in the caller's camera order
    std::vector<int> rows(n_cam);
    for (int64_t c = 0; c < n_cam; ++c) rows[c] = (int)c;
    std::sort(rows.begin(), rows.end(), [&](int a, int b) { return cam_ext[a] < cam_ext[b]; });
    // pairs per row: an observation pairs with the observations BEFORE it in its landmark's list
    std::vector<int64_t> rp(n_cam + 1, 0);
    parallel_rows(n_cam, [&](int64_t r) {
        const int c = rows[r];
        int64_t n = 0;
        for (int e = cam_ptr[c]; e < cam_ptr[c + 1]; ++e) { const int i = cam_obs[e]; n += i - pt_ptr[o_pt[i]]; }
        rp[r + 1] = n;
    });
    for (int64_t r = 0; r < n_cam; ++r) rp[r + 1] += rp[r];
    const int64_t n_pairs = rp[n_cam];
    tr.mark("pairs: count");
    // Pass A, per row: its blocks = the partner cameras it has pairs with (sorted) and how many.  A counting pass over
    // the partner lists; no pair is stored yet.
    std::vector<std::vector<Run>> row_runs(n_cam);
    parallel_ranges(n_cam, 16, [&](int64_t rb, int64_t re) {
        std::vector<int> cnt(n_cam, 0), touched;
        for (int64_t r = rb; r < re; ++r) {
            const int c = rows[r];
            touched.clear();
            for (int e = cam_ptr[c]; e < cam_ptr[c + 1]; ++e) {
                const int i = cam_obs[e];
                for (int j = pt_ptr[o_pt[i]]; j < i; ++j) {
                    const int cj = (int)o_cam[j];
                    if (cnt[cj]++ == 0) touched.push_back(cj);
                }
            }
            std::sort(touched.begin(), touched.end());
            auto& runs = row_runs[r];
            runs.reserve(touched.size());
            for (int cj : touched) { runs.push_back(Run{(uint32_t)cj, cnt[cj], 0}); cnt[cj] = 0; }
        }
    });
    tr.mark("pairs: blocks of every row");
    out->blocks.clear(); out->chunks.clear(); out->tasks.clear(); out->qdesc.clear(); out->queued = false;
    if (queued) {
        // ---- QUEUED layout (schur_pairs.h): tasks inside one row, blocks padded to nonets, seven queues per task -----------
        // nonets per task: 7 x 64 at most (a task's chunk descriptors sit one per lane); fewer = more tasks per row = fewer rows
        // in flight per XCD, whose L2 then sees a row's landmark records again before they are evicted
        // (round 5) nine-column cameras: seven queues of NONETS (the lanes of a group hold the 9 x 9 block as nine 3 x 3 sub-blocks);
        // six-column cameras: sixteen queues of QUARTETS (four lanes, four sub-blocks).  QL = slots of a queue per chunk, NQ = queues.
        const int QL = pair_queue_len(dc), NQ = pair_queues(dc), ND = NQ + 1, kPiecePairs = 64 * QL;
        const int kMaxNonets = std::min(NQ * 64, std::max(63, (task_slots > 0 ? task_slots : (dc == 9 ? kPairQTaskSlots : 2048)) / QL));
        struct QPiece { int task; int nonet0; int len; int block; };
        struct QTask { int chunk0, nchunks, nonets, ci, piece0, piece1; };
        std::vector<QPiece> pieces;
        std::vector<QTask> qtasks;
        int64_t n_blocks = 0;
        for (int64_t r = 0; r < n_cam; ++r) n_blocks += (int64_t)row_runs[r].size();
        pieces.reserve(n_blocks + 16);
        out->blocks.reserve(n_blocks + 16);
        out->queued = true;
        int64_t total_chunks = 0;
        for (int64_t r = 0; r < n_cam; ++r) {
            const int ci = rows[r];
            int64_t row_nonets = 0;
            for (const Run& run : row_runs[r])
                for (int len = run.len; len > 0; len -= kPiecePairs) row_nonets += (std::min(len, kPiecePairs) + QL - 1) / QL;
            if (row_nonets == 0) continue;
            const int n_row_tasks = (int)((row_nonets + kMaxNonets - 1) / kMaxNonets);
            const int target = (int)((row_nonets + n_row_tasks - 1) / n_row_tasks);   // even tasks, cut at block boundaries
            int task_nonets = 0, task_piece0 = (int)pieces.size(), max_nn = 0;
            auto close_task = [&]() {
                if (task_nonets == 0) return;
                const int nchunks = std::max((task_nonets + NQ - 1) / NQ, max_nn);   // >= the longest piece: a piece spans <= 2 queues
                qtasks.push_back(QTask{(int)total_chunks, nchunks, task_nonets, ci, task_piece0, (int)pieces.size()});
                out->tasks.push_back(PairTask{(int32_t)total_chunks, (int32_t)nchunks});
                total_chunks += nchunks;
                task_nonets = 0; max_nn = 0; task_piece0 = (int)pieces.size();
            };
            for (Run& run : row_runs[r]) {
                const uint32_t cj = run.cj;
                const int I = ci / cpt, J = (int)cj / cpt;
                const int sl = slot[(size_t)I * nt + J];
                const int64_t dst = (int64_t)sl * kNB * kNB + (int64_t)((ci % cpt) * dc) * kNB + ((int)cj % cpt) * dc;
                const uint32_t diag = ((int)cj == ci) ? kPairBlockDiag : 0u;
                const bool split = run.len > kPiecePairs;
                run.piece0 = (int)pieces.size();
                for (int len = run.len; len > 0; len -= kPiecePairs) {
                    const int take = std::min(len, kPiecePairs), nn = (take + QL - 1) / QL;
                    if (task_nonets > 0 && (task_nonets + nn > kMaxNonets || task_nonets >= target)) close_task();
                    const int bi = (int)out->blocks.size();
                    out->blocks.push_back(PairBlock{dst, (uint32_t)ci, cj, diag | ((split || diag) ? kPairBlockAtomic : 0u), 0u});
                    pieces.push_back(QPiece{(int)qtasks.size(), task_nonets, take, bi});
                    task_nonets += nn; max_nn = std::max(max_nn, nn);
                }
            }
            close_task();
        }
        if (total_chunks * 64 > (int64_t)0x7fffffff * 64) { out->tasks.clear(); return; }
        out->chunks.resize((size_t)total_chunks);        // every chunk belongs to one task: initialised in the loop over the tasks
        out->qdesc.resize((size_t)total_chunks * ND);
        const bool host_recs = dev_tables == nullptr;     // (else the device writes the records: launch_build_pair_recs_q)
        if (host_recs) out->recs.resize((size_t)total_chunks * 64);
        tr.mark("pairs: blocks, tasks");
        // descriptors and padding, task by task; a nonet's slot t is (chunk0 + idx % nchunks) * 64 + idx / nchunks + NQ t
        parallel_rows((int64_t)qtasks.size(), [&](int64_t ti) {
            const QTask& tk = qtasks[ti];
            auto slot_of = [&](int idx, int t) { return ((int64_t)tk.chunk0 + idx % tk.nchunks) * 64 + idx / tk.nchunks + NQ * t; };
            for (int q = 0; q < tk.nchunks; ++q) {
                PairQDesc* qd = out->qdesc.data() + ((size_t)tk.chunk0 + q) * ND;
                out->chunks[(size_t)tk.chunk0 + q] = PairChunk{0u, 0};
                for (int g = 0; g < ND; ++g) qd[g] = PairQDesc{0, (uint32_t)tk.ci, 0u};
                if (host_recs)
                    for (int sl = NQ * QL; sl < 64; ++sl) out->recs[((size_t)tk.chunk0 + q) * 64 + sl] = PairRec{kPairPad, 0u, 0u, 0u};   // (d_c = 9: slot 63)
            }
            for (int pi = tk.piece0; pi < tk.piece1; ++pi) {
                const QPiece& pc = pieces[pi];
                const PairBlock& pb = out->blocks[pc.block];
                const int nn = (pc.len + QL - 1) / QL;
                const bool two_queues = pc.nonet0 / tk.nchunks != (pc.nonet0 + nn - 1) / tk.nchunks;
                for (int n = 0; n < nn; ++n) {
                    const int idx = pc.nonet0 + n, g = idx / tk.nchunks, q = idx % tk.nchunks;
                    const bool last = n == nn - 1 || q == tk.nchunks - 1;   // the piece ends, or its queue does
                    PairQDesc& d = out->qdesc[((size_t)tk.chunk0 + q) * ND + g];
                    d.dst = pb.dst; d.cj = pb.cj;
                    d.flags = pb.flags | (last ? kPairQFlush : 0u);
                    if (two_queues && last) d.flags |= (n == nn - 1) ? kPairQCarry : kPairQJoin;   // the head part ends the piece, the tail part its queue
                    if (last) out->chunks[(size_t)tk.chunk0 + q].mask |= 1u << g;
                    if (host_recs)
                        for (int t = (n == nn - 1 ? pc.len - QL * n : QL); t < QL; ++t) out->recs[slot_of(idx, t)] = PairRec{kPairPad, 0u, 0u, (uint32_t)g};
                }
            }
            if (host_recs)
                for (int idx = tk.nonets; idx < NQ * tk.nchunks; ++idx)   // the empty tail of the last queues
                    for (int t = 0; t < QL; ++t) out->recs[slot_of(idx, t)] = PairRec{kPairPad, 0u, 0u, (uint32_t)(idx / tk.nchunks)};
        }, 64);
        if (!host_recs) {
            PairDeviceTables& dt = *dev_tables;
            dt.rows = rows;
            dt.run_ptr.assign(n_cam + 1, 0);
            for (int64_t r = 0; r < n_cam; ++r) dt.run_ptr[r + 1] = dt.run_ptr[r] + (int)row_runs[r].size();
            dt.run_cj.resize((size_t)dt.run_ptr[n_cam]); dt.run_piece0.resize((size_t)dt.run_ptr[n_cam]);
            parallel_rows(n_cam, [&](int64_t r) {
                int k = dt.run_ptr[r];
                for (const Run& run : row_runs[r]) { dt.run_cj[k] = run.cj; dt.run_piece0[k] = run.piece0; ++k; }
            });
            dt.piece.resize(pieces.size());
            for (size_t i = 0; i < pieces.size(); ++i) dt.piece[i] = make_int2(pieces[i].task, pieces[i].nonet0);
            dt.task.resize(qtasks.size());
            for (size_t i = 0; i < qtasks.size(); ++i) dt.task[i] = make_int2(qtasks[i].chunk0, qtasks[i].nchunks);
            dt.n_slots = total_chunks * 64;
            tr.mark("pairs: descriptors, device tables");
            out->n_pairs = n_pairs;
            out->n_blocks = n_blocks;
            return;
        }
        // records: the k-th pair of a row with one partner goes to pair k % (64 QL) of piece k / (64 QL) of that block
        parallel_ranges(n_cam, 16, [&](int64_t rb, int64_t re) {
            std::vector<int> pos(n_cam, 0), ridx(n_cam, 0);
            for (int64_t r = rb; r < re; ++r) {
                const int c = rows[r];
                const auto& runs = row_runs[r];
                for (size_t q = 0; q < runs.size(); ++q) { ridx[runs[q].cj] = (int)q; pos[runs[q].cj] = 0; }
                // The order of a block's pairs is the DEVICE builder's (k_build_pair_recs_q: a wave takes 64 observations of the row
                // camera at a time and walks their partner slots s = 0, 1, ... in lockstep, lanes in order inside a step), so that a
                // list built here and a list built there are the same list, slot for slot (round 5).
                for (int e0 = cam_ptr[c]; e0 < cam_ptr[c + 1]; e0 += 64) {
                    const int e1 = std::min(e0 + 64, cam_ptr[c + 1]);
                    int maxnp = 0;
                    for (int e = e0; e < e1; ++e) { const int i = cam_obs[e]; maxnp = std::max(maxnp, i - pt_ptr[o_pt[i]]); }
                    for (int sp = 0; sp < maxnp; ++sp)
                        for (int e = e0; e < e1; ++e) {
                            const int i = cam_obs[e];
                            const uint32_t l = o_pt[i];
                            const int b = pt_ptr[l];
                            if (sp >= i - b) continue;
                            const int j = b + sp;
                            const uint32_t cj = o_cam[j];
                            const Run& run = runs[ridx[cj]];
                            const int k = pos[cj]++;
                            const QPiece& pc = pieces[run.piece0 + k / kPiecePairs];
                            const QTask& tk = qtasks[pc.task];
                            const int kk = k % kPiecePairs, idx = pc.nonet0 + kk / QL, g = idx / tk.nchunks;
                            out->recs[((int64_t)tk.chunk0 + idx % tk.nchunks) * 64 + g + NQ * (kk % QL)] = PairRec{(uint32_t)i, (uint32_t)j, l, (uint32_t)g};
                        }
                }
            }
        });
        tr.mark("pairs: records");
        // (A workgroup is four consecutive tasks and lives as long as its longest one; sorting the tasks by length inside
        // windows of 8 / 32 evens the four out -- 96 % -> 99 % of the wave slots busy -- and LOSES 0.07 / 0.1 ms: consecutive
        // tasks are the same row, and four waves of a CU on one row share its landmark lines in the L1.  Row order kept.)
        out->n_pairs = n_pairs;
        out->n_blocks = n_blocks;
        return;
    }
    // ---- serial pass over the BLOCKS (not the pairs): slot offsets, block table, chunk descriptors, tasks ------------
    struct Piece { int64_t slot0; int len; };   // a block (or a piece of a split block) -> its slots
    std::vector<Piece> pieces;
    int64_t n_blocks = 0;
    for (int64_t r = 0; r < n_cam; ++r) n_blocks += (int64_t)row_runs[r].size();
    pieces.reserve(n_blocks + 16);
    out->blocks.reserve(n_blocks + 16);
    out->chunks.reserve((size_t)(n_pairs / 64 + n_blocks / 32 + 1024));
    int64_t slot_pos = 0, task_begin = 0;
    auto chunk_touch = [&](int64_t s0, int64_t s1, int block_index, bool starts) {
        const int64_t c1 = (s1 - 1) / 64;
        if ((int64_t)out->chunks.size() <= c1) out->chunks.resize(c1 + 1, PairChunk{0u, -1});
        for (int64_t c = s0 / 64; c <= c1; ++c)
            if (out->chunks[c].first_block < 0) out->chunks[c].first_block = block_index;
        if (starts) out->chunks[s0 / 64].mask |= 1u << ((s0 % 64) / 2);
    };
    auto close_task = [&]() {
        if (slot_pos == task_begin) return;
        slot_pos = (slot_pos + 63) / 64 * 64;
        out->tasks.push_back(PairTask{(int32_t)(task_begin / 64), (int32_t)((slot_pos - task_begin) / 64)});
        task_begin = slot_pos;
    };
    for (int64_t r = 0; r < n_cam; ++r) {
        const int ci = rows[r];
        for (Run& run : row_runs[r]) {
            const uint32_t cj = run.cj;
            const int I = ci / cpt, J = (int)cj / cpt;
            const int sl = slot[(size_t)I * nt + J];
            const int64_t dst = (int64_t)sl * kNB * kNB + (int64_t)((ci % cpt) * dc) * kNB + ((int)cj % cpt) * dc;
            const uint32_t diag = ((int)cj == ci) ? kPairBlockDiag : 0u;
            int64_t len = run.len;
            const bool split = (len + 1) / 2 * 2 > kPairMaxBlockSlots;
            if (split) close_task();
            run.piece0 = (int)pieces.size();
            while (len > 0) {
                const int take = (int)std::min<int64_t>(len, split ? kPairMaxBlockSlots : len);
                const int padded = (take + 1) / 2 * 2;
                if (!split && slot_pos - task_begin > 0 && slot_pos - task_begin + padded > 2 * kTask) close_task();
                const int bi = (int)out->blocks.size();
                out->blocks.push_back(PairBlock{dst, (uint32_t)ci, cj, diag | ((split || diag) ? kPairBlockAtomic : 0u), 0u});
                pieces.push_back(Piece{slot_pos, take});
                chunk_touch(slot_pos, slot_pos + padded, bi, true);
                slot_pos += padded;
                len -= take;
                if (split || slot_pos - task_begin >= kTask) close_task();
            }
        }
    }
    close_task();
    const int64_t n_slots = slot_pos;
    out->chunks.resize(n_slots / 64, PairChunk{0u, -1});
    tr.mark("pairs: blocks, tasks");
    // ---- records, written straight to their slots.  Pass B, per row: the observations of a camera are visited in
    // increasing landmark-major index i, so the pairs of one partner arrive ordered by i: a cursor per partner is the
    // whole sort.  Every slot is written exactly once (pairs here; the odd block's zero pair and the padding behind a
    // task in the loop over the pieces).
    out->recs.resize((size_t)n_slots);
    parallel_ranges(n_cam, 16, [&](int64_t rb, int64_t re) {
        std::vector<int> pos(n_cam, 0), ridx(n_cam, 0);
        for (int64_t r = rb; r < re; ++r) {
            const int c = rows[r];
            const auto& runs = row_runs[r];
            for (size_t q = 0; q < runs.size(); ++q) { ridx[runs[q].cj] = (int)q; pos[runs[q].cj] = 0; }
            for (int e = cam_ptr[c]; e < cam_ptr[c + 1]; ++e) {
                const int i = cam_obs[e];
                const uint32_t l = o_pt[i];
                for (int j = pt_ptr[l]; j < i; ++j) {
                    const uint32_t cj = o_cam[j];
                    const Run& run = runs[ridx[cj]];
                    const int k = pos[cj]++;
                    const int pi = run.piece0 + k / kPairMaxBlockSlots;   // (only a split block has more than one piece)
                    const int64_t s = pieces[pi].slot0 + k % kPairMaxBlockSlots;
                    out->recs[s] = PairRec{(uint32_t)i, (uint32_t)j, l, (uint32_t)(pi - out->chunks[s / 64].first_block)};
                }
            }
        }
    });
    const int64_t n_pieces = (int64_t)pieces.size();
    parallel_rows(n_pieces, [&](int64_t b) {
        const Piece& pc = pieces[b];
        int64_t s = pc.slot0 + pc.len;
        if (pc.len & 1) {   // the odd block's last K-step: a zero pair that still belongs to the block
            out->recs[s] = PairRec{kPairPad, 0u, 0u, (uint32_t)((int)b - out->chunks[s / 64].first_block)};
            ++s;
        }
        const int64_t next = b + 1 < n_pieces ? pieces[b + 1].slot0 : n_slots;
        for (; s < next; ++s) out->recs[s] = PairRec{kPairPad, 0u, 0u, 0u};
    }, 1024);
    tr.mark("pairs: records");
    out->n_pairs = n_pairs;
    out->n_blocks = n_blocks;
}

// ------------------------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------------------------
// The records of the queued layout written by the device (round 5; PairDeviceTables).  One WAVE per row of S: it walks the row
// camera's observations 64 at a time, in order; an observation i pairs with the observations j < i of its landmark, pair s of
// every lane in turn.  The k-th pair of the row with partner cj goes to pair k % 576 of piece k / 576 of that block: k is a
// running count per partner (LDS; rows with more partners than kRecsLdsPartners count in their piece of a global scratch array,
// cleared by the host), advanced in LANE order inside a step -- the lanes that meet the same partner are found with ballots --,
// so the list is a fixed function of the observation lists, the same at every build.
constexpr int kRecsLdsPartners = 2048;
__global__ __launch_bounds__(64) void k_build_pair_recs_q(int64_t n_rows, const int* __restrict__ rows, const int* __restrict__ run_ptr,
                                                           const uint32_t* __restrict__ run_cj, const int* __restrict__ run_piece0,
                                                           const int2* __restrict__ piece, const int2* __restrict__ task,
                                                           const int* __restrict__ cam_ptr, const int* __restrict__ cam_obs,
                                                           const uint32_t* __restrict__ o_pt, const int* __restrict__ pt_ptr,
                                                           const uint32_t* __restrict__ o_cam, int* __restrict__ cnt_global,
                                                           PairRec* __restrict__ recs, int QL, int NQ) {
    __shared__ int cnt_lds[kRecsLdsPartners];
    const int64_t r = blockIdx.x;
    if (r >= n_rows) return;
    const int lane = threadIdx.x;
    const int c = rows[r], r0 = run_ptr[r], P = run_ptr[r + 1] - r0;
    if (P == 0) return;
    const bool in_lds = P <= kRecsLdsPartners;
    int* cnt = in_lds ? cnt_lds : cnt_global + r0;
    if (in_lds) for (int p = lane; p < P; p += 64) cnt_lds[p] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint32_t* cjs = run_cj + r0;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int e0 = cam_ptr[c]; e0 < cam_ptr[c + 1]; e0 += 64) {
        const int e = e0 + lane;
        const bool act = e < cam_ptr[c + 1];
        const int i = act ? cam_obs[e] : 0;
        const uint32_t l = act ? o_pt[i] : 0u;
        const int b = act ? pt_ptr[l] : 0;
        const int np = act ? i - b : 0;
        int maxnp = np;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxnp = max(maxnp, __shfl_xor(maxnp, off, 64));
        for (int s = 0; s < maxnp; ++s) {
            const bool has = s < np;
            const int j = b + (has ? s : 0);
            const uint32_t cj = has ? o_cam[j] : 0u;
            int p = 0;
            if (has) {   // the partner's run: cj is in the list (binary search over the row's ascending partners)
                int lo = 0, hi = P - 1;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (cjs[mid] < cj) lo = mid + 1; else hi = mid; }
                p = lo;
            }
            int k = 0;
            unsigned long long todo = __ballot(has);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int pl = __builtin_amdgcn_readlane(p, leader);
                const unsigned long long same = __ballot(has && p == pl) & todo;
                // (the global counters of a row with very many partners are read and written past the L1: a plain load behind
                // another lane's store of an earlier step could hit a stale line)
                const int base = in_lds ? cnt[pl] : __hip_atomic_load(cnt + pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (has && p == pl) k = base + __popcll(same & lt_mask);
                __builtin_amdgcn_wave_barrier();
                if (lane == leader) {
                    if (in_lds) cnt[pl] = base + __popcll(same);
                    else __hip_atomic_store(cnt + pl, base + __popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_wave_barrier();
                todo &= ~same;
            }
            if (has) {
                const int piece_pairs = 64 * QL;
                const int2 pc = piece[run_piece0[r0 + p] + k / piece_pairs];
                const int2 tk = task[pc.x];
                const int kk = k % piece_pairs, idx = pc.y + kk / QL, g = idx / tk.y;
                PairRec rec{(uint32_t)i, (uint32_t)j, l, (uint32_t)g};
                recs[((int64_t)tk.x + idx % tk.y) * 64 + g + NQ * (kk % QL)] = rec;
            }
        }
    }
}
hipError_t launch_build_pair_recs_q(int64_t n_rows, const int* rows, const int* run_ptr, const uint32_t* run_cj, const int* run_piece0,
                                    const int2* piece, const int2* task, const int* cam_ptr, const int* cam_obs, const uint32_t* o_pt,
                                    const int* pt_ptr, const uint32_t* o_cam, PairRec* recs, int64_t n_slots, hipStream_t s) {
    if (n_rows <= 0 || n_slots <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(recs, 0xFF, (size_t)n_slots * sizeof(PairRec), s);   // i = kPairPad everywhere: padding unless written below
    if (e != hipSuccess) return e;
    // rows with more partners than the LDS counters hold count in global memory; their number is known only on the device side
    // of the tables, so the scratch covers every run (4 bytes each: 3.4 MB on final-13682) and lives for this call
    int n_runs = 0;
    e = hipMemcpyAsync(&n_runs, run_ptr + n_rows, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    int* scratch = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&scratch), (size_t)std::max(n_runs, 1) * sizeof(int));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(scratch, 0, (size_t)std::max(n_runs, 1) * sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_build_pair_recs_q, dim3((unsigned)n_rows), dim3(64), 0, s, n_rows, rows, run_ptr, run_cj, run_piece0, piece, task,
                           cam_ptr, cam_obs, o_pt, pt_ptr, o_cam, scratch, recs, pair_queue_len(9), pair_queues(9));
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void)hipFree(scratch);
    return e;
}

constexpr int kPairDmaBlocks = 4;   // a chunk with at most this many blocks stages its <= 8 cameras in LDS (one 16-byte piece per lane)

// Camera staging: the 8 lanes t = 8 c .. 8 c + 7 fetch the 128-byte prepared camera c of the chunk (camera c & 1 of block
// c >> 1) a chunk ahead; chunks of many tiny blocks read their cameras from memory instead.
__device__ __forceinline__ uint32_t pairs_dma_cam(const PairBlock* __restrict__ blocks, const PairChunk ck, int lane) {
    const int nblk = 1 + __popc(ck.mask & ~1u);
    const int c = lane >> 3;
    const PairBlock* pb = blocks + ck.first_block + min(c >> 1, nblk - 1);
    return (c & 1) ? pb->cj : pb->ci;
}

// ------------------------------------------------------------------------------------------------------------------
// RECORD FORM (round 3).  The fused kernels above re-linearise both observations of every pair from the 24-byte
// observation records: ~140 fp64 instructions per observation, k - 1 = 5.5 times per observation and iteration -- 40 % of
// the kernel's vector instructions, and it is bound by exactly those.  k_landmark_reduce linearises
// every observation once anyway; it now also writes the observation's PROJECTION RECORD (xn, yn, p_w.z, sqrt(rho'); -1/z until the end of round 4), 32
// bytes, in place of which the pair kernel used to gather the 16-byte measurement.  From the record and the camera
// (R, t, f, k1, k2: staged in LDS as before) and the point the Jacobian is ~70 instructions, one reciprocal (-1/z, rebuilt
// with linearize_obs' own operations: rec_inz), no square root:
//     a = d(u,v)/d p_w = w f (-1/z) [dxx dxy xn dxx + yn dxy ; dxy dyy xn dxy + yn dyy] R          (2 x 3, = Jl)
//     Jc = [ a | -a [p_w]x | (xn w, yn w)^T (dist, f r2, f r4) ]                                    (2 x 9)
// and the row side never forms Jc at all: U = Jc_i^T M = [ G ; p_w x G ; t (s M) ] with G = a_i^T M  (34 instead of 66).
// Same lists, same product phase, same flush.  Traffic: 16 more bytes per observation gathered (the record and the
// landmark record are the whole gather; no camera / observation index arrays), 0.93 GB more written by k_landmark_reduce.
// ------------------------------------------------------------------------------------------------------------------
// (RecJac, jac_from_rec: ba_device.hpp -- the landmark-major kernels rebuild J from the same records)

// One PAIR per lane, 64 pairs per step (two waves per SIMD by LDS: 18.4 KB of U / V per wave).
// MASKED: OptimizeParams modes that drop a column group (mask code in slot 15 of the camera); the default modes never pay for it.
// The finished block: the NG groups' partial sums are folded into group 0 and stored.  Fast path (a block owned by one wave,
// off the diagonal of S: all but a handful): nine plain stores by the nine lanes of group 0, no branches per element.
struct Acc9 { double a0, a1, a2, a3, a4, a5, a6, a7, a8; };   // by value: an array argument would pin acc[] to scratch memory
template <int DC>
__device__ __noinline__ void pairs_flush_slow(double* __restrict__ dst, const uint32_t flags, const Acc9 av, int sub) {
    constexpr int NB3 = DC / 3;
    const int bi = sub / NB3, bj = sub % NB3;
    const double acc[9] = {av.a0, av.a1, av.a2, av.a3, av.a4, av.a5, av.a6, av.a7, av.a8};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int row = 3 * bi + r, col = 3 * bj + c;
            const double val = acc[3 * r + c];
            if (flags & kPairBlockDiag) {   // B + B^T, kept in the lower triangle of the diagonal block
                if (row >= col) unsafeAtomicAdd(&dst[row * kNB + col], val);
                if (col >= row) unsafeAtomicAdd(&dst[col * kNB + row], val);
            } else {
                unsafeAtomicAdd(&dst[row * kNB + col], val);
            }
        }
}
// ... the same for a lane that holds COLUMN col of the finished block (queued layout: av.a0 .. a8 = rows 0 .. 8)
template <int DC>
__device__ __noinline__ void pairs_flush_slow_col(double* __restrict__ dst, const uint32_t flags, const Acc9 av, int col) {
    const double acc[9] = {av.a0, av.a1, av.a2, av.a3, av.a4, av.a5, av.a6, av.a7, av.a8};
#pragma unroll
    for (int row = 0; row < 9; ++row) {
        const double val = acc[row];
        if (flags & kPairBlockDiag) {
            if (row >= col) unsafeAtomicAdd(&dst[row * kNB + col], val);
            if (col >= row) unsafeAtomicAdd(&dst[col * kNB + row], val);
        } else {
            unsafeAtomicAdd(&dst[row * kNB + col], val);
        }
    }
}
// Lane mapping of the record kernel's product phase.  DC = 6: lane = 4 g + sub (16 groups x 4 sub-blocks).  DC = 9: the 9
// sub-blocks x 7 groups are laid out so that the fold over the groups is DPP arithmetic on the vector unit instead of three
// dependent round trips through the LDS crossbar (ds_bpermute): sub-blocks 0..7 own one aligned OCTET of lanes each
// (lane = 8 sub + g, g = 0..6) and sub-block 8 takes the octets' eighth lanes (lane = 8 g + 7; lane 63 idles and ends up
// holding sub-block 8's total).
template <int DC>
__device__ __forceinline__ void pairs_lane_map(int lane, int& g, int& sub) {
    if (DC == 9) {
        const bool eighth = (lane & 7) == 7;
        sub = eighth ? 8 : lane >> 3;
        g = eighth ? lane >> 3 : lane & 7;      // (g == 7 only for lane 63: not a worker)
    } else {
        g = lane >> 2; sub = lane & 3;
    }
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_add_masked(double x) {   // x + (x moved by CTRL), lanes outside the masks add 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, BANK_MASK, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, BANK_MASK, true);
    return x + __hiloint2double(hi, lo);
}
template <int DC>
__device__ __forceinline__ void pairs_flush2(double* __restrict__ tiles, const int64_t pb_dst, const uint32_t pb_flags,
                                             double acc[9], int lane) {
    constexpr int NB3 = DC / 3, GL = NB3 * NB3;
    constexpr int NG = (DC == 9) ? 7 : 16, P2 = (DC == 9) ? 8 : 16;
    int g, sub;
    pairs_lane_map<DC>(lane, g, sub);
    bool storer;
    if (DC == 9) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            // sub-blocks 0..7: sum over the octet's lanes 0..6 into lane 6 (row_shr 4, 2, 1 into the upper half-octets only)
            double a = acc[k];
            a = dpp_add_masked<0x114, 0xF, 0xA>(a);
            a = dpp_add_masked<0x112, 0xF, 0xA>(a);
            a = dpp_add_masked<0x111, 0xF, 0xA>(a);
            // sub-block 8: lanes 7, 15, ..., 55 (63 holds 0) into lane 63: row_shr 8, then row_bcast 15 / 31
            double b = acc[k];
            b = dpp_add_masked<0x118, 0xF, 0xF>(b);
            b = dpp_add_masked<0x142, 0xA, 0xF>(b);
            b = dpp_add_masked<0x143, 0xC, 0xF>(b);
            acc[k] = lane == 63 ? b : a;
        }
        storer = lane == 63 || (lane & 7) == 6;
        sub = lane == 63 ? 8 : lane >> 3;
    } else {
#pragma unroll
        for (int st = P2 / 2; st >= 1; st >>= 1) {
            const bool take = g < st && g + st < NG;
            const int src = take ? lane + st * GL : lane;
            double other[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) other[k] = __shfl(acc[k], src, 64);
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] += take ? other[k] : 0.0;
        }
        storer = lane < GL;
    }
    if (storer) {
        double* dst = tiles + pb_dst;
        if (pb_flags == 0) {
            const int bi = sub / NB3, bj = sub - bi * NB3;
            double* d0 = dst + (3 * bi) * kNB + 3 * bj;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) d0[r * kNB + c] = acc[3 * r + c];
        } else {
            pairs_flush_slow<DC>(dst, pb_flags, Acc9{acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], acc[8]}, sub);
        }
    }
}

// Nothing in the loop goes through the scalar memory path: s_load shares the lgkm counter with the LDS and returns out of
// order, so one descriptor load in flight turns every LDS wait of the product loop into a wait for memory (the first
// version of this kernel stalled ~1 us per chunk on its own chunk descriptor).  The task's chunk descriptors are loaded
// once, one per lane, and read with v_readlane; a chunk's block descriptors (destination, flags) come one chunk ahead as
// a vector load by the first lanes and are read the same way.
// QL: the QUEUED layout (schur_pairs.h): every lane group owns a block of its own -- nine product steps per chunk without
// segment bookkeeping, no fold over the groups, one descriptor per (chunk, queue) fetched by the lanes that store.
// (The per-phase ablations and cycle stamps that rounds 3-5 timed this kernel with -- no gathers, no products, no stores, no
// transposition ... -- are gone from the source: their numbers are profiles/r03_pairs_ablation.txt, r04_pairs_queued_*.txt,
// r05_pairs_bundle_ablation.txt; the code is in the history, last at the round-5 head.)
template <int DC, bool MASKED, bool QL = false>
__global__ __launch_bounds__(256, 2) void k_schur_pairs_r(BAView v, double* __restrict__ tiles, const PairTask* __restrict__ tasks,
                                                          int n_tasks, const PairChunk* __restrict__ chunks,
                                                          const PairBlock* __restrict__ blocks, const PairRec* __restrict__ recs,
                                                          const double* __restrict__ lmrec, const double* __restrict__ orec,
                                                          const PairQDesc* __restrict__ qdesc) {
    // the queued layout (d_c = 9 only): seven groups of nine lanes, nonets
    static_assert(!QL || DC == 9, "the queued layout exists for nine-column cameras");
    constexpr int NQ = pair_queues(DC), ND = NQ + 1;
    constexpr int NCAMS = 8;   // cameras staged per chunk
    constexpr int UV = 2 * DC;
    constexpr int NB3 = DC / 3;
    constexpr int GL = NB3 * NB3;
    constexpr int NG = (DC == 9) ? 7 : 16;
    constexpr int NACC = 9;
    constexpr int WAVE_LDS = 2 * 64 * UV + UV;   // U[64][UV] | V[64][UV] | zeros[UV]
    __shared__ double lds_all[4 * WAVE_LDS];
    // The chunk's <= 8 cameras are staged through REGISTERS (one 16-byte load per lane a chunk ahead, one ds_write at the top
    // of the chunk), not by LDS-DMA as in the fused kernels above: global_load_lds writes LDS behind the VECTOR-MEMORY
    // counter, the compiler cannot tell its destination from U / V and puts s_waitcnt vmcnt(0) in front of every LDS read of
    // the product loop -- which then waits for the very gathers that were issued to overlap with it (the fused kernels have
    // exactly this: their prefetch never overlapped their products).
    // (round 6) a staged camera takes kCamLds = 18 doubles, not kCamStride = 16: the lanes of a chunk read the cameras of seven
    // different queues, and at a stride of 32 dwords those addresses fall on two bank groups -- every ds_read_b128 of a camera took
    // 16 LDS cycles instead of 4 (tools/lds_conflict_sim.py: 96 of a chunk's 556 LDS cycles)
    constexpr int kCamLds = kCamStride + 2;
    __shared__ double lds_cams[4 * NCAMS * kCamLds];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int wg = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    const int t = wg * 4 + w;
    if (t >= n_tasks) return;
    double* U = lds_all + w * WAVE_LDS;
    double* V = U + 64 * UV;
    double* CAMS = lds_cams + w * NCAMS * kCamLds;
    double* Z = V + 64 * UV;
    if (lane < UV) Z[lane] = 0.0;
    int g, sub;
    pairs_lane_map<DC>(lane, g, sub);
    // (lane 63 has no sub-block of its own: it shadows lane 62 -- queue 6, sub-block 8 -- and never stores.  Lane 62, not lane 0
    // as until round 5: a shadow of lane 0 read row 7t of U / V from inside the lane group that reads rows 4 + 7t .. 6 + 7t --
    // a different address on busy banks, one conflict cycle on every ds_read_b128 of the product phase, 54 of a chunk's 270 read
    // cycles (tools/lds_conflict_sim.py); beside lane 62 it reads lane 62's address and rides on the broadcast)
    if (QL) { g = lane == 63 ? 6 : lane / 9; sub = lane == 63 ? 8 : lane - 9 * g; }
    const int bi = sub / NB3, bj = sub - bi * NB3;
    const bool worker = g < NG;
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
    int cur = -1;
    int64_t cur_dst = 0;
    uint32_t cur_flags = 0;
    double pend[QL ? 9 : 1];          // queued layout: a finished block on its way to memory (stored a chunk later)
    double carry[QL ? 9 : 1];         // ... and the head part of a block cut between two queues (see kPairQCarry)
#pragma unroll
    for (int k = 0; k < (QL ? 9 : 1); ++k) carry[k] = 0.0;
    int2 pend_dst = make_int2(0, 0);
    uint32_t pend_fl = 0;
    bool pend_on = false;
    auto store_pending = [&]() {   // pend[r] = element (r, sub) of the block: store instruction r writes one whole 72-byte row per group
        if (QL && pend_on) {
            double* dst = tiles + (((int64_t)pend_dst.y << 32) | (uint32_t)pend_dst.x);
            const uint32_t fl = pend_fl & (kPairBlockAtomic | kPairBlockDiag);
#define PEND_ACC9 Acc9{pend[0], pend[1 % (QL ? 9 : 1)], pend[2 % (QL ? 9 : 1)], pend[3 % (QL ? 9 : 1)], pend[4 % (QL ? 9 : 1)], pend[5 % (QL ? 9 : 1)], \
                      pend[6 % (QL ? 9 : 1)], pend[7 % (QL ? 9 : 1)], pend[8 % (QL ? 9 : 1)]}
            if (fl == 0) {
#pragma unroll
                for (int r = 0; r < 9; ++r) dst[r * kNB + sub] = pend[r % (QL ? 9 : 1)];
            } else {
                pairs_flush_slow_col<DC>(dst, fl, PEND_ACC9, sub);
            }
#undef PEND_ACC9
        }
        pend_on = false;
    };
    const double mp = MASKED ? ((v.mask_code & 4) ? 1.0 : 0.0) : 1.0, ml = MASKED ? ((v.mask_code & 2) ? 1.0 : 0.0) : 1.0,
                 mi = MASKED ? ((v.mask_code & 1) ? 1.0 : 0.0) : 1.0;

    // ---- the task and its chunk descriptors (vector loads, read back with v_readlane) --------------------------------------
    int chunk0, nchunks;
    {
        const int2 tk = reinterpret_cast<const int2*>(tasks)[t];   // the same address in every lane: one request
        chunk0 = __builtin_amdgcn_readfirstlane(tk.x); nchunks = __builtin_amdgcn_readfirstlane(tk.y);
    }
    const uint2* chunk2 = reinterpret_cast<const uint2*>(chunks);
    // chunk chunk0 + lane of the task (a task has at most 64 chunks: build_pair_lists).  Loaded ONCE, before the loop: a
    // reload inside the loop -- even on a path that is never taken -- makes the compiler wait for every outstanding load
    // (s_waitcnt vmcnt(0)) before each v_readlane of these registers, i.e. for the gathers it has just issued.
    const uint2 ckv = chunk2[(size_t)chunk0 + min(lane, nchunks - 1)];
    auto chunk_desc = [&](int q) -> PairChunk {                      // q: task-relative chunk index
        PairChunk c;
        c.mask = (uint32_t)__builtin_amdgcn_readlane((int)ckv.x, q);
        c.first_block = __builtin_amdgcn_readlane((int)ckv.y, q);
        return c;
    };
    // The per-pair gathers (the first 64-byte line of the landmark record, the two projection records 32 B each) are COOPERATIVE: the L1 serves one
    // 64-byte line per clock whatever the lanes take from it, and a lane that fetches its own 160 bytes as ten 16-byte loads
    // costs ten line accesses per pair -- 1.6 ms of pure tag-lookup time per launch, the largest single item of the first
    // record kernel (profiles/r03_pairs_ablation.txt).  Here four lanes share a 64-byte line in ONE instruction (two lanes a
    // 32-byte record): three line accesses per pair, eight gather instructions per chunk.  The pieces land in the registers of
    // the lanes that fetched them and reach the lanes that need them by a register transpose (unstage_dpp).
    struct Coop { double2 a0, a1, a2, a3, i0, i1, j0, j1; };   // what a lane fetches for OTHER lanes' pairs (scalars: arrays in a loop-carried struct went to scratch memory)
    struct Gather { double2 ri0, ri1, rj0, rj1; double2 lm[4]; };  // a lane's own pair
    // Which lane fetches what: the four lanes of a quad fetch the four quarters of the landmark line of the quad's k-th pair
    // (k = 0..3: four instructions, 16 lines each), the two lanes of a lane pair the two halves of a 32-byte piece of the
    // lane pair's k-th pair (k = 0, 1).  The indices therefore come from a lane of the same quad: one DPP quad_perm move each,
    // no trip through the LDS crossbar.
    auto quad_bcast = [&](int x, auto sel) -> int {
        constexpr int k = decltype(sel)::value;
        return __builtin_amdgcn_mov_dpp(x, k | (k << 2) | (k << 4) | (k << 6), 0xf, 0xf, true);          // quad_perm:[k,k,k,k]
    };
    auto pair_bcast = [&](int x, auto sel) -> int {
        constexpr int k = decltype(sel)::value;
        return __builtin_amdgcn_mov_dpp(x, k | (k << 2) | ((2 + k) << 4) | ((2 + k) << 6), 0xf, 0xf, true);   // quad_perm:[k,k,2+k,2+k]
    };
    auto issue = [&](const uint4 rr, Coop& d) {
        const bool valid = rr.x != kPairPad;
        // padding lanes read element 0
        const int i = valid ? (int)rr.x : 0, j = valid ? (int)rr.y : 0, l = valid ? (int)rr.z : 0;
        const int qq = lane & 3, h = lane & 1;
        auto lm0 = [&](auto sel) { return *reinterpret_cast<const double2*>(lmrec + (size_t)kLmStride * (size_t)quad_bcast(l, sel) + 2 * qq); };
        d.a0 = lm0(std::integral_constant<int, 0>{}); d.a1 = lm0(std::integral_constant<int, 1>{});
        d.a2 = lm0(std::integral_constant<int, 2>{}); d.a3 = lm0(std::integral_constant<int, 3>{});
        {
            const std::integral_constant<int, 0> k{};
            d.i0 = *reinterpret_cast<const double2*>(orec + 4 * (size_t)pair_bcast(i, k) + 2 * h);
            d.j0 = *reinterpret_cast<const double2*>(orec + 4 * (size_t)pair_bcast(j, k) + 2 * h);
        }
        {
            const std::integral_constant<int, 1> k{};
            d.i1 = *reinterpret_cast<const double2*>(orec + 4 * (size_t)pair_bcast(i, k) + 2 * h);
            d.j1 = *reinterpret_cast<const double2*>(orec + 4 * (size_t)pair_bcast(j, k) + 2 * h);
        }
    };
    // From the lanes that fetched the pieces to the lanes that need them, in registers (round 4; through LDS before: ten
    // writes, ten reads and their round trip, 0.14 ms): the pieces form a 4 x 4 matrix per quad (lane q holds quarter q of the
    // quad's pair k in a_k) and 2 x 2 matrices per lane pair (half h of pair k in i_k / j_k); a lane wants the row of its own
    // pair.  A butterfly transpose -- exchange with the lane one away, then two away, each a DPP quad_perm move folded into a
    // select.
    auto unstage_dpp = [&](const Coop& d, Gather& o) {
        const bool odd = (lane & 1) != 0, hi2 = (lane & 2) != 0;
        constexpr int X1 = 1 | (0 << 2) | (3 << 4) | (2 << 6);   // quad_perm:[1,0,3,2]
        constexpr int X2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);   // quad_perm:[2,3,0,1]
        auto xch = [&](const bool take, const double2 own, const double2 other, auto ctrl) -> double2 {   // take ? other[partner lane] : own
            constexpr int C = decltype(ctrl)::value;
            const double v[2] = {other.x, other.y};
            double r[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v[k]), C, 0xf, 0xf, true);
                const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v[k]), C, 0xf, 0xf, true);
                r[k] = __hiloint2double(hi, lo);
            }
            return make_double2(take ? r[0] : own.x, take ? r[1] : own.y);
        };
        const std::integral_constant<int, X1> c1{};
        const std::integral_constant<int, X2> c2{};
        const double2 b0 = xch(odd, d.a0, d.a1, c1), b1 = xch(!odd, d.a1, d.a0, c1);
        const double2 b2 = xch(odd, d.a2, d.a3, c1), b3 = xch(!odd, d.a3, d.a2, c1);
        o.lm[0] = xch(hi2, b0, b2, c2); o.lm[2] = xch(!hi2, b2, b0, c2);
        o.lm[1] = xch(hi2, b1, b3, c2); o.lm[3] = xch(!hi2, b3, b1, c2);
        o.ri0 = xch(odd, d.i0, d.i1, c1); o.ri1 = xch(!odd, d.i1, d.i0, c1);
        o.rj0 = xch(odd, d.j0, d.j1, c1); o.rj1 = xch(!odd, d.j1, d.j0, c1);
    };
    // block descriptors of a chunk: lane b < nblk holds block first_block + b (destination offset, flags)
    struct BlockDesc { int2 dst; uint32_t flags; };
    auto load_blocks = [&](const PairChunk c, BlockDesc& b, int qrel) {
        if (QL) {   // the descriptor of this lane's queue in chunk chunk0 + qrel: what it stores to after that chunk
            const uint4 d = *reinterpret_cast<const uint4*>(qdesc + ND * (size_t)(chunk0 + qrel) + g);
            b.dst = make_int2((int)d.x, (int)d.y); b.flags = d.w;
            return;
        }
        const int nblk = 1 + __popc(c.mask & ~1u);
        const PairBlock* pb = blocks + c.first_block + min(lane, nblk - 1);
        b.dst = *reinterpret_cast<const int2*>(&pb->dst);
        b.flags = pb->flags;
    };
    // the chunk's cameras, eight lanes each: queued layout = the row's camera (entry 7) and the seven queues' partners
    auto chunk_cam = [&](const PairChunk c, int qrel) -> uint32_t {
        if (QL) { const int cc = lane >> 3; return qdesc[ND * (size_t)(chunk0 + qrel) + (cc == 0 ? NQ : cc - 1)].cj; }
        return pairs_dma_cam(blocks, c, lane);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ckv
    const uint4* rec4 = reinterpret_cast<const uint4*>(recs);
    int q = 0;                                          // task-relative chunk index
    PairChunk ck = chunk_desc(0);
    uint4 rr = rec4[(size_t)chunk0 * 64 + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the first chunk's indices travel through the shuffles at once)
    Coop coop;
    issue(rr, coop);
    BlockDesc bd, bd_next;
    load_blocks(ck, bd, 0);
    bd_next = bd;
    bool dma = QL || 1 + __popc(ck.mask & ~1u) <= kPairDmaBlocks;
    auto cam_piece = [&](uint32_t cam) {   // lanes 8 c .. 8 c + 7 fetch the 128-byte prepared camera c of the chunk
        return *reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)cam + 2 * (lane & 7));
    };
    double2 cam_stage = cam_piece(chunk_cam(ck, 0));
    PairChunk ck_next = chunk_desc(min(1, nchunks - 1));
    uint4 rr_next = rec4[(size_t)(chunk0 + min(1, nchunks - 1)) * 64 + lane];
    uint32_t cam_next = chunk_cam(ck_next, min(1, nchunks - 1));

    // queued layout: the row's camera is the same for every pair of the task -- it stays in registers (the same address in
    // every lane: one request per load), and only the seven partners are read from the staged copy chunk by chunk
    double cvi_task[QL ? 16 : 1];
    if constexpr (QL) {
        const double2* pc = reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)qdesc[ND * (size_t)chunk0 + NQ].cj);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double2 a = pc[k]; cvi_task[2 * k] = a.x; cvi_task[2 * k + 1] = a.y; }
    }
    for (; q < nchunks; ++q) {
        const bool valid = rr.x != kPairPad;
        const uint32_t blk = valid ? rr.w : 0u;   // (queued layout: the slot's queue)
        // ---- cameras of this lane's pair: LDS (DMA issued a chunk ago) or, in a chunk of many tiny blocks, memory ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // gathers, block descriptors, and the LDS-DMA (it writes LDS behind the VM counter)
        *reinterpret_cast<double2*>(CAMS + (lane >> 3) * kCamLds + 2 * (lane & 7)) = cam_stage;   // (the previous chunk's camera reads are long done)
        __builtin_amdgcn_wave_barrier();
        Gather dat;
        unstage_dpp(coop, dat);
        double cvi[16], cvj[16];
        {
            if (QL) {
                const double2* cj = reinterpret_cast<const double2*>(CAMS + (1 + blk) * kCamLds);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 b = cj[k]; cvj[2 * k] = b.x; cvj[2 * k + 1] = b.y; }
#pragma unroll
                for (int k = 0; k < 16; ++k) cvi[k] = cvi_task[QL ? k : 0];
            } else if (dma) {
                const double2* ci = reinterpret_cast<const double2*>(CAMS + (2 * blk) * kCamLds);
                const double2* cj = reinterpret_cast<const double2*>(CAMS + (2 * blk + 1) * kCamLds);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 a = ci[k], b = cj[k]; cvi[2 * k] = a.x; cvi[2 * k + 1] = a.y; cvj[2 * k] = b.x; cvj[2 * k + 1] = b.y; }
            } else {   // a chunk of many tiny blocks: the cameras come straight from memory
                const PairBlock* pb = blocks + ck.first_block + blk;
                const double2* ci = reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)pb->ci);
                const double2* cj = reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)pb->cj);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 a = ci[k], b = cj[k]; cvi[2 * k] = a.x; cvi[2 * k + 1] = a.y; cvj[2 * k] = b.x; cvj[2 * k + 1] = b.y; }
            }
        }
        double Hi[9], pw[3];
        // the landmark record's first line: Hll^-1 (bitwise symmetric) as (00, 01, 02, 11, 12, 22) | p_w.x, p_w.y; p_w.z rides in
        // slot 2 of the projection records (both carry the same landmark's)
        Hi[0] = dat.lm[0].x; Hi[1] = dat.lm[0].y; Hi[2] = dat.lm[1].x; Hi[4] = dat.lm[1].y; Hi[5] = dat.lm[2].x; Hi[8] = dat.lm[2].y;
        Hi[3] = Hi[1]; Hi[6] = Hi[2]; Hi[7] = Hi[5];
        pw[0] = dat.lm[3].x; pw[1] = dat.lm[3].y; pw[2] = dat.ri1.x;
        // A padding slot (and the filler slot of an odd block) contributes exact zeros: its lanes carry element 0's records
        // with the Huber weight forced to 0, which zeroes a, (xn w, yn w) and with them V, M and U.
        const double2 rj1 = make_double2(dat.rj1.x, valid ? dat.rj1.y : 0.0), ri1 = make_double2(dat.ri1.x, valid ? dat.ri1.y : 0.0);
        // ---- column side j: V = Jc_j, Q = Hll^-1 Jl_j^T ---------------------------------------------------------------
        double Q[3][2];
        {
            RecJac J;
            jac_from_rec(cvj, dat.rj0, rj1, pw, J);
            double Jc[2][DC];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const double a0 = J.a[r][0], a1 = J.a[r][1], a2 = J.a[r][2];
                Jc[r][0] = a0 * mp; Jc[r][1] = a1 * mp; Jc[r][2] = a2 * mp;
                Jc[r][3] = fma(a2, pw[1], -(a1 * pw[2])) * mp;
                Jc[r][4] = fma(a0, pw[2], -(a2 * pw[0])) * mp;
                Jc[r][5] = fma(a1, pw[0], -(a0 * pw[1])) * mp;
                if (DC == 9) {
                    const double sw = (r == 0 ? J.xw : J.yw) * mi;
                    Jc[r][6] = sw * J.t[0]; Jc[r][7] = sw * J.t[1]; Jc[r][8] = sw * J.t[2];
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int m = 0; m < 2; ++m) Q[a][m] = (Hi[3 * a] * J.a[m][0] + Hi[3 * a + 1] * J.a[m][1] + Hi[3 * a + 2] * J.a[m][2]) * ml;
            double2* pv = reinterpret_cast<double2*>(V + lane * UV);   // element order [bj][m][3]: V[m][3 bj + c]
#pragma unroll
            for (int k = 0; k < DC; ++k) {
                const int e0 = 2 * k, e1 = 2 * k + 1;
                const int s0 = e0 / 6, m0 = (e0 % 6) / 3, c0 = e0 % 3, s1 = e1 / 6, m1 = (e1 % 6) / 3, c1 = e1 % 3;
                pv[k] = make_double2(Jc[m0][3 * s0 + c0], Jc[m1][3 * s1 + c1]);
            }
        }
        // ---- row side i: M = -Jl_i Q, U = Jc_i^T M = [G ; p_w x G ; t (s M)], G = a_i^T M -----------------------------------
        {
            RecJac J;
            jac_from_rec(cvi, dat.ri0, ri1, pw, J);
            double M[2][2];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m) M[n][m] = -((J.a[n][0] * Q[0][m] + J.a[n][1] * Q[1][m] + J.a[n][2] * Q[2][m]) * ml);
            double u[UV];   // U[3 bi + c][m] at [bi][m][3]
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const double g0 = (J.a[0][0] * M[0][m] + J.a[1][0] * M[1][m]) * mp;
                const double g1 = (J.a[0][1] * M[0][m] + J.a[1][1] * M[1][m]) * mp;
                const double g2 = (J.a[0][2] * M[0][m] + J.a[1][2] * M[1][m]) * mp;
                u[0 * 6 + m * 3 + 0] = g0; u[0 * 6 + m * 3 + 1] = g1; u[0 * 6 + m * 3 + 2] = g2;
                u[1 * 6 + m * 3 + 0] = fma(g2, pw[1], -(g1 * pw[2]));
                u[1 * 6 + m * 3 + 1] = fma(g0, pw[2], -(g2 * pw[0]));
                u[1 * 6 + m * 3 + 2] = fma(g1, pw[0], -(g0 * pw[1]));
                if (DC == 9) {
                    const double sm = (J.xw * M[0][m] + J.yw * M[1][m]) * mi;
                    u[2 * 6 + m * 3 + 0] = sm * J.t[0]; u[2 * 6 + m * 3 + 1] = sm * J.t[1]; u[2 * 6 + m * 3 + 2] = sm * J.t[2];
                }
            }
            double2* pu = reinterpret_cast<double2*>(U + lane * UV);
#pragma unroll
            for (int k = 0; k < UV / 2; ++k) pu[k] = make_double2(u[2 * k], u[2 * k + 1]);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- the next chunk's gathers, block descriptors and cameras go out now and land during the product phase ------------
        const PairChunk ck_cur = ck;
        const BlockDesc bd_cur = bd_next;     // (the descriptors of THIS chunk: loaded a chunk ago, or before the loop)
        {
            // Branch-free and clamped (the last chunks of a task fetch their own data again): a conditional load into a
            // loop-carried register makes the compiler resolve the phi with a copy right behind the load -- and wait for
            // every outstanding load (s_waitcnt vmcnt(0)) to do it, i.e. for the gathers issued three lines earlier.
            ck = ck_next; rr = rr_next;
            issue(rr, coop);
            load_blocks(ck, bd_next, min(q + 1, nchunks - 1));
            dma = QL || 1 + __popc(ck.mask & ~1u) <= kPairDmaBlocks;
            cam_stage = cam_piece(cam_next);
            const int q2 = min(q + 2, nchunks - 1);
            ck_next = chunk_desc(q2);
            rr_next = rec4[(size_t)(chunk0 + q2) * 64 + lane];
            cam_next = chunk_cam(ck_next, q2);
        }
        // The block finished by the PREVIOUS chunk goes to memory here, behind this chunk's gathers: the vector-memory counter
        // counts stores too and in issue order, so stores issued before loads that the code waits for (the compiler puts its own
        // s_waitcnt vmcnt(small) in front of the first use of a prefetched descriptor) stall the wave until they are ACKNOWLEDGED --
        // 0.5 ms of the kernel when they sat at the top of the chunk.  Here the next wait for memory is a whole product phase away.
        if constexpr (QL) store_pending();
        // ---- block products over the 64 slots.  Uniform loop, two pairs per trip with all twelve operand reads issued up
        // front; a lane whose pair lies beyond the segment (and the idle 64th lane) reads the zero row instead.
        if constexpr (QL) {
            // nine steps, pair g + 7 t of the chunk in step t; ping-pong operand registers as below
            {
                auto ld = [&](int t, double2& u0, double2& u1, double2& u2, double2& v0, double2& v1, double2& v2) {
                    const double2* qu = reinterpret_cast<const double2*>(U + (g + NQ * t) * UV + bi * 6);
                    const double2* qv = reinterpret_cast<const double2*>(V + (g + NQ * t) * UV + bj * 6);
                    u0 = qu[0]; u1 = qu[1]; u2 = qu[2]; v0 = qv[0]; v1 = qv[1]; v2 = qv[2];
                };
                auto mac = [&](const double2 u0, const double2 u1, const double2 u2, const double2 v0, const double2 v1, const double2 v2) {
                    const double um0[3] = {u0.x, u0.y, u1.x}, um1[3] = {u1.y, u2.x, u2.y};
                    const double vm0[3] = {v0.x, v0.y, v1.x}, vm1[3] = {v1.y, v2.x, v2.y};
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[3 * r + c] = fma(um1[r], vm1[c], fma(um0[r], vm0[c], acc[3 * r + c]));
                };
                double2 a0, a1, a2, a3, a4, a5, b0, b1, b2, b3, b4, b5;
                ld(0, a0, a1, a2, a3, a4, a5);
#pragma unroll
                for (int t = 0; t < 8; t += 2) {
                    ld(t + 1, b0, b1, b2, b3, b4, b5);
                    mac(a0, a1, a2, a3, a4, a5);
                    ld(t + 2, a0, a1, a2, a3, a4, a5);
                    mac(b0, b1, b2, b3, b4, b5);
                }
                mac(a0, a1, a2, a3, a4, a5);
            }
            // a queue whose block (piece) ends with this chunk: its nine lanes hold the finished 3 x 3 sub-blocks.  They are
            // parked and stored at the top of the NEXT chunk, behind its wait for the gathers: stored here they would be the
            // youngest entries of the memory counter that wait drains, i.e. a store round trip on every chunk's critical path
            // (a block cut between queue g's tail and queue g + 1's head: the head's sum waits in `carry` for the last chunk)
            if (q == nchunks - 1) {   // wave-uniform: the tails that join a carried head (kPairQJoin only occurs here)
                const bool join = ((ck_cur.mask >> g) & 1u) && (bd_cur.flags & kPairQJoin);
#pragma unroll
                for (int k = 0; k < 9; ++k) { const double o = __shfl_down(carry[k], GL, 64); acc[k] += join ? o : 0.0; }
            }
            if (ck_cur.mask & ((1u << NQ) - 1u)) {   // wave-uniform: some queue's block (piece) ends with this chunk
                // The nine lanes of a group hold the block as 3 x 3 sub-blocks; stored like that, one store instruction touches
                // six 64-byte lines per block (three rows, each 72-byte row astride two lines) and a block costs 54 line
                // writes -- 0.77 ms of the kernel went there.  Turned through the (now free) U area into one COLUMN per lane,
                // store instruction r writes row r of every finished block as nine adjacent lanes: 18 line writes per block.
                const bool mine = (ck_cur.mask >> g) & 1u;
                const bool keep = mine && (bd_cur.flags & kPairQCarry) != 0;
                double* T = U + 81 * g;
                if (mine && !keep && lane < 63) {   // (lane 63 is a shadow: it must not write)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int r = 0; r < 3; ++r) T[(3 * bj + c) * 9 + 3 * bi + r] = acc[3 * r + c];
                }
                __builtin_amdgcn_wave_barrier();
                if (mine) {
                    if (!keep) {
#pragma unroll
                        for (int r = 0; r < 9; ++r) pend[r] = T[9 * sub + r];
                        pend_dst = bd_cur.dst; pend_fl = bd_cur.flags; pend_on = lane < 63;
                    }
#pragma unroll
                    for (int k = 0; k < 9; ++k) { carry[k] = keep ? acc[k] : carry[k]; acc[k] = 0.0; }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
        uint32_t mask = ck_cur.mask;
        int seg0 = 0, lb = 0;                 // lb: index of the running block inside this chunk's descriptors
        auto start_block = [&]() {
            cur_dst = ((int64_t)__builtin_amdgcn_readlane(bd_cur.dst.y, lb) << 32) | (uint32_t)__builtin_amdgcn_readlane(bd_cur.dst.x, lb);
            cur_flags = (uint32_t)__builtin_amdgcn_readlane((int)bd_cur.flags, lb);
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
        };
        auto flush = [&]() {
            pairs_flush2<DC>(tiles, cur_dst, cur_flags, acc, lane);
        };
        if (mask & 1u) {   // the chunk opens a new block
            if (cur >= 0) flush();
            cur = 0;
            start_block();
        } else if (cur < 0) {   // (cannot happen: a task starts with a block; keeps the descriptor valid anyway)
            cur = 0;
            start_block();
        }
        mask &= ~1u;
        for (;;) {
            const int seg1 = mask ? 2 * (__ffs(mask) - 1) : 64;      // wave-uniform
            {
                // One pair per step and lane, ping-pong operand registers: the six reads of the NEXT pair are in flight while
                // the 18 FMA of this one run.  Uniform control flow; a lane whose pair lies beyond the segment (and the idle
                // lane) reads the zero row.
                int p = seg0 + g;
                auto ld = [&](int pp, double2& u0, double2& u1, double2& u2, double2& v0, double2& v1, double2& v2) {
                    const bool ok = worker && pp < seg1;
                    const double2* qu = reinterpret_cast<const double2*>(ok ? U + pp * UV + bi * 6 : Z);
                    const double2* qv = reinterpret_cast<const double2*>(ok ? V + pp * UV + bj * 6 : Z);
                    u0 = qu[0]; u1 = qu[1]; u2 = qu[2]; v0 = qv[0]; v1 = qv[1]; v2 = qv[2];
                };
                auto mac = [&](const double2 u0, const double2 u1, const double2 u2, const double2 v0, const double2 v1, const double2 v2) {
                    const double um0[3] = {u0.x, u0.y, u1.x}, um1[3] = {u1.y, u2.x, u2.y};
                    const double vm0[3] = {v0.x, v0.y, v1.x}, vm1[3] = {v1.y, v2.x, v2.y};
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[3 * r + c] = fma(um1[r], vm1[c], fma(um0[r], vm0[c], acc[3 * r + c]));
                };
                double2 a0, a1, a2, a3, a4, a5, b0, b1, b2, b3, b4, b5;
                int it = seg0;
                ld(p, a0, a1, a2, a3, a4, a5);
                for (;;) {
                    if (it + NG < seg1) ld(p + NG, b0, b1, b2, b3, b4, b5);
                    mac(a0, a1, a2, a3, a4, a5);
                    it += NG; p += NG;
                    if (it >= seg1) break;
                    if (it + NG < seg1) ld(p + NG, a0, a1, a2, a3, a4, a5);
                    mac(b0, b1, b2, b3, b4, b5);
                    it += NG; p += NG;
                    if (it >= seg1) break;
                }
            }
            if (!mask) break;
            flush();
            ++lb;
            start_block();
            seg0 = seg1;
            mask &= mask - 1;
        }
        }
        __builtin_amdgcn_wave_barrier();
        bd = bd_next;
    }
    if constexpr (QL) store_pending();
    if (!QL && cur >= 0) {
        pairs_flush2<DC>(tiles, cur_dst, cur_flags, acc, lane);
    }
}

// The pair kernel is the record form (k_schur_pairs_r): it needs the projection records k_landmark_reduce writes.
// qdesc != NULL: the queued layout (nine-column cameras).  Rounds 2-3 kept three more pair kernels for A/B (fused one pair per
// lane, fused two lanes per pair, record form two lanes per pair: 3.96 / 3.75-3.91 / 3.94 ms against 3.43-3.8 for this one on
// final-13682, profiles/r03_pairs_ablation.txt); they were deleted in round 4, the LDS row form of round 2 in round 6.
void launch_schur_pairs(int dc, const BAView& v, double* tiles, const PairTask* tasks, int n_tasks, const PairChunk* chunks,
                        const PairBlock* blocks, const PairRec* recs, const double* lmrec, hipStream_t s, const double* orec,
                        const PairQDesc* qdesc) {
    if (n_tasks == 0) return;
    const unsigned grid = (unsigned)((n_tasks + 3) / 4);
    const bool masked = v.mask_code != (dc == 9 ? 7 : 6);
#define PAIRS_R(DCV, MK, Q) hipLaunchKernelGGL((k_schur_pairs_r<DCV, MK, Q>), dim3(grid), dim3(256), 0, s, v, tiles, tasks, n_tasks, chunks, blocks, recs, lmrec, orec, qdesc)
    if (dc == 9 && qdesc) { if (masked) PAIRS_R(9, true, true); else PAIRS_R(9, false, true); }
    else if (dc == 9) { if (masked) PAIRS_R(9, true, false); else PAIRS_R(9, false, false); }
    else { if (masked) PAIRS_R(6, true, false); else PAIRS_R(6, false, false); }
#undef PAIRS_R
}

// (set-up: the first launch of a kernel of this translation unit loads its code object -- tens of milliseconds for the big
// ones; Solver::set_structure pays that on a background thread while the host builds its lists: warm_device_code)
__global__ void k_warm_schur_pairs() {}
void warm_schur_pairs(hipStream_t s) { hipLaunchKernelGGL(k_warm_schur_pairs, dim3(1), dim3(64), 0, s); }

}  // namespace apex
