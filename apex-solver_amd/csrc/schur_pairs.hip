// schur_pairs.hip -- see schur_pairs.h
#include "schur_pairs.h"

#include <algorithm>
#include <type_traits>

#include "ba_device.hpp"
#include "host_parallel.h"

namespace apex {


constexpr int kPairTaskSlots = 1536;     // a wave's task is closed once it holds this many slots (24 chunks: the pipeline's
                                         // prologue -- three dependent loads -- is paid once per task)
constexpr int kPairMaxBlockSlots = 4096; // a block with more slots is split over several waves (atomic flush).  With 2 kTask <= 4096
                                         // a task never has more than 64 chunks: the record kernel keeps a task's chunk descriptors
                                         // one per lane (k_schur_pairs_r)
constexpr int kPairCamPitch = 18;        // doubles per staged camera: 144 B keeps 16-byte alignment and spreads the banks

// ------------------------------------------------------------------------------------------------------------------
// host: the sorted pair list
// ------------------------------------------------------------------------------------------------------------------
namespace {

}  // namespace

void build_pair_lists(int dc, int nt, const int* slot, int64_t n_cam, const int* cam_ext, const uint32_t* o_cam,
                      const uint32_t* o_pt, const int* pt_ptr, const int* cam_ptr, const int* cam_obs, PairLists* out,
                      int task_slots) {
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
    struct Run { uint32_t cj; int len; int piece0; };
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
    // ---- serial pass over the BLOCKS (not the pairs): slot offsets, block table, chunk descriptors, tasks ------------
    out->blocks.clear(); out->chunks.clear(); out->tasks.clear();
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
// Branch-free linearisation for this kernel: same formulas as linearize_obs (ba_device.hpp), with the division and the
// Huber weight on the reciprocal / reciprocal-square-root units refined by Newton steps (full double precision, a
// quarter of the instructions of the IEEE sequences) and the cheirality test as a select -- the 64 lanes of a wave
// linearise 64 different observations and must not serialise on each other's branches.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {   // x > 0
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    return y;
}

// J = [Jl | -Jl [pw]x | e] scaled by the Huber weight (0 for a point behind the camera): Jl 2x3, the rest derived
template <int DC>
__device__ __forceinline__ void linearize_pairside(const double* __restrict__ cv, const double pw[3], double u_obs, double v_obs,
                                                   double huber_delta, double Jc[2][DC], double Jl[2][3]) {
    const double pcx = cv[0] * pw[0] + cv[1] * pw[1] + cv[2] * pw[2] + cv[9];
    const double pcy = cv[3] * pw[0] + cv[4] * pw[1] + cv[5] * pw[2] + cv[10];
    const double pcz = cv[6] * pw[0] + cv[7] * pw[1] + cv[8] * pw[2] + cv[11];
    const bool ok = pcz < -kMinDepth;
    const double f = cv[12], k1 = cv[13], k2 = cv[14];
    const double inz = -fast_rcp(ok ? pcz : -1.0);
    const double xn = pcx * inz, yn = pcy * inz;
    const double r2 = xn * xn + yn * yn, r4 = r2 * r2;
    const double dist = 1.0 + k1 * r2 + k2 * r4;
    const double r0 = f * (xn * dist) - u_obs, r1 = f * (yn * dist) - v_obs;
    const double sn = r0 * r0 + r1 * r1;
    // Huber: sqrt(rho') = sqrt(delta / sqrt(s)) for s > delta^2, else 1 (corrector.rs:156-162)
    double w = 1.0;
    {
        const bool out = huber_delta > 0.0 && sn > huber_delta * huber_delta;
        const double ss = out ? sn : 1.0;
        const double t = huber_delta * fast_rsqrt(ss);      // delta / sqrt(s)
        const double wq = t * fast_rsqrt(t);                // sqrt(t)
        w = out ? wq : 1.0;
    }
    w = ok ? w : 0.0;
    const int mcode = (int)cv[15];   // OptimizeParams column masks (ba_device.hpp): 4 POSE + 2 LANDMARK + INTRINSIC
    const double mp = (mcode & 4) ? 1.0 : 0.0, ml = (mcode & 2) ? 1.0 : 0.0, mi = (mcode & 1) ? 1.0 : 0.0;
    const double dd = k1 + 2.0 * k2 * r2;
    const double dxn_dz = xn * inz, dyn_dz = yn * inz;
    const double dxd_dxn = dist + xn * dd * 2.0 * xn, dxd_dyn = xn * dd * 2.0 * yn;
    const double dyd_dxn = yn * dd * 2.0 * xn, dyd_dyn = dist + yn * dd * 2.0 * yn;
    const double fw = f * w;
    double Jp[2][3];
    Jp[0][0] = fw * (dxd_dxn * inz); Jp[0][1] = fw * (dxd_dyn * inz); Jp[0][2] = fw * (dxd_dxn * dxn_dz + dxd_dyn * dyn_dz);
    Jp[1][0] = fw * (dyd_dxn * inz); Jp[1][1] = fw * (dyd_dyn * inz); Jp[1][2] = fw * (dyd_dxn * dxn_dz + dyd_dyn * dyn_dz);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const double a0 = Jp[rr][0] * cv[0] + Jp[rr][1] * cv[3] + Jp[rr][2] * cv[6];
        const double a1 = Jp[rr][0] * cv[1] + Jp[rr][1] * cv[4] + Jp[rr][2] * cv[7];
        const double a2 = Jp[rr][0] * cv[2] + Jp[rr][1] * cv[5] + Jp[rr][2] * cv[8];
        Jl[rr][0] = a0 * ml; Jl[rr][1] = a1 * ml; Jl[rr][2] = a2 * ml;
        Jc[rr][0] = a0 * mp; Jc[rr][1] = a1 * mp; Jc[rr][2] = a2 * mp;
        Jc[rr][3] = (a2 * pw[1] - a1 * pw[2]) * mp;
        Jc[rr][4] = (a0 * pw[2] - a2 * pw[0]) * mp;
        Jc[rr][5] = (a1 * pw[0] - a0 * pw[1]) * mp;
    }
    if (DC == 9) {
        const double xw = xn * w * mi, yw = yn * w * mi, fr2 = f * r2, fr4 = f * r4;
        Jc[0][DC - 3] = xw * dist; Jc[0][DC - 2] = xw * fr2; Jc[0][DC - 1] = xw * fr4;
        Jc[1][DC - 3] = yw * dist; Jc[1][DC - 2] = yw * fr2; Jc[1][DC - 1] = yw * fr4;
    }
}

// The data one lane needs for its pair, fetched one chunk AHEAD (while the previous chunk's block products run): both
// measurements, Hll^-1 and the point, and -- in the first 2 nblk lanes -- one camera of the chunk's blocks.
struct PairData {
    double2 uvi, uvj;
    double2 lm[6];
};
constexpr int kPairDmaBlocks = 4;   // a chunk with at most this many blocks stages its <= 8 cameras by ONE LDS-DMA

// Camera staging.  Fast path (a chunk with <= kPairDmaBlocks blocks, i.e. nearly every chunk of a capture with real
// overlap): the 8 lanes t = 8 c .. 8 c + 7 copy the 128-byte prepared camera c of the chunk (camera c & 1 of block c >> 1)
// straight into the wave's camera area with one global_load_lds_dwordx4 -- no registers, no ds_write, issued a chunk ahead.
// Slow path (many tiny blocks): lane t < 2 nblk loads camera t through registers into the area U will overwrite.
__device__ __forceinline__ uint32_t pairs_dma_cam(const PairBlock* __restrict__ blocks, const PairChunk ck, int lane) {
    const int nblk = 1 + __popc(ck.mask & ~1u);
    const int c = lane >> 3;
    const PairBlock* pb = blocks + ck.first_block + min(c >> 1, nblk - 1);
    return (c & 1) ? pb->cj : pb->ci;
}
__device__ __forceinline__ uint32_t pairs_slow_cam(const PairBlock* __restrict__ blocks, const PairChunk ck, int lane) {
    const int nblk = 1 + __popc(ck.mask & ~1u);
    const PairBlock* pb = blocks + ck.first_block + min(lane >> 1, nblk - 1);
    return (lane & 1) ? pb->cj : pb->ci;
}
__device__ __forceinline__ void pairs_dma_issue(const BAView& v, uint32_t cam, int lane, double* lds_cams) {
    const char* src = reinterpret_cast<const char*>(v.camp + kCamStride * (size_t)cam) + 16 * (lane & 7);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_cams, 16, 0, 0);
}

__device__ __forceinline__ void pairs_issue_loads(const BAView& v, const double* __restrict__ lmrec, const uint4 rr, PairData& d) {
    const bool valid = rr.x != kPairPad;
    const uint32_t i = valid ? rr.x : 0u, j = valid ? rr.y : 0u, l = valid ? rr.z : 0u;   // padding lanes read element 0
    d.uvi = v.o_uv[i];
    d.uvj = v.o_uv[j];
    const double2* q = reinterpret_cast<const double2*>(lmrec + kLmStride * (size_t)l);
#pragma unroll
    for (int k = 0; k < 6; ++k) d.lm[k] = q[k];
}

// Sum of the lane groups' partial blocks and the ONE store of S(ci, cj).  Lane L = 9 g + sub (DC = 9; 4 g + sub for DC = 6)
// holds the 3 x 3 sub-block (bi, bj) = (sub / NB3, sub % NB3) of group g's partial sum.
template <int DC>
__device__ __forceinline__ void pairs_flush(double* __restrict__ tiles, const int64_t pb_dst, const uint32_t pb_flags,
                                            double acc[9], int lane) {
    constexpr int NB3 = DC / 3, GL = NB3 * NB3;
    constexpr int NG = (DC == 9) ? 7 : 16, P2 = (DC == 9) ? 8 : 16;
    const int g = lane / GL;
    // groups g >= 1 fold into group 0 in log2 steps (a group beyond the last one contributes nothing)
#pragma unroll
    for (int st = P2 / 2; st >= 1; st >>= 1) {
        const bool take = g < st && g + st < NG;
        const int src = take ? lane + st * GL : lane;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double other = __shfl(acc[k], src, 64);
            if (take) acc[k] += other;
        }
    }
    if (lane < GL) {
        struct { int64_t dst; uint32_t flags; } pb = {pb_dst, pb_flags};
        double* dst = tiles + pb.dst;
        const int bi = lane / NB3, bj = lane % NB3;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int row = 3 * bi + r, col = 3 * bj + c;
                const double val = acc[3 * r + c];
                if (pb.flags == 0) {
                    dst[row * kNB + col] = val;
                } else if (pb.flags & kPairBlockDiag) {   // B + B^T, kept in the lower triangle of the diagonal block
                    if (row >= col) unsafeAtomicAdd(&dst[row * kNB + col], val);
                    if (col >= row) unsafeAtomicAdd(&dst[col * kNB + row], val);
                } else {
                    unsafeAtomicAdd(&dst[row * kNB + col], val);
                }
            }
    }
}

// ABL: timing-only ablation switches (results are wrong when != 0): 1 = the per-pair gathers (measurements, landmark
// record) replaced by registers, 2 = no block products, 4 = no linearisation (U, V from the loaded data directly)
template <int DC, int ABL>
__global__ __launch_bounds__(256) void k_schur_pairs(BAView v, double* __restrict__ tiles, const PairTask* __restrict__ tasks,
                                                       int n_tasks, const PairChunk* __restrict__ chunks,
                                                       const PairBlock* __restrict__ blocks, const PairRec* __restrict__ recs,
                                                       const double* __restrict__ lmrec) {
    constexpr int UV = 2 * DC;                    // doubles of U (and of V) per pair
    constexpr int NB3 = DC / 3;                   // 3 x 3 sub-blocks per block edge
    constexpr int GL = NB3 * NB3;                 // lanes of one group = sub-blocks of a block (9 / 4)
    constexpr int NG = (DC == 9) ? 7 : 16;        // lane groups that split a segment's pairs (63 / 64 lanes busy)
    constexpr int REG_A = 64 * kPairCamPitch;     // U[64][UV] overlays the staged cameras (64 x 18 doubles >= 64 x UV)
    constexpr int WAVE_LDS = REG_A + 64 * UV + 8 * kCamStride;   // | V[64][UV] | 8 cameras staged by LDS-DMA
    static_assert(64 * UV <= REG_A, "U must fit the camera staging area");
    __shared__ double lds_all[4 * WAVE_LDS];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Tasks are ordered by row camera, and the
    // ~17 pairs that use one landmark record sit in rows a capture window apart: giving every XCD a CONTIGUOUS eighth of
    // the task list keeps a landmark's pairs behind one L2 (measured: L2 hit rate 36 % -> see DESIGN.md) instead of
    // spreading them over all eight.  Speed only: any mapping is correct.
    int wg = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    const int t = wg * 4 + w;
    if (t >= n_tasks) return;                     // no workgroup barrier anywhere: the four waves are independent
    double* U = lds_all + w * WAVE_LDS;
    double* V = U + REG_A;
    double* CAMS = V + 64 * UV;
    const PairTask task = tasks[t];
    // product phase: lane = (group g, sub-block (bi, bj)); U and V are stored per pair as [sub-row][m][3] so that a lane's
    // six U values (and six V values) are 48 contiguous, 16-byte aligned bytes
    const int g = lane / GL, sub = lane - g * GL, bi = sub / NB3, bj = sub - bi * NB3;
    const bool worker = g < NG;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    int cur = -1;
    int64_t cur_dst = 0;       // descriptor of the block being accumulated, fetched when the block STARTS (a scalar load
    uint32_t cur_flags = 0;    // whose latency the block's own pairs hide), not when it is flushed

    // Software pipeline over the task's chunks: the gathers of chunk n+1 are issued before the product phase of chunk n
    // and land while it runs; the 16-byte records run two chunks ahead.
    const int ch_end = task.chunk0 + task.nchunks;
    int ch = task.chunk0;
    PairChunk ck = chunks[ch];
    uint4 rr = reinterpret_cast<const uint4*>(recs)[(size_t)ch * 64 + lane];
    PairData dat;
    pairs_issue_loads(v, lmrec, rr, dat);
    bool dma = 1 + __popc(ck.mask & ~1u) <= kPairDmaBlocks;
    if (dma) pairs_dma_issue(v, pairs_dma_cam(blocks, ck, lane), lane, CAMS);
    // two chunks ahead: the 16-byte record, the chunk descriptor and the camera index this lane will stage (so that the
    // gathers one chunk ahead depend on nothing that is still in flight)
    PairChunk ck_next = ck;
    uint4 rr_next = rr;
    uint32_t cam_next = 0;
    if (ch + 1 < ch_end) {
        ck_next = chunks[ch + 1];
        rr_next = reinterpret_cast<const uint4*>(recs)[(size_t)(ch + 1) * 64 + lane];
        cam_next = pairs_dma_cam(blocks, ck_next, lane);
    }

    for (; ch < ch_end; ++ch) {
        // ---- A: the chunk's cameras are in LDS: by the DMA issued a chunk ago, or (many tiny blocks) staged now ----------
        const double* cam_base = CAMS;
        int cam_pitch = kCamStride;
        if (!dma) {
            const uint32_t cam = pairs_slow_cam(blocks, ck, lane);
            const double2* src = reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)cam);
            double2* dstc = reinterpret_cast<double2*>(U + lane * kPairCamPitch);
#pragma unroll
            for (int k = 0; k < 8; ++k) dstc[k] = src[k];
            cam_base = U; cam_pitch = kPairCamPitch;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA writes LDS behind the VM counter
        const bool valid = rr.x != kPairPad;
        __builtin_amdgcn_wave_barrier();
        // ---- B: one pair per lane: both observations linearised, U = Jc_i^T M, V = Jc_j ---------------------------------
        double u[UV];
        {
            double Hi[9], pw[3];
            Hi[0] = dat.lm[0].x; Hi[1] = dat.lm[0].y; Hi[2] = dat.lm[1].x; Hi[3] = dat.lm[1].y; Hi[4] = dat.lm[2].x; Hi[5] = dat.lm[2].y;
            Hi[6] = dat.lm[3].x; Hi[7] = dat.lm[3].y; Hi[8] = dat.lm[4].x; pw[0] = dat.lm[4].y; pw[1] = dat.lm[5].x; pw[2] = dat.lm[5].y;
            const uint32_t blk = valid ? rr.w : 0u;
            double N[2][3];
            double Jci[2][DC];
            {
                const double2* c2 = reinterpret_cast<const double2*>(cam_base + (2 * blk) * cam_pitch);
                double cv[16];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 tq = c2[k]; cv[2 * k] = tq.x; cv[2 * k + 1] = tq.y; }
                double Jl[2][3];
                if (ABL & 4) {
#pragma unroll
                    for (int a = 0; a < DC; ++a) { Jci[0][a] = cv[a] + dat.uvi.x; Jci[1][a] = cv[a + 6] * dat.uvi.y; }
#pragma unroll
                    for (int a = 0; a < 3; ++a) { Jl[0][a] = pw[a]; Jl[1][a] = cv[a]; }
                } else
                linearize_pairside<DC>(cv, pw, dat.uvi.x, dat.uvi.y, v.huber_delta, Jci, Jl);
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int b = 0; b < 3; ++b) N[n][b] = Jl[n][0] * Hi[b] + Jl[n][1] * Hi[3 + b] + Jl[n][2] * Hi[6 + b];
            }
            double M[2][2];
            {
                const double2* c2 = reinterpret_cast<const double2*>(cam_base + (2 * blk + 1) * cam_pitch);
                double cv[16];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 tq = c2[k]; cv[2 * k] = tq.x; cv[2 * k + 1] = tq.y; }
                double Jcj[2][DC], Jl[2][3];
                if (ABL & 4) {
#pragma unroll
                    for (int a = 0; a < DC; ++a) { Jcj[0][a] = cv[a] + dat.uvj.x; Jcj[1][a] = cv[a + 6] * dat.uvj.y; }
#pragma unroll
                    for (int a = 0; a < 3; ++a) { Jl[0][a] = pw[a]; Jl[1][a] = cv[a]; }
                } else
                linearize_pairside<DC>(cv, pw, dat.uvj.x, dat.uvj.y, v.huber_delta, Jcj, Jl);
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m) M[n][m] = -(N[n][0] * Jl[m][0] + N[n][1] * Jl[m][1] + N[n][2] * Jl[m][2]);
                // V goes to its own LDS region straight away (it never overlaps the staged cameras); element order
                // [sub-column bj][m][3]: V[m][3 bj + c]
                double2* pv = reinterpret_cast<double2*>(V + lane * UV);
#pragma unroll
                for (int k = 0; k < DC; ++k) {
                    const int e0 = 2 * k, e1 = 2 * k + 1;
                    const int s0 = e0 / 6, m0 = (e0 % 6) / 3, c0 = e0 % 3, s1 = e1 / 6, m1 = (e1 % 6) / 3, c1 = e1 % 3;
                    // (a padding slot contributes exact zeros on both sides, by selection: see k_schur_pairs_h)
                    if (!(ABL & 16) || k == 0) pv[k] = make_double2(valid ? Jcj[m0][3 * s0 + c0] : 0.0, valid ? Jcj[m1][3 * s1 + c1] : 0.0);
                    else asm volatile("" ::"v"(Jcj[m0][3 * s0 + c0]), "v"(Jcj[m1][3 * s1 + c1]));
                }
            }
#pragma unroll
            for (int e = 0; e < UV; ++e) {   // U[m][3 bi + c] in the order [bi][m][3]
                const int s0 = e / 6, m = (e % 6) / 3, c = e % 3;
                const double uv = Jci[0][3 * s0 + c] * M[0][m] + Jci[1][3 * s0 + c] * M[1][m];
                u[e] = valid ? uv : 0.0;
            }
        }
        // every lane has read its cameras (program order, one wave): U may now overwrite the staging area
        __builtin_amdgcn_wave_barrier();
        {
            double2* pu = reinterpret_cast<double2*>(U + lane * UV);
#pragma unroll
            for (int k = 0; k < UV / 2; ++k) {
                if (!(ABL & 16) || k == 0) pu[k] = make_double2(u[2 * k], u[2 * k + 1]);
                else asm volatile("" ::"v"(u[2 * k]), "v"(u[2 * k + 1]));
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- D: the next chunk's gathers go out now and land during the product phase ---------------------------------
        const PairChunk ck_cur = ck;
        if (ch + 1 < ch_end) {
            ck = ck_next; rr = rr_next;
            if (!(ABL & 1)) pairs_issue_loads(v, lmrec, rr, dat);
            // (the camera area is free: every lane read its cameras in phase B, before the wave barriers above)
            dma = 1 + __popc(ck.mask & ~1u) <= kPairDmaBlocks;
            if (dma) pairs_dma_issue(v, cam_next, lane, CAMS);
            if (ch + 2 < ch_end) {
                ck_next = chunks[ch + 2];
                rr_next = reinterpret_cast<const uint4*>(recs)[(size_t)(ch + 2) * 64 + lane];
                cam_next = pairs_dma_cam(blocks, ck_next, lane);
            }
        }
        // ---- E: block products.  The chunk is a sequence of segments (runs of pairs of one block); the NG lane groups deal
        // a segment's pairs among themselves, every lane adds its 3 x 3 sub-block of U_p V_p (18 FMA per pair) --------------
        uint32_t mask = ck_cur.mask;
        int seg0 = 0;
        if (mask & 1u) {   // the chunk opens a new block
            if (cur >= 0) pairs_flush<DC>(tiles, cur_dst, cur_flags, acc, lane);
            cur = cur < 0 ? ck_cur.first_block : cur + 1;
            cur_dst = blocks[cur].dst; cur_flags = blocks[cur].flags;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] = 0.0;
        }
        mask &= ~1u;
        for (;;) {
            const int seg1 = mask ? 2 * (__ffs(mask) - 1) : 64;      // wave-uniform
            for (int p = seg0 + g; p < seg1; p += NG) {
                if (worker && !(ABL & 2)) {
                    const double2* qu = reinterpret_cast<const double2*>(U + p * UV + bi * 6);
                    const double2* qv = reinterpret_cast<const double2*>(V + p * UV + bj * 6);
                    const double2 u0 = qu[0], u1 = qu[1], u2 = qu[2], v0 = qv[0], v1 = qv[1], v2 = qv[2];
                    const double um0[3] = {u0.x, u0.y, u1.x}, um1[3] = {u1.y, u2.x, u2.y};
                    const double vm0[3] = {v0.x, v0.y, v1.x}, vm1[3] = {v1.y, v2.x, v2.y};
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[3 * r + c] = fma(um1[r], vm1[c], fma(um0[r], vm0[c], acc[3 * r + c]));
                }
            }
            if (!mask) break;
            pairs_flush<DC>(tiles, cur_dst, cur_flags, acc, lane);
            ++cur;
            cur_dst = blocks[cur].dst; cur_flags = blocks[cur].flags;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] = 0.0;
            seg0 = seg1;
            mask &= mask - 1;
        }
        __builtin_amdgcn_wave_barrier();   // the next chunk's camera staging overwrites U
    }
    if (cur >= 0) pairs_flush<DC>(tiles, cur_dst, cur_flags, acc, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// Variant H ("half-pair lanes"): one OBSERVATION per lane, two lanes per pair, 32 pair slots per step.
// Lane 2p linearises observation i of pair p, lane 2p+1 observation j; the odd lane forms Q = Hll^-1 Jl_j^T (3 x 2) and hands it
// to its neighbour (DPP quad_perm), the even lane forms M = -Jl_i Q and U = Jc_i^T M, the odd lane's V is its Jc_j.  One
// linearisation's temporaries instead of two and half the U / V staging per wave (10 KB of LDS): three waves per SIMD
// instead of two.  Same lists, same product phase (lane = (group, sub-block)), over the two 32-slot halves of a chunk.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dpp_swap1(double x) {   // value of lane ^ 1
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

struct HalfData {     // what a lane needs for its observation, fetched one half-chunk ahead
    double2 uv;
    double2 lm[6];
};
__device__ __forceinline__ void half_issue_loads(const BAView& v, const double* __restrict__ lmrec, const uint4 rr, int side, HalfData& d) {
    const bool valid = rr.x != kPairPad;
    const uint32_t o = valid ? (side ? rr.y : rr.x) : 0u, l = valid ? rr.z : 0u;
    d.uv = v.o_uv[o];
    const double2* q = reinterpret_cast<const double2*>(lmrec + kLmStride * (size_t)l);
#pragma unroll
    for (int k = 0; k < 6; ++k) d.lm[k] = q[k];
}

template <int DC>
__global__ __launch_bounds__(256, 3) void k_schur_pairs_h(BAView v, double* __restrict__ tiles, const PairTask* __restrict__ tasks,
                                                          int n_tasks, const PairChunk* __restrict__ chunks,
                                                          const PairBlock* __restrict__ blocks, const PairRec* __restrict__ recs,
                                                          const double* __restrict__ lmrec) {
    constexpr int UV = 2 * DC;
    constexpr int NB3 = DC / 3;
    constexpr int GL = NB3 * NB3;
    constexpr int NG = (DC == 9) ? 7 : 16;
    // U[32][UV] | skew | V[32][UV] | skew | 8 cameras staged by LDS-DMA.  The skew between U and V: lanes 2p and 2p+1 store
    // U[p] and V[p] in the same instruction, and 32 * UV doubles apart they would hit the same banks
    constexpr int kSkew = 8;   // ds_write_b128: groups of 8 lanes, bank = dword address mod 32 -> V sixteen banks away from U
    constexpr int WAVE_LDS = 2 * (32 * UV + kSkew) + 8 * kCamStride;
    __shared__ double lds_all[4 * WAVE_LDS];
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
    double* V = U + 32 * UV + kSkew;
    double* CAMS = V + 32 * UV + kSkew;
    const PairTask task = tasks[t];
    const int g = lane / GL, sub = lane - g * GL, bi = sub / NB3, bj = sub - bi * NB3;
    const bool worker = g < NG;
    const int side = lane & 1, pl = lane >> 1;     // this lane's observation of pair slot pl of the half
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    int cur = -1;
    int64_t cur_dst = 0;
    uint32_t cur_flags = 0;

    const int ch_end = task.chunk0 + task.nchunks;
    int ch = task.chunk0;
    PairChunk ck = chunks[ch];
    const uint4* rec4 = reinterpret_cast<const uint4*>(recs);
    uint4 rr = rec4[(size_t)ch * 64 + pl];
    HalfData dat;
    half_issue_loads(v, lmrec, rr, side, dat);
    bool dma = 1 + __popc(ck.mask & ~1u) <= kPairDmaBlocks;
    if (dma) pairs_dma_issue(v, pairs_dma_cam(blocks, ck, lane), lane, CAMS);
    // one half ahead: the record; one chunk ahead: the descriptor and the camera this lane stages
    uint4 rr_next = rec4[(size_t)ch * 64 + 32 + pl];
    PairChunk ck_next = ck;
    uint32_t cam_next = 0;
    if (ch + 1 < ch_end) { ck_next = chunks[ch + 1]; cam_next = pairs_dma_cam(blocks, ck_next, lane); }

    for (; ch < ch_end; ++ch) {
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const bool valid = rr.x != kPairPad;
            const uint32_t blk = valid ? rr.w : 0u;
            // ---- B: one observation per lane ---------------------------------------------------------------------------
            double cv[16];
            if (dma) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA writes LDS behind the VM counter
                __builtin_amdgcn_wave_barrier();
                const double2* c2 = reinterpret_cast<const double2*>(CAMS + (2 * blk + side) * kCamStride);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 tq = c2[k]; cv[2 * k] = tq.x; cv[2 * k + 1] = tq.y; }
            } else {   // many tiny blocks in this chunk: the camera comes straight from memory
                const PairBlock* pb = blocks + ck.first_block + blk;
                const uint32_t cam = side ? pb->cj : pb->ci;
                const double2* c2 = reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)cam);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 tq = c2[k]; cv[2 * k] = tq.x; cv[2 * k + 1] = tq.y; }
            }
            double Hi[9], pw[3];
            Hi[0] = dat.lm[0].x; Hi[1] = dat.lm[0].y; Hi[2] = dat.lm[1].x; Hi[3] = dat.lm[1].y; Hi[4] = dat.lm[2].x; Hi[5] = dat.lm[2].y;
            Hi[6] = dat.lm[3].x; Hi[7] = dat.lm[3].y; Hi[8] = dat.lm[4].x; pw[0] = dat.lm[4].y; pw[1] = dat.lm[5].x; pw[2] = dat.lm[5].y;
            double Jc[2][DC], Jl[2][3];
            linearize_pairside<DC>(cv, pw, dat.uv.x, dat.uv.y, v.huber_delta, Jc, Jl);
            // odd lane: Q = Hll^-1 Jl_j^T (3 x 2); every lane forms it, the even lane takes its neighbour's
            double Q[3][2];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int m = 0; m < 2; ++m) Q[a][m] = Hi[3 * a] * Jl[m][0] + Hi[3 * a + 1] * Jl[m][1] + Hi[3 * a + 2] * Jl[m][2];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int m = 0; m < 2; ++m) Q[a][m] = dpp_swap1(Q[a][m]);
            double out[UV];
            // (a padding slot contributes U = 0 through sgn; the record form, k_schur_pairs_r, zeroes both sides by selection --
            // here eighteen more selects push the kernel over its 168 registers into scratch)
            if (side == 0) {
                const double sgn = valid ? -1.0 : 0.0;
                double M[2][2];
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m) M[n][m] = sgn * (Jl[n][0] * Q[0][m] + Jl[n][1] * Q[1][m] + Jl[n][2] * Q[2][m]);
#pragma unroll
                for (int e = 0; e < UV; ++e) {   // U[m][3 bi + c] in the order [bi][m][3]
                    const int s0 = e / 6, m = (e % 6) / 3, c = e % 3;
                    out[e] = Jc[0][3 * s0 + c] * M[0][m] + Jc[1][3 * s0 + c] * M[1][m];
                }
            } else {
#pragma unroll
                for (int e = 0; e < UV; ++e) {   // V[m][3 bj + c] in the order [bj][m][3]
                    const int s0 = e / 6, m = (e % 6) / 3, c = e % 3;
                    out[e] = Jc[m][3 * s0 + c];
                }
            }
            __builtin_amdgcn_wave_barrier();   // the previous half's products are done with U / V (one wave: program order)
            {
                double2* po = reinterpret_cast<double2*>((side ? V : U) + pl * UV);
#pragma unroll
                for (int k = 0; k < UV / 2; ++k) po[k] = make_double2(out[2 * k], out[2 * k + 1]);
            }
            __builtin_amdgcn_wave_barrier();
            // ---- D: the next half's gathers go out now and land during the product phase ----------------------------------
            const PairChunk ck_cur = ck;
            const uint32_t hmask = (ck_cur.mask >> (16 * half)) & 0xFFFFu;
            const bool more = half == 0 || ch + 1 < ch_end;
            if (more) {
                rr = rr_next;
                half_issue_loads(v, lmrec, rr, side, dat);
                if (half == 1) {
                    // (every lane read its cameras of this chunk above: the camera area is free)
                    ck = ck_next;
                    dma = 1 + __popc(ck.mask & ~1u) <= kPairDmaBlocks;
                    if (dma) pairs_dma_issue(v, cam_next, lane, CAMS);
                    rr_next = rec4[(size_t)(ch + 1) * 64 + 32 + pl];
                    if (ch + 2 < ch_end) { ck_next = chunks[ch + 2]; cam_next = pairs_dma_cam(blocks, ck_next, lane); }
                } else {
                    if (ch + 1 < ch_end) rr_next = rec4[(size_t)(ch + 1) * 64 + pl];
                }
            }
            // ---- E: block products over the 32 slots of the half -------------------------------------------------------------
            uint32_t mask = hmask;
            int seg0 = 0;
            if (mask & 1u) {
                if (cur >= 0) pairs_flush<DC>(tiles, cur_dst, cur_flags, acc, lane);
                cur = cur < 0 ? ck_cur.first_block : cur + 1;
                cur_dst = blocks[cur].dst; cur_flags = blocks[cur].flags;
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[k] = 0.0;
            }
            mask &= ~1u;
            for (;;) {
                const int seg1 = mask ? 2 * (__ffs(mask) - 1) : 32;      // wave-uniform
                for (int p = seg0 + g; p < seg1; p += NG) {
                    if (worker) {
                        const double2* qu = reinterpret_cast<const double2*>(U + p * UV + bi * 6);
                        const double2* qv = reinterpret_cast<const double2*>(V + p * UV + bj * 6);
                        const double2 u0 = qu[0], u1 = qu[1], u2 = qu[2], v0 = qv[0], v1 = qv[1], v2 = qv[2];
                        const double um0[3] = {u0.x, u0.y, u1.x}, um1[3] = {u1.y, u2.x, u2.y};
                        const double vm0[3] = {v0.x, v0.y, v1.x}, vm1[3] = {v1.y, v2.x, v2.y};
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 3; ++c) acc[3 * r + c] = fma(um1[r], vm1[c], fma(um0[r], vm0[c], acc[3 * r + c]));
                    }
                }
                if (!mask) break;
                pairs_flush<DC>(tiles, cur_dst, cur_flags, acc, lane);
                ++cur;
                cur_dst = blocks[cur].dst; cur_flags = blocks[cur].flags;
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[k] = 0.0;
                seg0 = seg1;
                mask &= mask - 1;
            }
        }
    }
    if (cur >= 0) pairs_flush<DC>(tiles, cur_dst, cur_flags, acc, lane);
}


// ------------------------------------------------------------------------------------------------------------------
// RECORD FORM (round 3).  The fused kernels above re-linearise both observations of every pair from the 24-byte
// observation records: ~140 fp64 instructions per observation, k - 1 = 5.5 times per observation and iteration -- 40 % of
// the kernel's vector instructions, and it is bound by exactly those.  k_landmark_reduce linearises
// every observation once anyway; it now also writes the observation's PROJECTION RECORD (xn, yn, -1/z, sqrt(rho')), 32
// bytes, in place of which the pair kernel used to gather the 16-byte measurement.  From the record and the camera
// (R, f, k1, k2: staged in LDS as before) the Jacobian is ~60 instructions with no reciprocal, no square root and no
// dependent chain longer than six:
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
__device__ unsigned long long g_pair_phase[8];
void pairs_phase_cycles(unsigned long long out[8], bool reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_phase), 8 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pair_phase), z, sizeof z); }
}
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
// Lane mapping of the record kernel's product phase.  DC = 6: lane = 4 g + sub (16 groups x 4 sub-blocks).  DC = 9: the 9
// sub-blocks x 7 groups are laid out so that the fold over the groups is DPP arithmetic on the vector unit instead of three
// dependent round trips through the LDS crossbar (ds_bpermute): sub-blocks 0..7 own one aligned OCTET of lanes each
// (lane = 8 sub + g, g = 0..6) and sub-block 8 takes the octets' eighth lanes (lane = 8 g + 7; lane 63 idles and ends up
// holding sub-block 8's total).
template <int DC, bool LEGACY = false>
__device__ __forceinline__ void pairs_lane_map(int lane, int& g, int& sub) {
    if (DC == 9 && LEGACY) { g = lane / 9; sub = lane - 9 * g; return; }   // lane = 9 g + sub, fold by ds_bpermute (A/B)
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
template <int DC, bool LEGACY = false, int FABL = 0>   // FABL (timing experiments): 1 no fold over the groups, 2 no stores
__device__ __forceinline__ void pairs_flush2(double* __restrict__ tiles, const int64_t pb_dst, const uint32_t pb_flags,
                                             double acc[9], int lane) {
    constexpr int NB3 = DC / 3, GL = NB3 * NB3;
    constexpr int NG = (DC == 9) ? 7 : 16, P2 = (DC == 9) ? 8 : 16;
    int g, sub;
    pairs_lane_map<DC, LEGACY>(lane, g, sub);
    bool storer;
    if (FABL & 1) {
        storer = DC == 9 ? (lane == 63 || (lane & 7) == 6) : lane < GL;
        if (DC == 9) sub = lane == 63 ? 8 : lane >> 3;
    } else if (DC == 9 && !LEGACY) {
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
    if ((FABL & 2) && acc[0] != 1.2345e300) return;
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

// ABL (timing experiments only, results wrong when != 0): 1 no per-pair gathers, 2 no block products, 4 no U / V stores,
// 8 no flush (fold + store of the finished block) -- NOTE: with the flush gone seven of the nine accumulators are dead and the
// compiler drops their FMAs, so 8 measures "no flush and 7/9 of the products", not the flush (round 4: 512 / 1024 below do) --,
// 512 no fold over the groups, 1024 no stores of the finished block, 16 no camera reads, 32 no Jacobian arithmetic; 64 (results RIGHT): phase
// stamps -- every wave adds the shader cycles it spent in each phase of its chunks to g_pair_phase (read with
// pairs_phase_cycles): 0 wait for the gathers, 1 unstage, 2 cameras + both Jacobians + U / V stores, 3 issue, 4 products and
// flushes, 5 chunks, 6 flushes alone, 7 number of flushes
//
// Nothing in the loop goes through the scalar memory path: s_load shares the lgkm counter with the LDS and returns out of
// order, so one descriptor load in flight turns every LDS wait of the product loop into a wait for memory (the first
// version of this kernel stalled ~1 us per chunk on its own chunk descriptor).  The task's chunk descriptors are loaded
// once, one per lane, and read with v_readlane; a chunk's block descriptors (destination, flags) come one chunk ahead as
// a vector load by the first lanes and are read the same way.
template <int DC, bool MASKED, int ABL = 0>
__global__ __launch_bounds__(256, 2) void k_schur_pairs_r(BAView v, double* __restrict__ tiles, const PairTask* __restrict__ tasks,
                                                          int n_tasks, const PairChunk* __restrict__ chunks,
                                                          const PairBlock* __restrict__ blocks, const PairRec* __restrict__ recs,
                                                          const double* __restrict__ lmrec, const double* __restrict__ orec) {
    constexpr int UV = 2 * DC;
    constexpr int NB3 = DC / 3;
    constexpr int GL = NB3 * NB3;
    constexpr int NG = (DC == 9) ? 7 : 16;
    constexpr int WAVE_LDS = 2 * 64 * UV + UV;   // U[64][UV] | V[64][UV] | zeros[UV]
    __shared__ double lds_all[4 * WAVE_LDS];
    // The chunk's <= 8 cameras are staged through REGISTERS (one 16-byte load per lane a chunk ahead, one ds_write at the top
    // of the chunk), not by LDS-DMA as in the fused kernels above: global_load_lds writes LDS behind the VECTOR-MEMORY
    // counter, the compiler cannot tell its destination from U / V and puts s_waitcnt vmcnt(0) in front of every LDS read of
    // the product loop -- which then waits for the very gathers that were issued to overlap with it (the fused kernels have
    // exactly this: their prefetch never overlapped their products).
    __shared__ double lds_cams[4 * 8 * kCamStride];
    __shared__ double lds_occ[(ABL & 128) ? 6000 : 1];   // ABL 128: +47 KB of LDS = ONE workgroup per CU (occupancy experiment)
    if ((ABL & 128) && n_tasks == -12345) { lds_occ[threadIdx.x * 23] = 1.0; __syncthreads(); tiles[0] = lds_occ[threadIdx.x * 7 + 1]; }
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
    double* CAMS = lds_cams + w * 8 * kCamStride;
    double* Z = V + 64 * UV;
    if (lane < UV) Z[lane] = 0.0;
    int g, sub;
    pairs_lane_map<DC, (ABL & 256) != 0>(lane, g, sub);
    const int bi = sub / NB3, bj = sub - bi * NB3;
    const bool worker = g < NG;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    int cur = -1;
    int64_t cur_dst = 0;
    uint32_t cur_flags = 0;
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
    // The per-pair gathers (landmark record 96 B, the two projection records 32 B each) are COOPERATIVE: the L1 serves one
    // 64-byte line per clock whatever the lanes take from it, and a lane that fetches its own 160 bytes as ten 16-byte loads
    // costs ten line accesses per pair -- 1.6 ms of pure tag-lookup time per launch, the largest single item of the first
    // record kernel (profiles/r03_pairs_ablation.txt).  Here four lanes share a 64-byte line in ONE instruction (two lanes a
    // 32-byte record): four line accesses per pair.  The pieces land in the registers of the lanes that fetched them and go
    // to the lanes that need them through the LDS area U / V leave free between two product phases (stage / unstage).
    struct Coop { double2 a0, a1, a2, a3, b0, b1, i0, i1, j0, j1; };   // what a lane fetches for OTHER lanes' pairs (scalars: arrays in a loop-carried struct went to scratch memory)
    struct Gather { double2 ri0, ri1, rj0, rj1; double2 lm[6]; };  // a lane's own pair
    constexpr int kStage = 176;   // bytes per pair in the staging image: 160 of data; 44 dwords = 4 x odd keeps the b128 reads conflict-free
    static_assert(64 * kStage <= 2 * 64 * UV * 8, "the staging image must fit the U / V area");
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
        const int i = valid ? (int)rr.x : 0, j = valid ? (int)rr.y : 0, l = valid ? (int)rr.z : 0;   // padding lanes read element 0
        const int qq = lane & 3, h = lane & 1;
        auto lm0 = [&](auto sel) { return *reinterpret_cast<const double2*>(lmrec + kLmStride * (size_t)quad_bcast(l, sel) + 2 * qq); };
        d.a0 = lm0(std::integral_constant<int, 0>{}); d.a1 = lm0(std::integral_constant<int, 1>{});
        d.a2 = lm0(std::integral_constant<int, 2>{}); d.a3 = lm0(std::integral_constant<int, 3>{});
        {
            const std::integral_constant<int, 0> k{};
            d.b0 = *reinterpret_cast<const double2*>(lmrec + kLmStride * (size_t)pair_bcast(l, k) + 8 + 2 * h);
            d.i0 = *reinterpret_cast<const double2*>(orec + 4 * (size_t)pair_bcast(i, k) + 2 * h);
            d.j0 = *reinterpret_cast<const double2*>(orec + 4 * (size_t)pair_bcast(j, k) + 2 * h);
        }
        {
            const std::integral_constant<int, 1> k{};
            d.b1 = *reinterpret_cast<const double2*>(lmrec + kLmStride * (size_t)pair_bcast(l, k) + 8 + 2 * h);
            d.i1 = *reinterpret_cast<const double2*>(orec + 4 * (size_t)pair_bcast(i, k) + 2 * h);
            d.j1 = *reinterpret_cast<const double2*>(orec + 4 * (size_t)pair_bcast(j, k) + 2 * h);
        }
    };
    auto unstage = [&](const Coop& d, Gather& o) {   // the U / V area is free: no product phase is running
        char* T = reinterpret_cast<char*>(U);
        char* A = T + kStage * (lane & ~3) + 16 * (lane & 3);      // pair (lane & ~3) + k, quarter lane & 3
        *reinterpret_cast<double2*>(A) = d.a0;
        *reinterpret_cast<double2*>(A + kStage) = d.a1;
        *reinterpret_cast<double2*>(A + 2 * kStage) = d.a2;
        *reinterpret_cast<double2*>(A + 3 * kStage) = d.a3;
        char* R = T + kStage * (lane & ~1) + 16 * (lane & 1);      // pair (lane & ~1) + k, half lane & 1
        *reinterpret_cast<double2*>(R + 64) = d.b0;
        *reinterpret_cast<double2*>(R + 96) = d.i0;
        *reinterpret_cast<double2*>(R + 128) = d.j0;
        *reinterpret_cast<double2*>(R + kStage + 64) = d.b1;
        *reinterpret_cast<double2*>(R + kStage + 96) = d.i1;
        *reinterpret_cast<double2*>(R + kStage + 128) = d.j1;
        __builtin_amdgcn_wave_barrier();
        const double2* me = reinterpret_cast<const double2*>(T + kStage * lane);
#pragma unroll
        for (int k = 0; k < 6; ++k) o.lm[k] = me[k];
        o.ri0 = me[6]; o.ri1 = me[7]; o.rj0 = me[8]; o.rj1 = me[9];
        __builtin_amdgcn_wave_barrier();   // (the LDS executes a wave's operations in order: the U / V stores below come after these reads)
    };
    // block descriptors of a chunk: lane b < nblk holds block first_block + b (destination offset, flags)
    struct BlockDesc { int2 dst; uint32_t flags; };
    auto load_blocks = [&](const PairChunk c, BlockDesc& b) {
        const int nblk = 1 + __popc(c.mask & ~1u);
        const PairBlock* pb = blocks + c.first_block + min(lane, nblk - 1);
        b.dst = *reinterpret_cast<const int2*>(&pb->dst);
        b.flags = pb->flags;
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
    load_blocks(ck, bd);
    bd_next = bd;
    bool dma = 1 + __popc(ck.mask & ~1u) <= kPairDmaBlocks;
    auto cam_piece = [&](uint32_t cam) {   // lanes 8 c .. 8 c + 7 fetch the 128-byte prepared camera c of the chunk
        return *reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)cam + 2 * (lane & 7));
    };
    double2 cam_stage = cam_piece(pairs_dma_cam(blocks, ck, lane));
    PairChunk ck_next = chunk_desc(min(1, nchunks - 1));
    uint4 rr_next = rec4[(size_t)(chunk0 + min(1, nchunks - 1)) * 64 + lane];
    uint32_t cam_next = pairs_dma_cam(blocks, ck_next, lane);

    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&]() -> unsigned long long { return (ABL & 64) ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull; };
    for (; q < nchunks; ++q) {
        const bool valid = rr.x != kPairPad;
        const uint32_t blk = valid ? rr.w : 0u;
        const unsigned long long t0 = stamp();
        // ---- cameras of this lane's pair: LDS (DMA issued a chunk ago) or, in a chunk of many tiny blocks, memory ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // gathers, block descriptors, and the LDS-DMA (it writes LDS behind the VM counter)
        reinterpret_cast<double2*>(CAMS)[lane] = cam_stage;   // (the previous chunk's camera reads are long done)
        __builtin_amdgcn_wave_barrier();
        const unsigned long long t1 = stamp();
        Gather dat;
        unstage(coop, dat);
        if (ABL & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t2 = stamp();
        double cvi[16], cvj[16];
        if (ABL & 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { cvi[k] = dat.lm[k % 6].x + k; cvj[k] = dat.lm[k % 6].y - k; }
        } else {
            if (dma) {
                const double2* ci = reinterpret_cast<const double2*>(CAMS + (2 * blk) * kCamStride);
                const double2* cj = reinterpret_cast<const double2*>(CAMS + (2 * blk + 1) * kCamStride);
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
        Hi[0] = dat.lm[0].x; Hi[1] = dat.lm[0].y; Hi[2] = dat.lm[1].x; Hi[3] = dat.lm[1].y; Hi[4] = dat.lm[2].x; Hi[5] = dat.lm[2].y;
        Hi[6] = dat.lm[3].x; Hi[7] = dat.lm[3].y; Hi[8] = dat.lm[4].x; pw[0] = dat.lm[4].y; pw[1] = dat.lm[5].x; pw[2] = dat.lm[5].y;
        // A padding slot (and the filler slot of an odd block) contributes exact zeros: its lanes carry element 0's records
        // with the Huber weight forced to 0, which zeroes a, (xn w, yn w) and with them V, M and U.
        const double2 rj1 = make_double2(dat.rj1.x, valid ? dat.rj1.y : 0.0), ri1 = make_double2(dat.ri1.x, valid ? dat.ri1.y : 0.0);
        // ---- column side j: V = Jc_j, Q = Hll^-1 Jl_j^T ---------------------------------------------------------------
        double Q[3][2];
        {
            RecJac J;
            if (ABL & 32) { J.a[0][0] = dat.rj0.x; J.a[0][1] = dat.rj0.y; J.a[0][2] = rj1.x; J.a[1][0] = rj1.y; J.a[1][1] = cvj[0]; J.a[1][2] = cvj[1]; J.xw = cvj[2]; J.yw = cvj[3]; J.t[0] = cvj[4]; J.t[1] = cvj[5]; J.t[2] = cvj[6]; }
            else jac_from_rec(cvj, dat.rj0, rj1, J);
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
                if (!(ABL & 4) || k == 0) pv[k] = make_double2(Jc[m0][3 * s0 + c0], Jc[m1][3 * s1 + c1]);
                else asm volatile("" ::"v"(Jc[m0][3 * s0 + c0]), "v"(Jc[m1][3 * s1 + c1]));
            }
        }
        // ---- row side i: M = -Jl_i Q, U = Jc_i^T M = [G ; p_w x G ; t (s M)], G = a_i^T M -----------------------------------
        {
            RecJac J;
            if (ABL & 32) { J.a[0][0] = dat.ri0.x; J.a[0][1] = dat.ri0.y; J.a[0][2] = ri1.x; J.a[1][0] = ri1.y; J.a[1][1] = cvi[0]; J.a[1][2] = cvi[1]; J.xw = cvi[2]; J.yw = cvi[3]; J.t[0] = cvi[4]; J.t[1] = cvi[5]; J.t[2] = cvi[6]; }
            else jac_from_rec(cvi, dat.ri0, ri1, J);
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
            for (int k = 0; k < UV / 2; ++k) {
                if (!(ABL & 4) || k == 0) pu[k] = make_double2(u[2 * k], u[2 * k + 1]);
                else asm volatile("" ::"v"(u[2 * k]), "v"(u[2 * k + 1]));
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (ABL & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t3 = stamp();
        // ---- the next chunk's gathers, block descriptors and cameras go out now and land during the product phase ------------
        const PairChunk ck_cur = ck;
        const BlockDesc bd_cur = bd_next;     // (the descriptors of THIS chunk: loaded a chunk ago, or before the loop)
        {
            // Branch-free and clamped (the last chunks of a task fetch their own data again): a conditional load into a
            // loop-carried register makes the compiler resolve the phi with a copy right behind the load -- and wait for
            // every outstanding load (s_waitcnt vmcnt(0)) to do it, i.e. for the gathers issued three lines earlier.
            ck = ck_next; rr = rr_next;
            if (!(ABL & 1)) issue(rr, coop);
            load_blocks(ck, bd_next);
            dma = 1 + __popc(ck.mask & ~1u) <= kPairDmaBlocks;
            cam_stage = cam_piece(cam_next);
            const int q2 = min(q + 2, nchunks - 1);
            ck_next = chunk_desc(q2);
            rr_next = rec4[(size_t)(chunk0 + q2) * 64 + lane];
            cam_next = pairs_dma_cam(blocks, ck_next, lane);
        }
        const unsigned long long t4 = stamp();
        // ---- block products over the 64 slots.  Uniform loop, two pairs per trip with all twelve operand reads issued up
        // front; a lane whose pair lies beyond the segment (and the idle 64th lane) reads the zero row instead.
        uint32_t mask = ck_cur.mask;
        int seg0 = 0, lb = 0;                 // lb: index of the running block inside this chunk's descriptors
        auto start_block = [&]() {
            cur_dst = ((int64_t)__builtin_amdgcn_readlane(bd_cur.dst.y, lb) << 32) | (uint32_t)__builtin_amdgcn_readlane(bd_cur.dst.x, lb);
            cur_flags = (uint32_t)__builtin_amdgcn_readlane((int)bd_cur.flags, lb);
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] = 0.0;
        };
        auto flush = [&]() {
            const unsigned long long f0 = stamp();
            if (!(ABL & 8)) pairs_flush2<DC, (ABL & 256) != 0, (ABL >> 9) & 3>(tiles, cur_dst, cur_flags, acc, lane);
            else if (acc[0] == 1.2345e300) tiles[0] = acc[1];
            if (ABL & 64) { ph[6] += stamp() - f0; ph[7] += 1; }
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
            if (!(ABL & 2)) {
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
        __builtin_amdgcn_wave_barrier();
        bd = bd_next;
        if (ABL & 64) {
            const unsigned long long t5 = stamp();
            ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; ph[5] += 1;
        }
    }
    if ((ABL & 64) && lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&g_pair_phase[k], ph[k]);
    }
    if (cur >= 0) {
        if (!(ABL & 8)) pairs_flush2<DC, (ABL & 256) != 0>(tiles, cur_dst, cur_flags, acc, lane);
        else if (acc[0] == 1.2345e300) tiles[0] = acc[1];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Record form, TWO LANES PER PAIR (lane 2p: observation i, lane 2p + 1: observation j of pair slot p; 32 pairs per step,
// two steps per chunk).  Same data, same product phase and flush as k_schur_pairs_r; half the U / V staging per wave
// (10.4 KB) and half the gather registers, i.e. THREE waves per SIMD instead of two -- the record kernels are bound by
// how much latency the resident waves can cover (one workgroup per CU instead of two: 5.1 instead of 3.5 ms), not by any
// one unit.  Each lane fetches its own projection record and HALF of the pair's landmark record (the odd lane Hll^-1
// rows 0-1, the even lane row 2 and the point); the odd lane, which alone needs Hll^-1, gets the rest by DPP.
// Odd lane: V = Jc_j and Q = Hll^-1 Jl_j^T, handed to its neighbour by DPP; even lane: M = -Jl_i Q, U = Jc_i^T M.
// ------------------------------------------------------------------------------------------------------------------
template <int DC, bool MASKED>
__global__ __launch_bounds__(256, 3) void k_schur_pairs_r2(BAView v, double* __restrict__ tiles, const PairTask* __restrict__ tasks,
                                                           int n_tasks, const PairChunk* __restrict__ chunks,
                                                           const PairBlock* __restrict__ blocks, const PairRec* __restrict__ recs,
                                                           const double* __restrict__ lmrec, const double* __restrict__ orec) {
    constexpr int UV = 2 * DC;
    constexpr int NB3 = DC / 3;
    constexpr int NG = (DC == 9) ? 7 : 16;
    constexpr int kSkew = 8;   // (see k_schur_pairs_h: lanes 2p and 2p+1 store U[p] and V[p] in the same instruction)
    constexpr int WAVE_LDS = 2 * (32 * UV + kSkew) + UV;   // U[32][UV] | skew | V[32][UV] | skew | zeros[UV]
    __shared__ double lds_all[4 * WAVE_LDS];
    __shared__ double lds_cams[4 * 8 * kCamStride];        // staged through registers, never by LDS-DMA (see k_schur_pairs_r)
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
    double* V = U + 32 * UV + kSkew;
    double* Z = V + 32 * UV + kSkew;
    double* CAMS = lds_cams + w * 8 * kCamStride;
    if (lane < UV) Z[lane] = 0.0;
    int g, sub;
    pairs_lane_map<DC>(lane, g, sub);
    const int bi = sub / NB3, bj = sub - bi * NB3;
    const bool worker = g < NG;
    const int side = lane & 1, pl = lane >> 1;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    int cur = -1;
    int64_t cur_dst = 0;
    uint32_t cur_flags = 0;
    const double mp = MASKED ? ((v.mask_code & 4) ? 1.0 : 0.0) : 1.0, ml = MASKED ? ((v.mask_code & 2) ? 1.0 : 0.0) : 1.0,
                 mi = MASKED ? ((v.mask_code & 1) ? 1.0 : 0.0) : 1.0;
    int chunk0, nchunks;
    {
        const int2 tk = reinterpret_cast<const int2*>(tasks)[t];
        chunk0 = __builtin_amdgcn_readfirstlane(tk.x); nchunks = __builtin_amdgcn_readfirstlane(tk.y);
    }
    const uint2 ckv = reinterpret_cast<const uint2*>(chunks)[(size_t)chunk0 + min(lane, nchunks - 1)];   // <= 64 chunks per task
    auto chunk_desc = [&](int q) -> PairChunk {
        PairChunk c;
        c.mask = (uint32_t)__builtin_amdgcn_readlane((int)ckv.x, q);
        c.first_block = __builtin_amdgcn_readlane((int)ckv.y, q);
        return c;
    };
    struct Half { double2 r0, r1, m0, m1, m2; };   // own projection record; odd lane: Hll^-1[0..5], even lane: Hll^-1[6..8] | point
    auto issue = [&](const uint4 rr, Half& d) {
        const bool valid = rr.x != kPairPad;
        const uint32_t o = valid ? (side ? rr.y : rr.x) : 0u, l = valid ? rr.z : 0u;   // padding lanes read element 0
        const double2* qo = reinterpret_cast<const double2*>(orec + 4 * (size_t)o);
        d.r0 = qo[0]; d.r1 = qo[1];
        const double2* ql = reinterpret_cast<const double2*>(lmrec + kLmStride * (size_t)l) + (side ? 0 : 3);
        d.m0 = ql[0]; d.m1 = ql[1]; d.m2 = ql[2];
    };
    struct BlockDesc { int2 dst; uint32_t flags; };
    auto load_blocks = [&](const PairChunk c, BlockDesc& b) {
        const int nblk = 1 + __popc(c.mask & ~1u);
        const PairBlock* pb = blocks + c.first_block + min(lane, nblk - 1);
        b.dst = *reinterpret_cast<const int2*>(&pb->dst);
        b.flags = pb->flags;
    };
    auto cam_piece = [&](uint32_t cam) {
        return *reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)cam + 2 * (lane & 7));
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ckv
    const uint4* rec4 = reinterpret_cast<const uint4*>(recs);
    const int n_half = 2 * nchunks;
    auto rec_of = [&](int hh) { return rec4[(size_t)(chunk0 + (hh >> 1)) * 64 + (hh & 1) * 32 + pl]; };
    // prologue: half 0's gathers, half 1's record; chunk 0's cameras and block descriptors, chunk 1's camera indices
    PairChunk ck = chunk_desc(0);
    uint4 rr = rec_of(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Half dat;
    issue(rr, dat);
    BlockDesc bd, bd_next;
    load_blocks(ck, bd);
    bd_next = bd;
    double2 cam_stage = cam_piece(pairs_dma_cam(blocks, ck, lane));
    uint4 rr_next = rec_of(min(1, n_half - 1));
    PairChunk ck_next = chunk_desc(min(1, nchunks - 1));
    uint32_t cam_next = pairs_dma_cam(blocks, ck_next, lane);
    bool dma = 1 + __popc(ck.mask & ~1u) <= kPairDmaBlocks;

    for (int q = 0; q < nchunks; ++q) {
        BlockDesc bd_cur = bd;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int hh = 2 * q + half;
            const bool valid = rr.x != kPairPad;
            const uint32_t blk = valid ? rr.w : 0u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this half's gathers (and, at half 0, the chunk's cameras / descriptors)
            if (half == 0) {
                reinterpret_cast<double2*>(CAMS)[lane] = cam_stage;   // (the previous chunk's camera reads are long done)
                bd_cur = bd;
            }
            __builtin_amdgcn_wave_barrier();
            double cv[16];
            if (dma) {
                const double2* c2 = reinterpret_cast<const double2*>(CAMS + (2 * blk + side) * kCamStride);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 a = c2[k]; cv[2 * k] = a.x; cv[2 * k + 1] = a.y; }
            } else {   // a chunk of many tiny blocks: the camera comes straight from memory
                const PairBlock* pb = blocks + ck.first_block + blk;
                const double2* c2 = reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)(side ? pb->cj : pb->ci));
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 a = c2[k]; cv[2 * k] = a.x; cv[2 * k + 1] = a.y; }
            }
            // the landmark record: the odd lane completes Hll^-1 with its neighbour's pieces, both lanes take the point
            const double n0x = dpp_swap1(dat.m0.x), n0y = dpp_swap1(dat.m0.y), n1x = dpp_swap1(dat.m1.x), n1y = dpp_swap1(dat.m1.y),
                         n2x = dpp_swap1(dat.m2.x), n2y = dpp_swap1(dat.m2.y);
            const double Hi[9] = {dat.m0.x, dat.m0.y, dat.m1.x, dat.m1.y, dat.m2.x, dat.m2.y, n0x, n0y, n1x};   // (meaningful in the odd lane)
            const double pw[3] = {side ? n1y : dat.m1.y, side ? n2x : dat.m2.x, side ? n2y : dat.m2.y};
            RecJac J;
            jac_from_rec(cv, dat.r0, make_double2(dat.r1.x, valid ? dat.r1.y : 0.0), J);   // a padding slot: weight 0 = exact zeros
            double Q[3][2];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int m = 0; m < 2; ++m) Q[a][m] = dpp_swap1((Hi[3 * a] * J.a[m][0] + Hi[3 * a + 1] * J.a[m][1] + Hi[3 * a + 2] * J.a[m][2]) * ml);
            double out[UV];
            if (side == 0) {   // U = Jc_i^T M = [G ; p_w x G ; t (s M)], G = a_i^T M, M = -Jl_i Q
                double M[2][2];
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m) M[n][m] = -((J.a[n][0] * Q[0][m] + J.a[n][1] * Q[1][m] + J.a[n][2] * Q[2][m]) * ml);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const double g0 = (J.a[0][0] * M[0][m] + J.a[1][0] * M[1][m]) * mp;
                    const double g1 = (J.a[0][1] * M[0][m] + J.a[1][1] * M[1][m]) * mp;
                    const double g2 = (J.a[0][2] * M[0][m] + J.a[1][2] * M[1][m]) * mp;
                    out[0 * 6 + m * 3 + 0] = g0; out[0 * 6 + m * 3 + 1] = g1; out[0 * 6 + m * 3 + 2] = g2;
                    out[1 * 6 + m * 3 + 0] = fma(g2, pw[1], -(g1 * pw[2]));
                    out[1 * 6 + m * 3 + 1] = fma(g0, pw[2], -(g2 * pw[0]));
                    out[1 * 6 + m * 3 + 2] = fma(g1, pw[0], -(g0 * pw[1]));
                    if (DC == 9) {
                        const double sm = (J.xw * M[0][m] + J.yw * M[1][m]) * mi;
                        out[2 * 6 + m * 3 + 0] = sm * J.t[0]; out[2 * 6 + m * 3 + 1] = sm * J.t[1]; out[2 * 6 + m * 3 + 2] = sm * J.t[2];
                    }
                }
            } else {           // V = Jc_j in the order [bj][m][3]
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const double a0 = J.a[m][0], a1 = J.a[m][1], a2 = J.a[m][2];
                    out[0 * 6 + m * 3 + 0] = a0 * mp; out[0 * 6 + m * 3 + 1] = a1 * mp; out[0 * 6 + m * 3 + 2] = a2 * mp;
                    out[1 * 6 + m * 3 + 0] = fma(a2, pw[1], -(a1 * pw[2])) * mp;
                    out[1 * 6 + m * 3 + 1] = fma(a0, pw[2], -(a2 * pw[0])) * mp;
                    out[1 * 6 + m * 3 + 2] = fma(a1, pw[0], -(a0 * pw[1])) * mp;
                    if (DC == 9) {
                        const double sw = (m == 0 ? J.xw : J.yw) * mi;
                        out[2 * 6 + m * 3 + 0] = sw * J.t[0]; out[2 * 6 + m * 3 + 1] = sw * J.t[1]; out[2 * 6 + m * 3 + 2] = sw * J.t[2];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // the previous half's products are done with U / V (one wave: program order)
            {
                double2* po = reinterpret_cast<double2*>((side ? V : U) + pl * UV);
#pragma unroll
                for (int k = 0; k < UV / 2; ++k) po[k] = make_double2(out[2 * k], out[2 * k + 1]);
            }
            __builtin_amdgcn_wave_barrier();
            // ---- the next half's gathers go out now and land during the product phase; branch-free and clamped (see k_schur_pairs_r)
            const PairChunk ck_cur = ck;
            const uint32_t hmask = (ck_cur.mask >> (16 * half)) & 0xFFFFu;
            rr = rr_next;
            issue(rr, dat);
            rr_next = rec_of(min(hh + 2, n_half - 1));
            if (half == 1) {
                ck = ck_next;
                dma = 1 + __popc(ck.mask & ~1u) <= kPairDmaBlocks;
                cam_stage = cam_piece(cam_next);
                load_blocks(ck, bd);
                ck_next = chunk_desc(min(q + 2, nchunks - 1));
                cam_next = pairs_dma_cam(blocks, ck_next, lane);
            }
            // ---- block products over the 32 slots of the half ----------------------------------------------------------------
            uint32_t mask = hmask;
            int seg0 = 0;
            static_assert(sizeof(int) == 4, "");
            auto start_block = [&](int lbq) {
                cur_dst = ((int64_t)__builtin_amdgcn_readlane(bd_cur.dst.y, lbq) << 32) | (uint32_t)__builtin_amdgcn_readlane(bd_cur.dst.x, lbq);
                cur_flags = (uint32_t)__builtin_amdgcn_readlane((int)bd_cur.flags, lbq);
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[k] = 0.0;
            };
            // index, inside the chunk's descriptors, of the block running at the start of this half
            int lb = half == 0 ? 0 : __popc(ck_cur.mask & 0xFFFEu);
            if (mask & 1u) {   // the half opens a new block (at half 1, bit 16 of the chunk mask: one more block than counted above)
                if (cur >= 0) pairs_flush2<DC>(tiles, cur_dst, cur_flags, acc, lane);
                if (half == 1) ++lb;
                cur = 0;
                start_block(lb);
            } else if (cur < 0) {
                cur = 0;
                start_block(lb);
            }
            mask &= ~1u;
            for (;;) {
                const int seg1 = mask ? 2 * (__ffs(mask) - 1) : 32;      // wave-uniform
                {
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
                    if (it < seg1) {
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
                }
                if (!mask) break;
                pairs_flush2<DC>(tiles, cur_dst, cur_flags, acc, lane);
                ++lb;
                start_block(lb);
                seg0 = seg1;
                mask &= mask - 1;
            }
        }
    }
    if (cur >= 0) pairs_flush2<DC>(tiles, cur_dst, cur_flags, acc, lane);
}

// variant: 1 (default) one observation per lane (k_schur_pairs_h); 0: one pair per lane (k_schur_pairs).
// ablation: timing experiments only (tools/schur_bench.py --abl; results are WRONG when != 0).  Both are per-solver
// state handed in by the caller: nothing process-wide that one handle could leave behind for the next.
void launch_schur_pairs(int dc, const BAView& v, double* tiles, const PairTask* tasks, int n_tasks, const PairChunk* chunks,
                        const PairBlock* blocks, const PairRec* recs, const double* lmrec, hipStream_t s, int variant, int ablation,
                        const double* orec) {
    if (n_tasks == 0) return;
    const unsigned grid = (unsigned)((n_tasks + 3) / 4);
    const int g_pairs_variant = variant, g_pairs_ablation = ablation;
    if (variant >= 2 && orec && ablation != 0 && dc == 9) {   // timing experiments on the record form (SelfCalibration only)
#define PAIRS_RA(A) case A: hipLaunchKernelGGL((k_schur_pairs_r<9, false, A>), dim3(grid), dim3(256), 0, s, v, tiles, tasks, n_tasks, chunks, blocks, recs, lmrec, orec); break
        switch (ablation) { PAIRS_RA(512); PAIRS_RA(1024); PAIRS_RA(256); PAIRS_RA(128); PAIRS_RA(64); PAIRS_RA(1); PAIRS_RA(2); PAIRS_RA(4); PAIRS_RA(8); PAIRS_RA(16); PAIRS_RA(32); PAIRS_RA(6); PAIRS_RA(14); PAIRS_RA(15); PAIRS_RA(47); PAIRS_RA(63); PAIRS_RA(3); default: ablation = 0; break; }   // (an unlisted value: the plain kernel below, never a missing launch)
#undef PAIRS_RA
        if (ablation != 0) return;
    }
    if (variant >= 2 && orec && ablation == 0) {   // record form (needs k_landmark_reduce's projection records)
        const bool masked = v.mask_code != (dc == 9 ? 7 : 6);
        if (variant == 3) {   // two lanes per pair, three waves per SIMD
#define PAIRS_R2(DCV, MK) hipLaunchKernelGGL((k_schur_pairs_r2<DCV, MK>), dim3(grid), dim3(256), 0, s, v, tiles, tasks, n_tasks, chunks, blocks, recs, lmrec, orec)
            if (dc == 9) { if (masked) PAIRS_R2(9, true); else PAIRS_R2(9, false); }
            else { if (masked) PAIRS_R2(6, true); else PAIRS_R2(6, false); }
#undef PAIRS_R2
            return;
        }
#define PAIRS_R(DCV, MK) hipLaunchKernelGGL((k_schur_pairs_r<DCV, MK>), dim3(grid), dim3(256), 0, s, v, tiles, tasks, n_tasks, chunks, blocks, recs, lmrec, orec)
        if (dc == 9) { if (masked) PAIRS_R(9, true); else PAIRS_R(9, false); }
        else { if (masked) PAIRS_R(6, true); else PAIRS_R(6, false); }
#undef PAIRS_R
        return;
    }
    if (g_pairs_variant == 1 && g_pairs_ablation == 0) {
        if (dc == 9) hipLaunchKernelGGL((k_schur_pairs_h<9>), dim3(grid), dim3(256), 0, s, v, tiles, tasks, n_tasks, chunks, blocks, recs, lmrec);
        else hipLaunchKernelGGL((k_schur_pairs_h<6>), dim3(grid), dim3(256), 0, s, v, tiles, tasks, n_tasks, chunks, blocks, recs, lmrec);
        return;
    }
#define PAIRS_LAUNCH(DCV, A) hipLaunchKernelGGL((k_schur_pairs<DCV, A>), dim3(grid), dim3(256), 0, s, v, tiles, tasks, n_tasks, chunks, blocks, recs, lmrec)
    if (dc == 9) {
        switch (g_pairs_ablation) {
            case 1: PAIRS_LAUNCH(9, 1); break;
            case 2: PAIRS_LAUNCH(9, 2); break;
            case 4: PAIRS_LAUNCH(9, 4); break;
            case 6: PAIRS_LAUNCH(9, 6); break;
            case 7: PAIRS_LAUNCH(9, 7); break;
            case 15: PAIRS_LAUNCH(9, 15); break;
            case 23: PAIRS_LAUNCH(9, 23); break;
            case 31: PAIRS_LAUNCH(9, 31); break;
            default: PAIRS_LAUNCH(9, 0);
        }
    } else {
        PAIRS_LAUNCH(6, 0);
    }
#undef PAIRS_LAUNCH
}

}  // namespace apex
