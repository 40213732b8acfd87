// schur_pairs.hip -- see schur_pairs.h
#include "schur_pairs.h"

#include <algorithm>

#include "ba_device.hpp"
#include "host_parallel.h"

namespace apex {


constexpr int kPairTaskSlots = 1536;     // a wave's task is closed once it holds this many slots (24 chunks: the pipeline's
                                         // prologue -- three dependent loads -- is paid once per task)
constexpr int kPairMaxBlockSlots = 8192; // a block with more slots is split over several waves (atomic flush)
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
    const int kTask = task_slots > 0 ? (task_slots + 63) / 64 * 64 : kPairTaskSlots;
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
            // A padding slot (and the filler slot of an odd block) contributes EXACT zeros on both sides, by selection: its
            // lanes linearised element 0 against an unrelated camera, and 0 * (a non-finite value) would poison the block.
            if (side == 0) {
                double M[2][2];
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m) M[n][m] = -(Jl[n][0] * Q[0][m] + Jl[n][1] * Q[1][m] + Jl[n][2] * Q[2][m]);
#pragma unroll
                for (int e = 0; e < UV; ++e) {   // U[m][3 bi + c] in the order [bi][m][3]
                    const int s0 = e / 6, m = (e % 6) / 3, c = e % 3;
                    const double uv = Jc[0][3 * s0 + c] * M[0][m] + Jc[1][3 * s0 + c] * M[1][m];
                    out[e] = valid ? uv : 0.0;
                }
            } else {
#pragma unroll
                for (int e = 0; e < UV; ++e) {   // V[m][3 bj + c] in the order [bj][m][3]
                    const int s0 = e / 6, m = (e % 6) / 3, c = e % 3;
                    out[e] = valid ? Jc[m][3 * s0 + c] : 0.0;
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

// variant: 1 (default) one observation per lane (k_schur_pairs_h); 0: one pair per lane (k_schur_pairs).
// ablation: timing experiments only (tools/schur_bench.py --abl; results are WRONG when != 0).  Both are per-solver
// state handed in by the caller: nothing process-wide that one handle could leave behind for the next.
void launch_schur_pairs(int dc, const BAView& v, double* tiles, const PairTask* tasks, int n_tasks, const PairChunk* chunks,
                        const PairBlock* blocks, const PairRec* recs, const double* lmrec, hipStream_t s, int variant, int ablation) {
    if (n_tasks == 0) return;
    const unsigned grid = (unsigned)((n_tasks + 3) / 4);
    const int g_pairs_variant = variant, g_pairs_ablation = ablation;
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
