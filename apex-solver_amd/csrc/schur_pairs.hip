// schur_pairs.hip -- see schur_pairs.h
#include "schur_pairs.h"

#include <algorithm>

#include "ba_device.hpp"
#include "host_parallel.h"

namespace apex {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kPairTaskSlots = 768;      // a wave's task is closed once it holds this many slots (12 chunks)
constexpr int kPairMaxBlockSlots = 8192; // a block with more slots is split over several waves (atomic flush)
constexpr int kPairCamPitch = 18;        // doubles per staged camera: 144 B keeps 16-byte alignment and spreads the banks

// ------------------------------------------------------------------------------------------------------------------
// host: the sorted pair list
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct RawPair { uint32_t cj, i, j; };

}  // namespace

void build_pair_lists(int dc, int nt, const int* slot, int64_t n_cam, const int* cam_ext, const uint32_t* o_cam,
                      const uint32_t* o_pt, const int* pt_ptr, const int* cam_ptr, const int* cam_obs, PairLists* out) {
    const int cpt = kNB / dc;
    // rows in the caller's camera order
    std::vector<int> rows(n_cam);
    for (int64_t c = 0; c < n_cam; ++c) rows[c] = (int)c;
    std::sort(rows.begin(), rows.end(), [&](int a, int b) { return cam_ext[a] < cam_ext[b]; });
    // pairs per row: an observation pairs with the observations BEFORE it in its landmark's list
    std::vector<int64_t> rp(n_cam + 1, 0);
    for (int64_t r = 0; r < n_cam; ++r) {
        const int c = rows[r];
        int64_t n = 0;
        for (int e = cam_ptr[c]; e < cam_ptr[c + 1]; ++e) { const int i = cam_obs[e]; n += i - pt_ptr[o_pt[i]]; }
        rp[r + 1] = rp[r] + n;
    }
    const int64_t n_pairs = rp[n_cam];
    std::vector<RawPair> raw((size_t)n_pairs);
    // per row: its pairs sorted by (partner camera, observation) and the number of blocks
    std::vector<int> row_blocks(n_cam, 0);
    parallel_rows(n_cam, [&](int64_t r) {
        const int c = rows[r];
        RawPair* p = raw.data() + rp[r];
        for (int e = cam_ptr[c]; e < cam_ptr[c + 1]; ++e) {
            const int i = cam_obs[e];
            for (int j = pt_ptr[o_pt[i]]; j < i; ++j) *p++ = RawPair{o_cam[j], (uint32_t)i, (uint32_t)j};
        }
        RawPair* b = raw.data() + rp[r];
        std::sort(b, p, [](const RawPair& x, const RawPair& y) { return x.cj != y.cj ? x.cj < y.cj : x.i < y.i; });
        int nb = 0;
        for (RawPair* q = b; q < p; ++q) nb += (q == b || q->cj != q[-1].cj);
        row_blocks[r] = nb;
    });
    // ---- serial pass over the blocks: slot offsets, block table, chunk descriptors, tasks ---------------------------
    out->blocks.clear(); out->chunks.clear(); out->tasks.clear();
    struct Piece { int64_t raw0; int len; int64_t slot0; };   // a block (or a piece of a split block): raw pairs -> slots
    std::vector<Piece> pieces;
    int64_t n_blocks = 0;
    for (int64_t r = 0; r < n_cam; ++r) n_blocks += row_blocks[r];
    pieces.reserve(n_blocks + 16);
    out->blocks.reserve(n_blocks + 16);
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
        int64_t q = rp[r];
        while (q < rp[r + 1]) {
            int64_t e = q;
            const uint32_t cj = raw[q].cj;
            while (e < rp[r + 1] && raw[e].cj == cj) ++e;
            const int I = ci / cpt, J = (int)cj / cpt;
            const int sl = slot[(size_t)I * nt + J];
            const int64_t dst = (int64_t)sl * kNB * kNB + (int64_t)((ci % cpt) * dc) * kNB + ((int)cj % cpt) * dc;
            const uint32_t diag = ((int)cj == ci) ? kPairBlockDiag : 0u;
            int64_t len = e - q;
            const bool split = (len + 1) / 2 * 2 > kPairMaxBlockSlots;
            if (split) close_task();
            while (len > 0) {
                const int take = (int)std::min<int64_t>(len, split ? kPairMaxBlockSlots : len);
                const int padded = (take + 1) / 2 * 2;
                if (!split && slot_pos - task_begin > 0 && slot_pos - task_begin + padded > 2 * kPairTaskSlots) close_task();
                const int bi = (int)out->blocks.size();
                out->blocks.push_back(PairBlock{dst, (uint32_t)ci, cj, diag | ((split || diag) ? kPairBlockAtomic : 0u), 0u});
                pieces.push_back(Piece{q, take, slot_pos});
                chunk_touch(slot_pos, slot_pos + padded, bi, true);
                slot_pos += padded;
                q += take; len -= take;
                if (split || slot_pos - task_begin >= kPairTaskSlots) close_task();
            }
        }
    }
    close_task();
    const int64_t n_slots = slot_pos;
    out->chunks.resize(n_slots / 64, PairChunk{0u, -1});
    // ---- records -----------------------------------------------------------------------------------------------------
    out->recs.assign((size_t)n_slots, PairRec{kPairPad, 0u, 0u, 0u});
    parallel_rows((int64_t)pieces.size(), [&](int64_t b) {
        const Piece& pc = pieces[b];
        for (int k = 0; k < pc.len; ++k) {
            const RawPair& rw = raw[pc.raw0 + k];
            const int64_t s = pc.slot0 + k;
            out->recs[s] = PairRec{rw.i, rw.j, o_pt[rw.i], (uint32_t)((int)b - out->chunks[s / 64].first_block)};
        }
        if (pc.len & 1) {   // the odd block's last K-step: a zero pair that still belongs to the block
            const int64_t s = pc.slot0 + pc.len;
            out->recs[s].blk = (uint32_t)((int)b - out->chunks[s / 64].first_block);
        }
    });
    out->n_pairs = n_pairs;
    out->n_blocks = n_blocks;
}

// ------------------------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------------------------
template <int DC>
__device__ __forceinline__ void pairs_flush(double* __restrict__ tiles, const PairBlock* __restrict__ blocks, int b,
                                            const double4_t acc, int row0, int col) {
    const PairBlock pb = blocks[b];
    double* dst = tiles + pb.dst;
    if (col >= DC) return;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = row0 + 4 * reg;
        if (row >= DC) continue;
        const double val = acc[reg];
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

// The data one lane needs for its pair, fetched one chunk AHEAD (while the previous chunk's sums run on the matrix
// cores): both measurements, Hll^-1 and the point, and -- in the first 2 nblk lanes -- one camera of the chunk's blocks.
struct PairData {
    double2 uvi, uvj;
    double2 lm[6];
    double2 cam[8];
};

__device__ __forceinline__ void pairs_issue_loads(const BAView& v, const PairBlock* __restrict__ blocks,
                                                  const double* __restrict__ lmrec, const PairChunk ck, const uint4 rr, int lane,
                                                  PairData& d) {
    const bool valid = rr.x != kPairPad;
    const uint32_t i = valid ? rr.x : 0u, j = valid ? rr.y : 0u, l = valid ? rr.z : 0u;   // padding lanes read element 0
    d.uvi = v.o_uv[i];
    d.uvj = v.o_uv[j];
    const double2* q = reinterpret_cast<const double2*>(lmrec + kLmStride * (size_t)l);
#pragma unroll
    for (int k = 0; k < 6; ++k) d.lm[k] = q[k];
    const int nblk = 1 + __popc(ck.mask & ~1u);
    const int bsel = min(lane >> 1, nblk - 1);
    const PairBlock* pb = blocks + ck.first_block + bsel;
    const uint32_t cam = (lane & 1) ? pb->cj : pb->ci;
    const double2* src = reinterpret_cast<const double2*>(v.camp + kCamStride * (size_t)cam);
#pragma unroll
    for (int k = 0; k < 8; ++k) d.cam[k] = src[k];
}

template <int DC>
__global__ __launch_bounds__(256) void k_schur_pairs(BAView v, double* __restrict__ tiles, const PairTask* __restrict__ tasks,
                                                       int n_tasks, const PairChunk* __restrict__ chunks,
                                                       const PairBlock* __restrict__ blocks, const PairRec* __restrict__ recs,
                                                       const double* __restrict__ lmrec) {
    constexpr int UV = 2 * DC;                    // doubles of U (and of V) per pair
    constexpr int REG_A = 64 * kPairCamPitch;     // U[64][UV] overlays the staged cameras (64 x 18 doubles >= 64 x UV)
    constexpr int WAVE_LDS = REG_A + 64 * UV;     // | V[64][UV]
    static_assert(64 * UV <= REG_A, "U must fit the camera staging area");
    __shared__ double lds_all[4 * WAVE_LDS];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x * 4 + w;
    if (t >= n_tasks) return;                     // no workgroup barrier anywhere: the four waves are independent
    double* U = lds_all + w * WAVE_LDS;
    double* V = U + REG_A;
    const PairTask task = tasks[t];
    const int r16 = lane & 15, kk = lane >> 4;
    // operand element of this lane for K-step s: U[(2s + (kk >> 1)) * UV + (kk & 1) * DC + r] = U[s * 2 UV + kk DC + r].
    // Rows / columns >= DC of the 16 x 16 product are never stored, so those lanes may read anything (clamped index).
    const int aoff = kk * DC + (r16 < DC ? r16 : DC - 1);
    double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    int cur = -1;

    // Software pipeline over the task's chunks: the gathers of chunk n+1 are issued before the matrix-core phase of
    // chunk n and land while it runs; the 16-byte records run two chunks ahead.
    const int ch_end = task.chunk0 + task.nchunks;
    int ch = task.chunk0;
    PairChunk ck = chunks[ch];
    uint4 rr = reinterpret_cast<const uint4*>(recs)[(size_t)ch * 64 + lane];
    PairData dat;
    pairs_issue_loads(v, blocks, lmrec, ck, rr, lane, dat);
    PairChunk ck_next = ck;
    uint4 rr_next = rr;
    if (ch + 1 < ch_end) { ck_next = chunks[ch + 1]; rr_next = reinterpret_cast<const uint4*>(recs)[(size_t)(ch + 1) * 64 + lane]; }

    for (; ch < ch_end; ++ch) {
        // ---- A: the cameras of the chunk's blocks -> LDS (read back by every lane of the block: broadcast) -------------
        {
            double2* dstc = reinterpret_cast<double2*>(U + lane * kPairCamPitch);
#pragma unroll
            for (int k = 0; k < 8; ++k) dstc[k] = dat.cam[k];
        }
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
                Cam cam;
                const double2* c2 = reinterpret_cast<const double2*>(U + (2 * blk) * kPairCamPitch);
                double cv[16];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 tq = c2[k]; cv[2 * k] = tq.x; cv[2 * k + 1] = tq.y; }
                load_cam_prepared(cv, cam);
                double r[2], Jl[2][3];
                linearize_obs<DC>(cam, pw, dat.uvi.x, dat.uvi.y, v.huber_delta, r, Jci, Jl);
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int b = 0; b < 3; ++b) N[n][b] = Jl[n][0] * Hi[b] + Jl[n][1] * Hi[3 + b] + Jl[n][2] * Hi[6 + b];
            }
            double M[2][2];
            {
                Cam cam;
                const double2* c2 = reinterpret_cast<const double2*>(U + (2 * blk + 1) * kPairCamPitch);
                double cv[16];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 tq = c2[k]; cv[2 * k] = tq.x; cv[2 * k + 1] = tq.y; }
                load_cam_prepared(cv, cam);
                double r[2], Jcj[2][DC], Jl[2][3];
                linearize_obs<DC>(cam, pw, dat.uvj.x, dat.uvj.y, v.huber_delta, r, Jcj, Jl);
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m) M[n][m] = -(N[n][0] * Jl[m][0] + N[n][1] * Jl[m][1] + N[n][2] * Jl[m][2]);
                // V goes to its own LDS region straight away (it never overlaps the staged cameras)
                double2* pv = reinterpret_cast<double2*>(V + lane * UV);
#pragma unroll
                for (int k = 0; k < DC; ++k) {
                    const int e0 = 2 * k, e1 = 2 * k + 1;   // element e of V = Jcj[e / DC][e % DC]
                    const double x0 = valid ? Jcj[e0 / DC][e0 % DC] : 0.0, x1 = valid ? Jcj[e1 / DC][e1 % DC] : 0.0;
                    pv[k] = make_double2(x0, x1);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < DC; ++r) u[m * DC + r] = valid ? Jci[0][r] * M[0][m] + Jci[1][r] * M[1][m] : 0.0;
        }
        // every lane has read its cameras (program order, one wave): U may now overwrite the staging area
        __builtin_amdgcn_wave_barrier();
        {
            double2* pu = reinterpret_cast<double2*>(U + lane * UV);
#pragma unroll
            for (int k = 0; k < UV / 2; ++k) pu[k] = make_double2(u[2 * k], u[2 * k + 1]);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- D: the next chunk's gathers go out now and land during the matrix-core phase ------------------------------
        const PairChunk ck_cur = ck;
        if (ch + 1 < ch_end) {
            ck = ck_next; rr = rr_next;
            pairs_issue_loads(v, blocks, lmrec, ck, rr, lane, dat);
            if (ch + 2 < ch_end) { ck_next = chunks[ch + 2]; rr_next = reinterpret_cast<const uint4*>(recs)[(size_t)(ch + 2) * 64 + lane]; }
        }
        // ---- E: reduce over the lanes: 32 K-steps of two pairs each, four at a time --------------------------------------
        const double* pa = U + aoff;
        const double* pb = V + aoff;
        double an[4], bn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { an[k] = pa[k * 2 * UV]; bn[k] = pb[k * 2 * UV]; }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            double a[4], b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] = an[k]; b[k] = bn[k]; }
            if (g < 7) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { an[k] = pa[((g + 1) * 4 + k) * 2 * UV]; bn[k] = pb[((g + 1) * 4 + k) * 2 * UV]; }
            }
            const uint32_t m4 = (ck_cur.mask >> (4 * g)) & 0xFu;
            if (m4 == 0u) {   // wave-uniform: no block starts inside these four K-steps
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc1, 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if ((m4 >> k) & 1u) {
                        if (cur >= 0) pairs_flush<DC>(tiles, blocks, cur, acc0 + acc1, kk, r16);
                        cur = cur < 0 ? ck_cur.first_block : cur + 1;
                        acc0 = double4_t{0.0, 0.0, 0.0, 0.0}; acc1 = double4_t{0.0, 0.0, 0.0, 0.0};
                    }
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k], b[k], acc0, 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the next chunk's camera staging overwrites U
    }
    if (cur >= 0) pairs_flush<DC>(tiles, blocks, cur, acc0 + acc1, kk, r16);
}

void launch_schur_pairs(int dc, const BAView& v, double* tiles, const PairTask* tasks, int n_tasks, const PairChunk* chunks,
                        const PairBlock* blocks, const PairRec* recs, const double* lmrec, hipStream_t s) {
    if (n_tasks == 0) return;
    const unsigned grid = (unsigned)((n_tasks + 3) / 4);
    if (dc == 9) hipLaunchKernelGGL(k_schur_pairs<9>, dim3(grid), dim3(256), 0, s, v, tiles, tasks, n_tasks, chunks, blocks, recs, lmrec);
    else hipLaunchKernelGGL(k_schur_pairs<6>, dim3(grid), dim3(256), 0, s, v, tiles, tasks, n_tasks, chunks, blocks, recs, lmrec);
}

}  // namespace apex
