// chol_kernels.hip -- tile-sparse fp64 Cholesky of the reduced camera matrix S on gfx950.
//
// S is stored as lower-triangular 144 x 144 tiles (ba_kernels.h).  The factorisation is the
// right-looking tile algorithm; only tiles that are structurally non-zero after symbolic fill
// (computed once on the host, tile granularity) exist, so a banded S costs O(n b^2) and a dense
// one runs the classic dense schedule:
//     for K:   L_KK, L_KK^-1  <- potrf_inv(S_KK)                    k_potrf_inv_mf (1 workgroup)
//              L_IK  <- S_IK L_KK^-T            for I > K           k_tile_gemm   (NT GEMM with L_KK^-1)
//              S_IJ -= L_IK L_JK^T              for I >= J > K      k_tile_gemm   (fp64 MFMA 16x16x4)
// The trailing update is >99 % of the flops and runs on v_mfma_f64_16x16x4_f64.
// Reference: solve_with_cholesky (src/linalg/sparse/explicit_schur.rs:539-634); faer's sparse
// LL^T is not in the reference tree, the algorithm here is the textbook one it implements.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chol_kernels.h"
#include <algorithm>

namespace apex {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int NB = kNB;
// ------------------------------------------------------------------------------------------
// potrf + triangular inverse of one diagonal tile: the latency-critical link of the tile Cholesky (every column K waits for
// it).  One workgroup per tile; the tile lives in LDS as the 45 lower 16 x 16 blocks (pitch 18 doubles -> conflict-free MFMA
// operand reads), 101 KB, plus the 9 inverted diagonal blocks, 20 KB.  Blocked right-looking factorisation with look-ahead:
//   step kb:  P1  wave 0: the 16 x 16 diagonal block D_kb and its inverse  | the other waves: trailing update of step kb-1 for
//                                                                         | the columns > kb; row kb-1 of L^-1 (into registers)
//             P2  panel A(i,kb) <- A(i,kb) D_kb^-T (i > kb);  row kb of L goes to global memory and becomes
//                 L~(kb,k) = D_kb^-1 L(kb,k) in place (k < kb);  look-ahead: wave 0, which solved the panel block (kb+1, kb)
//                 itself, gives the next diagonal block this step's update right away
// with  L^-1(r,j) = - sum_{k=j}^{r-1} L~(r,k) L^-1(k,j),  L^-1(k,k) = D_k^-1  (row-oriented dtrtri): row r of L is dead once
// step r has used it, so its blocks are reused for L~ and then L^-1.  fail[0] is set to K+1 if a pivot is not positive
// (faer's Llt: NonPositivePivot).  (Rounds 1-3 ran this on the vector unit -- k_potrf_inv 85 us per tile, k_potrf_inv_la with
// the look-ahead above 58-65 us -- both deleted in round 6; k_potrf_inv_mf below, round 4, 37 us, is the one that runs.)
// ------------------------------------------------------------------------------------------
constexpr int BS = 16;            // block edge
constexpr int NBK = NB / BS;      // 9 blocks per tile edge
constexpr int BP = 18;            // block row pitch (doubles)
constexpr int BSZ = BS * BP;      // 288 doubles per block
constexpr int NLB = NBK * (NBK + 1) / 2;  // 45 lower blocks

__device__ __forceinline__ int bidx(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }  // bi >= bj

// value of `v` in lane `src` (src must be wave-uniform; here always a compile-time constant)
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// D = X * Y^T (NT) or X * Y (NN) on 16x16 blocks held in LDS with pitch BP
__device__ __forceinline__ double4_t blk_mma_nt(const double* X, const double* Y, double4_t acc, int lr, int lk, double sgn) {
#pragma unroll
    for (int kk = 0; kk < BS; kk += 4) {
        const double a = sgn * X[lr * BP + kk + lk];
        const double b = Y[lr * BP + kk + lk];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ double4_t blk_mma_nn(const double* X, const double* Y, double4_t acc, int lr, int lk) {
#pragma unroll
    for (int kk = 0; kk < BS; kk += 4) {
        const double a = X[lr * BP + kk + lk];
        const double b = Y[(kk + lk) * BP + lr];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ double4_t blk_load_cd(const double* C, int lr, int lk) {
    double4_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = C[(lk + 4 * r) * BP + lr];
    return v;
}
__device__ __forceinline__ void blk_store_cd(double* C, double4_t v, int lr, int lk, double sgn) {
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(lk + 4 * r) * BP + lr] = sgn * v[r];
}

#ifdef APEX_POTRF_TRACE   // tools/potrf_bench.hip: wall-clock stamps of workgroup 0 at the phase boundaries
__device__ unsigned long long g_potrf_trace[64], g_potrf_cycles[64];   // 100 MHz stamps and shader-clock stamps (s_memtime)
__device__ int g_potrf_trace_n;
#define POTRF_STAMP() do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_potrf_cycles[g_potrf_trace_n] = __builtin_amdgcn_s_memtime(); g_potrf_trace[g_potrf_trace_n++] = wall_clock64(); } } while (0)
#else
#define POTRF_STAMP() do {} while (0)
#endif
// ------------------------------------------------------------------------------------------
// potrf + inverse on the matrix pipe (round 4; twelve waves): the blocked schedule above with the serial part
// -- wave 0's 16 x 16 Cholesky and the inverse of its factor -- rewritten for the matrix pipe.  The look-ahead kernel
// spent 4.1 of its 5.4 us per block step there: lane r owned row r, so every one of the 16 pivots broadcast its column
// through ~30 v_readlane (SGPR round trips) for the rank-1 update, and the inverse was a second pass of 16 steps.
// Here the diagonal block lives in wave 0 as ONE MFMA accumulator (16 x 16 fp64 = 4 doubles per lane, element
// [lk + 4 i][lr] in register i of lane (lr, lk)), kept SYMMETRIC, and a pivot step is
//     d = D[j][j]                              one v_readlane pair (lane and register known at compile time)
//     1/sqrt(d), sqrt(d)                       rsq + two Newton steps, wave-uniform
//     row j of D scaled = column j of L        register j/4 of the lanes with lk == j%4 -- already where the MFMA wants
//                                              operand column k = j%4 of A (rows) AND row k of B (columns): no lane moves
//     D -= l l^T                               one v_mfma_f64_16x16x4 with the other three k-slices zero
//     X -= l (row j of X)                      a second one: X = L^-1 by forward substitution rides along (X starts as I),
// i.e. two matrix instructions and ~10 vector instructions per pivot instead of ~60; rows <= j see a zero operand and
// keep their values, so D ends as L^T in its upper triangle.  Wave 0 then solves the panel block below the diagonal
// block itself, TRANSPOSED (X N^T = L_(kb+1,kb)^T, whose accumulator registers are exactly the operand registers of
// L L^T), and gives the next diagonal block this step's update in registers: between two block factorisations wave 0
// touches LDS only to fetch N and the next D and to publish L / X for the other waves.
// ------------------------------------------------------------------------------------------
// Coherent tile accesses for the dataflow kernels: data produced by another workgroup of the SAME launch may sit in another
// XCD's L2, so it is read and written at agent scope -- the sc1 bit, exactly what a relaxed agent-scope atomic compiles to
// (flow_ld / flow_st below) -- but 16 bytes at a time, which the atomics cannot do: raw buffer loads / stores through a
// descriptor of the tile (bounds = the tile, so a stray offset reads zero instead of faulting), cache policy sc1.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x2_t __attribute__((ext_vector_type(2)));
constexpr int kCohAux = 16;   // gfx940+: bit 4 of the buffer instructions' cache-policy operand = sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const void* tile) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(tile), 0, NB * NB * (int)sizeof(double), 0x00020000);
}
template <bool COH>
__device__ __forceinline__ double2 tile_ld2(const double* tile, __amdgpu_buffer_rsrc_t r, int off) {   // off: in doubles, even
    if constexpr (COH) {
        const i32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(r, off * 8, 0, kCohAux);
        double2 v; v.x = __hiloint2double(q[1], q[0]); v.y = __hiloint2double(q[3], q[2]);
        return v;
    } else {
        return *reinterpret_cast<const double2*>(tile + off);
    }
}
template <bool COH>
__device__ __forceinline__ void tile_st2(double* tile, __amdgpu_buffer_rsrc_t r, int off, double2 v) {
    if constexpr (COH) {
        i32x4_t q;
        q[0] = __double2loint(v.x); q[1] = __double2hiint(v.x); q[2] = __double2loint(v.y); q[3] = __double2hiint(v.y);
        __builtin_amdgcn_raw_buffer_store_b128(q, r, off * 8, 0, kCohAux);
    } else {
        *reinterpret_cast<double2*>(tile + off) = v;
    }
}
__device__ __forceinline__ double coh_ld1(__amdgpu_buffer_rsrc_t r, int off) {
    const i32x2_t q = __builtin_amdgcn_raw_buffer_load_b64(r, off * 8, 0, kCohAux);
    return __hiloint2double(q[1], q[0]);
}
__device__ __forceinline__ void coh_st1(__amdgpu_buffer_rsrc_t r, int off, double v) {
    i32x2_t q; q[0] = __double2loint(v); q[1] = __double2hiint(v);
    __builtin_amdgcn_raw_buffer_store_b64(q, r, off * 8, 0, kCohAux);
}
template <bool COH>
__device__ __forceinline__ void blk_to_tile(const double* __restrict__ src, double* __restrict__ dst_tile, __amdgpu_buffer_rsrc_t r,
                                            int bi, int bj, int t, int nthreads) {
    for (int idx = t; idx < 128; idx += nthreads) {   // 16 rows x 8 double2 per block
        const int rr = idx >> 3, c2 = idx & 7;
        double2 v; v.x = src[rr * BP + 2 * c2]; v.y = src[rr * BP + 2 * c2 + 1];
        tile_st2<COH>(dst_tile, r, (16 * bi + rr) * NB + 16 * bj + 2 * c2, v);
    }
}

__device__ __forceinline__ double4_t blk_load_cd_sym(const double* C, int lr, int lk) {   // lower triangle mirrored
    double4_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = lk + 4 * r;
        v[r] = row >= lr ? C[row * BP + lr] : C[lr * BP + row];
    }
    return v;
}

template <int NW, bool COH>
__device__ __forceinline__ void potrf_tile_mf(double* __restrict__ A, double* __restrict__ Linv, int K, int* __restrict__ fail,
                                              double* sA, double* sD, int* bad, int* sync_cnt) {
    constexpr int NT = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    // Wave 0's pivot loop lives on ONE SIMD's matrix pipe: two dependent v_mfma_f64 per pivot, 161 cycles per pivot when it has
    // the pipe to itself (tools/lat_bench.hip) -- and twice that when another wave of the workgroup feeds the same pipe with
    // block products (measured: 360).  A workgroup's waves go to the four SIMDs cyclically, so waves 4 (and 8) share wave 0's:
    // they sit out the block products; the helpers are the waves with w % 4 != 0.
    constexpr int NH = NW - (NW + 3) / 4;
    const bool helper = (w & 3) != 0;
    const int hid = w - 1 - (w >> 2);   // 1,2,3,5,6,7,(9,10,11) -> 0..5,(6,7,8)
    const __amdgpu_buffer_rsrc_t rA = coh_rsrc(A), rL = coh_rsrc(Linv);
    if (tid == 0) { *bad = 0; *sync_cnt = 0; }
    {
        constexpr int NREG = (128 * 45 + NT - 1) / NT + NBK;
        double2 reg[NREG];
        int n = 0;
#pragma unroll
        for (int bi = 0; bi < NBK; ++bi) {
            const int per_row = 8 * (bi + 1), cnt = 16 * per_row;
#pragma unroll
            for (int it = 0; it < (16 * 8 * (bi + 1) + NT - 1) / NT; ++it, ++n) {
                const int idx = tid + NT * it;
                if (idx < cnt) {
                    const int rr = idx / per_row, c2 = idx - rr * per_row;
                    reg[n] = tile_ld2<COH>(A, rA, (16 * bi + rr) * NB + 2 * c2);
                }
            }
        }
        n = 0;
#pragma unroll
        for (int bi = 0; bi < NBK; ++bi) {
            const int per_row = 8 * (bi + 1), cnt = 16 * per_row;
#pragma unroll
            for (int it = 0; it < (16 * 8 * (bi + 1) + NT - 1) / NT; ++it, ++n) {
                const int idx = tid + NT * it;
                if (idx < cnt) {
                    const int rr = idx / per_row, c2 = idx - rr * per_row;
                    double* dst = sA + bidx(bi, c2 >> 3) * BSZ + rr * BP + 2 * (c2 & 7);
                    dst[0] = reg[n].x; dst[1] = reg[n].y;
                }
            }
        }
    }
    __syncthreads();
    POTRF_STAMP();

    double4_t Dm = (double4_t){0.0, 0.0, 0.0, 0.0};   // wave 0: the current diagonal block, symmetric, carried across the steps
    if (w == 0) Dm = blk_load_cd_sym(sA, lr, lk);
    for (int kb = 0; kb < NBK; ++kb) {
        // ---------------- P1 ------------------------------------------------------------------------------
        double4_t T[(NBK - 1 + NH - 1) / NH];   // row kb-1 of L^-1: blocks j = hid, hid + NH, ... of the helpers
        if (w == 0) {
            double4_t Xm, Ls = (double4_t){0.0, 0.0, 0.0, 0.0}, Xs = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) Xm[r] = (lk + 4 * r == lr) ? 1.0 : 0.0;
            // Nothing protects the rows and columns <= j from the later updates (a masked operand costs two v_cndmask per
            // double, and the 16 steps are bound by their instruction count): row j of L^T and of X are SAVED the moment they
            // are final -- the scaled row is non-zero on the lanes of row j only, so "save" is one add -- and what the
            // updates then do to the dead rows / columns of Dm and Xm never reaches a live element (an update of element
            // (r, c) uses column entries r and c of the pivot row only, and for r, c > j those are clean).
            // TWO pivots per matrix instruction: fp64 MFMA and fp64 VALU share one datapath (an MFMA holds it for 64 cycles,
            // tools/lat_bench.hip), so the loop costs the SUM of its instructions and the MFMAs are the largest item.  Pivots
            // j (even) and j + 1 sit in adjacent 16-lane rows of the same accumulator register: row j, scaled, is copied next
            // door (v_permlane16_swap), row j + 1 takes pivot j's update there as one FMA and is scaled in turn, and the two
            // scaled rows -- disjoint lanes of one operand register -- go through ONE rank-2 update.  The inverse likewise.
            double rmask[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) rmask[q] = lk == q ? 1.0 : 0.0;
            auto rsq_newton = [](double d) {
                double r = __builtin_amdgcn_rsq(d);
                r = r * fma(-0.5 * d * r, r, 1.5);
                r = r * fma(-0.5 * d * r, r, 1.5);
                return r;
            };
            auto even_rows_to_odd = [](double x) {   // 16-lane rows 1, 3 <- rows 0, 2 (rows 0, 2 keep their values)
                const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
                return __hiloint2double(hi[0], lo[0]);
            };
            int isbad = 0;
            double dnext = readlane_f64(Dm[0], 0);
#pragma unroll
            for (int j = 0; j < BS; j += 2) {
                const int jr = j >> 2, q0 = j & 3, q1 = q0 + 1;
                const double d0 = dnext;
                if (!(d0 > 0.0)) isbad = 1;      // (a non-positive pivot poisons what follows with NaN; the tile is reported failed)
                const double s0 = rsq_newton(d0) * rmask[q0];
                const double l0 = Dm[jr] * s0;                      // lanes of row j: L[lr][j] (lr == j: sqrt(d) = d / sqrt(d)); 0 elsewhere
                const double x0 = Xm[jr] * s0;                      // lanes of row j: X[j][lr] / L[j][j], final
                const double u = readlane_f64(l0, 16 * q0 + j + 1);               // L[j+1][j]
                const double r1 = fma(-u, even_rows_to_odd(l0), Dm[jr]);          // lanes of row j+1: that row after pivot j's update
                const double d1 = readlane_f64(r1, 16 * q1 + j + 1);
                if (!(d1 > 0.0)) isbad = 1;
                const double s1 = rsq_newton(d1) * rmask[q1];
                const double l1 = r1 * s1;                                         // lanes of row j+1: L[lr][j+1]
                const double x1 = fma(-u, even_rows_to_odd(x0), Xm[jr]) * s1;      // lanes of row j+1: X[j+1][lr], final
                const double la = l0 + l1, xb = x0 + x1;
                Ls[jr] += la;
                Xs[jr] += xb;
                if (j + 2 < BS) {
                    // the NEXT pivot ahead of the update that produces it, so that its 1/sqrt chain does not wait for the matrix pipe
                    const double dd = readlane_f64(Dm[(j + 2) >> 2], 16 * ((j + 2) & 3) + j + 2);
                    const double a0 = readlane_f64(l0, 16 * q0 + j + 2), a1 = readlane_f64(l1, 16 * q1 + j + 2);
                    dnext = fma(-a1, a1, fma(-a0, a0, dd));
                    Dm = __builtin_amdgcn_mfma_f64_16x16x4f64(-la, la, Dm, 0, 0, 0);
                    Xm = __builtin_amdgcn_mfma_f64_16x16x4f64(-la, xb, Xm, 0, 0, 0);
                }
            }
            // Ls holds L^T on and above its diagonal: register i of lane (lr, lk) is L[lr][lk + 4 i]; Xs holds X = L^-1
            double* D = sA + bidx(kb, kb) * BSZ;
            double* Xd = sD + kb * BSZ;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = lk + 4 * r;
                D[lr * BP + c] = lr >= c ? Ls[r] : 0.0;
                Xd[c * BP + lr] = Xs[r];
            }
            if (isbad) *bad = 1;
            POTRF_STAMP();
        } else if (helper) {
            if (kb >= 1) {
                const int ks = kb - 1;
                // column kb below its diagonal block first (wave 0 reads block (kb+1, kb) right behind the barrier) ...
                for (int i = kb + 1 + hid; i < NBK; i += NH) {
                    double* Cb = sA + bidx(i, kb) * BSZ;
                    double4_t acc = blk_load_cd(Cb, lr, lk);
                    acc = blk_mma_nt(sA + bidx(i, ks) * BSZ, sA + bidx(kb, ks) * BSZ, acc, lr, lk, -1.0);
                    blk_store_cd(Cb, acc, lr, lk, 1.0);
                }
                // ... then the rest of the trailing update of step ks: targets (i, j) with kb < j <= i
                const int m = NBK - 1 - kb;
                const int n_upd = m * (m + 1) / 2;
                for (int t = hid; t < n_upd; t += NH) {
                    int ii = 0;
                    while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
                    const int jj = t - ii * (ii + 1) / 2;
                    const int i = kb + 1 + ii, j = kb + 1 + jj;
                    double* Cb = sA + bidx(i, j) * BSZ;
                    double4_t acc = blk_load_cd(Cb, lr, lk);
                    acc = blk_mma_nt(sA + bidx(i, ks) * BSZ, sA + bidx(j, ks) * BSZ, acc, lr, lk, -1.0);
                    blk_store_cd(Cb, acc, lr, lk, 1.0);
                }
                // row r = ks of L^-1 from L~(r,.) and the rows above (kept in registers until the barrier)
                const int r = ks;
                int nt = 0;
                for (int j = hid; j < r; j += NH, ++nt) {
                    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
                    for (int k = j; k < r; ++k) {
                        const double* Y = (k == j) ? (sD + j * BSZ) : (sA + bidx(k, j) * BSZ);
                        acc = blk_mma_nn(sA + bidx(r, k) * BSZ, Y, acc, lr, lk);
                    }
                    T[nt] = acc;
                }
            }
        }
        __syncthreads();
        POTRF_STAMP();
        if (helper && kb >= 1) {
            const int r = kb - 1;
            int nt = 0;
            for (int j = hid; j < r; j += NH, ++nt) blk_store_cd(sA + bidx(r, j) * BSZ, T[nt], lr, lk, -1.0);
        }
        // ---------------- P2 ------------------------------------------------------------------------------
        // tasks 0 .. m-1: panel blocks (kb+1+t, kb); tasks m .. m+kb-1: row block (kb, t-m) -> global, then L~.
        // Wave 0 takes the panel block right below the diagonal (task 0) and nothing else; the other waves share the rest.
        {
            const int m = NBK - 1 - kb;
            if (w == 0) {
                if (m > 0) {
                    // L_(kb+1,kb)^T = X N^T, so that register i of lane (lr, lk) is L[lr][lk + 4 i]: the operand of L L^T
                    double* N = sA + bidx(kb + 1, kb) * BSZ;
                    double4_t Lt = (double4_t){0.0, 0.0, 0.0, 0.0};
                    Lt = blk_mma_nt(sD + kb * BSZ, N, Lt, lr, lk, 1.0);
                    Dm = blk_load_cd_sym(sA + bidx(kb + 1, kb + 1) * BSZ, lr, lk);   // through step kb-1: the other waves' P1
#pragma unroll
                    for (int r = 0; r < 4; ++r) N[lr * BP + lk + 4 * r] = Lt[r];
#pragma unroll
                    for (int r = 0; r < 4; ++r) Dm = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lt[r], Lt[r], Dm, 0, 0, 0);
                }
            } else if (helper) {
                const int t0 = m > 0 ? 1 : 0;
                for (int t = t0 + hid; t < m + kb; t += NH) {
                    if (t < m) {
                        double* P = sA + bidx(kb + 1 + t, kb) * BSZ;
                        double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
                        acc = blk_mma_nt(P, sD + kb * BSZ, acc, lr, lk, 1.0);
                        blk_store_cd(P, acc, lr, lk, 1.0);
                    } else {
                        const int k = t - m;
                        double* Bk = sA + bidx(kb, k) * BSZ;
                        blk_to_tile<COH>(Bk, A, rA, kb, k, lane, 64);
                        double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
                        acc = blk_mma_nn(sD + kb * BSZ, Bk, acc, lr, lk);
                        blk_store_cd(Bk, acc, lr, lk, 1.0);  // same wave read it: LDS accesses of one wave stay in order
                    }
                }
                if (hid == NH - 1) blk_to_tile<COH>(sA + bidx(kb, kb) * BSZ, A, rA, kb, kb, lane, 64);  // the diagonal block of L
            }
        }
        // P2 -> P1 WITHOUT a workgroup barrier: wave 0 needs nothing from the other waves' P2 -- its next block is in its
        // registers -- and goes straight to the next 16 pivots; the helpers' next job (the trailing update of this step) needs
        // the whole panel, theirs and wave 0's block: everyone who wrote a piece counts itself in, the helpers wait for the
        // count (an LDS counter: the LDS executes a wave's operations in order, the release / acquire fences are waitcnts).
        // One barrier per block step is left, behind P1, where wave 0 picks up the helpers' updates of its next block.
        if (w == 0 || helper) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(sync_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (helper) {
            const int target = (kb + 1) * (NH + 1);
            while (__hip_atomic_load(sync_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        POTRF_STAMP();
    }
    __syncthreads();   // (the last step's row blocks are in: the last row of L^-1 reads them)
    // last row of L^-1 (r = NBK-1), all waves
    {
        const int r = NBK - 1;
        double4_t T[(NBK - 1 + NW - 1) / NW];
        int nt = 0;
        for (int j = w; j < r; j += NW, ++nt) {
            double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < r; ++k) {
                const double* Y = (k == j) ? (sD + j * BSZ) : (sA + bidx(k, j) * BSZ);
                acc = blk_mma_nn(sA + bidx(r, k) * BSZ, Y, acc, lr, lk);
            }
            T[nt] = acc;
        }
        __syncthreads();
        nt = 0;
        for (int j = w; j < r; j += NW, ++nt) blk_store_cd(sA + bidx(r, j) * BSZ, T[nt], lr, lk, -1.0);
        __syncthreads();
    }
#pragma unroll
    for (int bi = 0; bi < NBK; ++bi) {
        const int per_row = 8 * (bi + 1), cnt = 16 * per_row;
        for (int idx = tid; idx < cnt; idx += NT) {
            const int rr = idx / per_row, c2 = idx - rr * per_row;
            const int bj = c2 >> 3;
            const double* src = ((bj == bi) ? (sD + bi * BSZ) : (sA + bidx(bi, bj) * BSZ)) + rr * BP + 2 * (c2 & 7);
            double2 v; v.x = src[0]; v.y = src[1];
            tile_st2<COH>(Linv, rL, (16 * bi + rr) * NB + 2 * c2, v);
        }
    }
    if (tid == 0 && *bad) atomicCAS(fail, 0, K + 1);
    POTRF_STAMP();
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_potrf_inv_mf(const PotrfTask* __restrict__ tasks, int* __restrict__ fail,
                                                          int* __restrict__ arrived) {
    if (arrived != nullptr && threadIdx.x == 0) __hip_atomic_fetch_add(arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    POTRF_STAMP();
    __shared__ double sA[NLB * BSZ];
    __shared__ double sD[NBK * BSZ];
    __shared__ int bad, sync_cnt;
    const PotrfTask pt = tasks[blockIdx.x];
    potrf_tile_mf<NW, false>(pt.A, pt.Linv, pt.K, fail, sA, sD, &bad, &sync_cnt);
}
#undef POTRF_STAMP

// ------------------------------------------------------------------------------------------
// Batched 144^3 tile GEMM, NT form:  C = beta*C + alpha * A * B^T   (all row-major tiles), on v_mfma_f64_16x16x4_f64: lane l
// supplies A[i=l&15][k=l>>4] and B^T[k=l>>4][j=l&15] = B[j][k]; result reg r of lane l is C[(l>>4)+4r][l&15].
// K is consumed in 16-wide chunks staged through LDS with an 18-double row pitch: the 16 rows x 2 k-values read by each
// half-wave of a ds_read_b64 then hit 32 distinct bank pairs.
// C may alias A (L_IK = S_IK L_KK^-T in place): every global read of A is staged into LDS before the last barrier of the K loop
// and the epilogue stores come after it.
// ------------------------------------------------------------------------------------------
constexpr int KC = 16;  // 24 (6 chunks) measured the same: the chunk size is not the limiter
constexpr int STRIP = 48;          // output rows per workgroup
constexpr int NSTRIP = NB / STRIP; // 3 workgroups per tile: 3x the parallelism of one per tile
constexpr int kGemmSmallMax = 56;  // batches of at most this many tasks use the 9-workgroups-per-task latency kernel

typedef double __attribute__((address_space(1)))* GlobalF64;
typedef const double __attribute__((address_space(1)))* GlobalCF64;
typedef double f64x2_t __attribute__((ext_vector_type(2)));
typedef const f64x2_t __attribute__((address_space(1)))* GlobalCF64x2;

// The large-batch kernel (round 6).  Unit = a 48-row strip of C (all 144 columns) on FOUR waves -- one per SIMD of the CU whatever
// the dispatcher does, so k resident workgroups are exactly k waves on every SIMD.  (Rounds 1-5 ran the strip on three waves of
// 48 x 48: three-wave workgroups spread evenly over the four SIMDs only on average -- tools/wave_placement_bench.hip, 67.5
// against 76.5 TF/s for a bare MFMA loop -- and each chunk's barrier makes the workgroup wait for its slowest wave: counters
// 0.71 -> 0.75 of the matrix pipe, 55.9 -> 59.0 TF/s cache-resident, profiles/r06_gemm_forms*.txt.)  The 27 blocks of a strip
// are dealt 7 / 7 / 7 / 6: wave w owns the block columns 2w, 2w + 1 (3 x 2 blocks) and, w < 3, block (w, 8).
// A chunk of A (48 rows) and B (144 rows) is one 192-row image in LDS, 1,536 double2 = six per thread; 122 VGPRs = four
// workgroups per CU.  Every element of C sums its k chunk by chunk, four k per instruction, the lane's k = kk + lane / 16 --
// the order of the small-batch kernels and the dataflow units below: same bits in every schedule.
// Tried beside it, same bits, all slower (profiles/r06_gemm_forms.txt): two LDS images with ONE barrier per chunk (52 KB = three
// workgroups per CU: 55.8), persistent workgroups that request the next unit's first chunk before their epilogue (168 VGPRs =
// three per CU: 54.0; two per CU 49.5) -- what pays in this kernel is resident waves per SIMD, not fewer barriers or prologues.
// TRI (the panel solves): B is a lower-triangular inverse whose 16 x 16 blocks right of the diagonal are never written (zero),
// so column block cb needs the K chunks 0 .. cb only -- 45 of the 81 block products; the columns are dealt by weight,
// {8, 1} {7, 2} {6, 3} {5, 4, 0} = 11 / 11 / 11 / 12 chunk products per block row.  Skipping a product with an exact-zero factor
// changes no finite value.
// ------------------------------------------------------------------------------------------
template <bool TRI>
__global__ __launch_bounds__(256, 3) void k_tile_gemm_nt(const GemmTask* __restrict__ tasks, int n_units, double alpha, double beta) {
    constexpr int P4 = KC + 2;
    constexpr int IMG = (NB + STRIP) * P4;
    __shared__ double sm[IMG];
    // XCD-aware unit order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2.  The unit
    // lists are sorted by source column, so giving every XCD one CONTIGUOUS eighth of the list makes the 3 strips of a task and
    // the tasks of one column share their operand tiles in one L2 instead of fetching them 8 times from HBM.  (Placement is a
    // speed assumption only: any mapping computes the same result.)
    const int per_xcd = (n_units + 7) >> 3;
    const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= n_units) return;
    GemmTask tg = tasks[unit / NSTRIP];
    // Bit 0 of C (round 5, TilePlan::build): this task is the FIRST writer of a fill tile -- the tile holds nothing yet, it is
    // not cleared before the factorisation and not read here (beta = 0; "+ 0.0" keeps the bits of the sum with a cleared tile)
    const bool first = (reinterpret_cast<uintptr_t>(tg.C) & 1) != 0;
    tg.C = reinterpret_cast<double*>(reinterpret_cast<uintptr_t>(tg.C) & ~uintptr_t(7));
    const int strip = unit % NSTRIP;
    // The task's pointers are loaded from memory, so the compiler only knows them as generic (flat) addresses; flat loads count
    // on lgkmcnt as well as vmcnt, which makes every wait for an LDS read also wait for the global prefetch of the next chunk.
    // Re-typed as global (address space 1) they become global_load / global_store and the prefetch stays asynchronous.
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    GlobalCF64 Ag = (GlobalCF64)tg.A + (size_t)strip * STRIP * NB;
    GlobalCF64 Bg = (GlobalCF64)tg.B;
    constexpr int NACC = TRI ? 9 : 7;
    int cb[3];
    if (TRI) { cb[0] = 16 * (8 - w); cb[1] = 16 * (1 + w); cb[2] = w == 3 ? 0 : -16; }   // -16: no third column (every chunk skipped)
    else { cb[0] = 32 * w; cb[1] = 32 * w + 16; cb[2] = 128; }
    const bool extra = TRI ? (w == 3) : (w < 3);   // wave-uniform: the third column block (TRI) / the ninth column's block of row w
    double4_t acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    constexpr int C2 = KC / 2, NR = (NB + STRIP) * C2 / 256, RPR = 256 / C2;   // six rounds of 32 rows x 8 double2
    static_assert((NB + STRIP) * C2 % 256 == 0 && 256 % C2 == 0 && NR == 6 && RPR == 32, "staging loop assumes whole rounds");
    // the staged rows of this thread: r0 + 32 i of the 192-row image = rows 0..143 of B, then the 48 rows of the A strip: rounds
    // 0..3 are B, round 5 is A, round 4 is B for r0 < 16 and A behind it.  Uniform base + 32-bit offset: global_load with saddr
    const int r0 = tid / C2, c2 = tid % C2;
    const unsigned offB = r0 * NB + 2 * c2, offA = (r0 + 16) * NB + 2 * c2;
    GlobalCF64 p4 = r0 < 16 ? Bg + (size_t)(r0 + 128) * NB + 2 * c2 : Ag + (size_t)(r0 - 16) * NB + 2 * c2;
    const int dst0 = r0 * P4 + 2 * c2;
    f64x2_t rg[NR];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rg[i] = *reinterpret_cast<GlobalCF64x2>(Bg + (offB + (unsigned)(RPR * i * NB + k0)));
        rg[4] = *reinterpret_cast<GlobalCF64x2>(p4 + k0);
        rg[5] = *reinterpret_cast<GlobalCF64x2>(Ag + (offA + (unsigned)k0));
    };
    auto stage = [&](double* img) {
#pragma unroll
        for (int i = 0; i < NR; ++i) { img[dst0 + RPR * i * P4] = rg[i].x; img[dst0 + RPR * i * P4 + 1] = rg[i].y; }
    };
    auto compute = [&](const double* img, int k0) {
        const double* sB = img;
        const double* sA = img + NB * P4;
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            if (TRI) {
                if (k0 > cb[0]) continue;   // (wave-uniform; cb[0] is the wave's largest column)
                double av[3], bv[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) av[i] = sA[(16 * i + lr) * P4 + kk + lk];
#pragma unroll
                for (int j = 0; j < 3; ++j) bv[j] = sB[((cb[j] < 0 ? 0 : cb[j]) + lr) * P4 + kk + lk];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (k0 > cb[j]) continue;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        acc[3 * i + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[3 * i + j], 0, 0, 0);
                }
            } else {
                double av[3], bv[2];
#pragma unroll
                for (int i = 0; i < 3; ++i) av[i] = sA[(16 * i + lr) * P4 + kk + lk];
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[j] = sB[(cb[j] + lr) * P4 + kk + lk];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[2 * i + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[2 * i + j], 0, 0, 0);
                if (extra) {
                    const double ax = sA[(16 * w + lr) * P4 + kk + lk], bx = sB[(128 + lr) * P4 + kk + lk];
                    acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bx, acc[6], 0, 0, 0);
                }
            }
        }
    };
    gload(0);
    for (int k0 = 0; k0 < NB; k0 += KC) {
        __syncthreads();  // previous chunk fully consumed
        stage(sm);
        __syncthreads();
        if (k0 + KC < NB) gload(k0 + KC);  // next chunk in flight while this one feeds the MFMAs
        compute(sm, k0);
    }
    // epilogue: block n of this wave is at (brow(n), bcol(n)); read-modify-write software-pipelined, the loads of block n + 2 in
    // flight while block n is stored (a load-modify-store per element serialises the round trips: tools/gemm_var.hip)
    GlobalF64 C = (GlobalF64)tg.C + (size_t)strip * STRIP * NB;
    auto brow = [&](int n) { return TRI ? 16 * (n / 3) : (n < 6 ? 16 * (n / 2) : 16 * w); };
    auto bcol = [&](int n) { return TRI ? cb[n % 3] : (n < 6 ? cb[n % 2] : 128); };
    if (beta == 0.0 || first) {
        const double zero = first ? 0.0 : -0.0;   // (x + -0.0 == x bit for bit: the panel solves' stores are unchanged)
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            if (TRI ? (n % 3 == 2 && !extra) : (n == 6 && !extra)) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)(brow(n) + lk + 4 * r) * NB + bcol(n) + lr] = alpha * acc[n][r] + zero;
        }
        return;
    }
    if (TRI) {   // (the panel solves run with beta == 0; kept general)
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            if (n % 3 == 2 && !extra) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                GlobalF64 p = C + (size_t)(brow(n) + lk + 4 * r) * NB + bcol(n) + lr;
                *p = alpha * acc[n][r] + beta * *p;
            }
        }
        return;
    }
    double cv[3][4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[n][r] = C[(size_t)(brow(n) + lk + 4 * r) * NB + bcol(n) + lr];
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
        if (n + 2 < NACC && (n + 2 < 6 || extra)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[(n + 2) % 3][r] = C[(size_t)(brow(n + 2) + lk + 4 * r) * NB + bcol(n + 2) + lr];
        }
        if (n == 6 && !extra) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            C[(size_t)(brow(n) + lk + 4 * r) * NB + bcol(n) + lr] = alpha * acc[n][r] + beta * cv[n % 3][r];
    }
}

// ------------------------------------------------------------------------------------------
// Small batches (the upper levels of the elimination tree, pose graphs): latency matters, not throughput.
// NINE workgroups per task, one per 48 x 48 block of C, three waves each (wave w: rows 16w.., three 16-wide
// column blocks -> 3 accumulators, 108 MFMAs instead of 324 per wave), K in three 48-wide chunks so that only
// three dependent global-load latencies remain instead of nine.  A lone tile product takes ~8 us instead of
// ~18 us; it moves 1 MB through L2 per task instead of 0.66 MB, which is irrelevant at these batch sizes.
// ------------------------------------------------------------------------------------------
constexpr int KS = 48;            // K chunk of the small-batch kernel
constexpr int PS = KS + 2;        // LDS pitch: 50 doubles = 100 dwords, rows shift by 36 banks -> conflict-free b64 reads

__global__ __launch_bounds__(192) void k_tile_gemm_nt_small(const GemmTask* __restrict__ tasks, int n_units, double alpha,
                                                              double beta) {
    __shared__ double sA[48 * PS];
    __shared__ double sB[48 * PS];
    const int unit = blockIdx.x;
    if (unit >= n_units) return;
    GemmTask tg = tasks[unit / 9];
    const bool first = (reinterpret_cast<uintptr_t>(tg.C) & 1) != 0;   // first writer of a fill tile (k_tile_gemm_nt)
    tg.C = reinterpret_cast<double*>(reinterpret_cast<uintptr_t>(tg.C) & ~uintptr_t(7));
    if (first) beta = 0.0;
    const int blk = unit % 9, bi = blk / 3, bj = blk % 3;
    GlobalCF64 Ag = (GlobalCF64)tg.A + (size_t)bi * 48 * NB;
    GlobalCF64 Bg = (GlobalCF64)tg.B + (size_t)bj * 48 * NB;
    GlobalF64 C = (GlobalF64)tg.C + (size_t)bi * 48 * NB + bj * 48;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    double4_t acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    constexpr int C2 = KS / 2, NR = 48 * C2 / 192;  // 1152 double2 per operand chunk: 6 per thread
    static_assert(48 * C2 % 192 == 0 && NB % KS == 0, "staging loops assume whole rounds");
    f64x2_t ra[NR], rb[NR];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int idx = tid + 192 * i, row = idx / C2, c2 = idx % C2;
            ra[i] = *reinterpret_cast<GlobalCF64x2>(Ag + (size_t)row * NB + k0 + 2 * c2);
            rb[i] = *reinterpret_cast<GlobalCF64x2>(Bg + (size_t)row * NB + k0 + 2 * c2);
        }
    };
    // the C block for the read-modify-write: requested first, consumed last
    double cv[3][4];
    if (beta != 0.0) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[j][r] = C[(size_t)(16 * w + lk + 4 * r) * NB + 16 * j + lr];
    }
    gload(0);
    for (int k0 = 0; k0 < NB; k0 += KS) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int idx = tid + 192 * i, row = idx / C2, c2 = idx % C2;
            sA[row * PS + 2 * c2] = ra[i].x; sA[row * PS + 2 * c2 + 1] = ra[i].y;
            sB[row * PS + 2 * c2] = rb[i].x; sB[row * PS + 2 * c2 + 1] = rb[i].y;
        }
        __syncthreads();
        if (k0 + KS < NB) gload(k0 + KS);
#pragma unroll
        for (int kk = 0; kk < KS; kk += 4) {
            const double a = sA[(16 * w + lr) * PS + kk + lk];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double b = sB[(16 * j + lr) * PS + kk + lk];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            }
        }
    }
    // C may alias A (in-place panel solve, beta == 0): a workgroup of block row bi reads A rows [48 bi, +48) over ALL of K
    // while the workgroups (bi, 0..2) write exactly those rows of C.  The launcher therefore never uses this kernel
    // for aliased tasks (launch_tile_gemm_nt: alpha == 1 && beta == 0 is the panel solve).
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = alpha * acc[j][r];
            C[(size_t)(16 * w + lk + 4 * r) * NB + 16 * j + lr] = (beta != 0.0) ? v + beta * cv[j][r] : (first ? v + 0.0 : v);
        }
}

// The in-place panel solves (C aliases A) of small batches: THREE workgroups per task, one per 48-row strip (a
// workgroup then reads only the rows it writes, all of them staged before its last barrier), NINE waves each
// (wave w: row block w / 3, column blocks 3 (w % 3) .. +2 -> 3 accumulators), the same 48-wide K chunks.
__global__ __launch_bounds__(576) void k_tile_gemm_nt_small_strip(const GemmTask* __restrict__ tasks, int n_units, double alpha,
                                                                    double beta) {
    __shared__ double sA[48 * PS];
    __shared__ double sB[NB * PS];
    const int unit = blockIdx.x;
    if (unit >= n_units) return;
    const GemmTask tg = tasks[unit / 3];
    const int bi = unit % 3;
    GlobalCF64 Ag = (GlobalCF64)tg.A + (size_t)bi * 48 * NB;
    GlobalCF64 Bg = (GlobalCF64)tg.B;
    GlobalF64 C = (GlobalF64)tg.C + (size_t)bi * 48 * NB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int wr = w / 3, wc = w % 3;
    double4_t acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    constexpr int C2 = KS / 2, NRA = 48 * C2 / 576, NRB = NB * C2 / 576;  // 2 and 6 double2 per thread
    static_assert(48 * C2 % 576 == 0 && NB * C2 % 576 == 0, "staging loops assume whole rounds");
    f64x2_t ra[NRA], rb[NRB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int idx = tid + 576 * i, row = idx / C2, c2 = idx % C2;
            ra[i] = *reinterpret_cast<GlobalCF64x2>(Ag + (size_t)row * NB + k0 + 2 * c2);
        }
#pragma unroll
        for (int i = 0; i < NRB; ++i) {
            const int idx = tid + 576 * i, row = idx / C2, c2 = idx % C2;
            rb[i] = *reinterpret_cast<GlobalCF64x2>(Bg + (size_t)row * NB + k0 + 2 * c2);
        }
    };
    gload(0);
    for (int k0 = 0; k0 < NB; k0 += KS) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int idx = tid + 576 * i, row = idx / C2, c2 = idx % C2;
            sA[row * PS + 2 * c2] = ra[i].x; sA[row * PS + 2 * c2 + 1] = ra[i].y;
        }
#pragma unroll
        for (int i = 0; i < NRB; ++i) {
            const int idx = tid + 576 * i, row = idx / C2, c2 = idx % C2;
            sB[row * PS + 2 * c2] = rb[i].x; sB[row * PS + 2 * c2 + 1] = rb[i].y;
        }
        __syncthreads();
        if (k0 + KS < NB) gload(k0 + KS);
#pragma unroll
        for (int kk = 0; kk < KS; kk += 4) {
            const double a = sA[(16 * wr + lr) * PS + kk + lk];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double b = sB[(48 * wc + 16 * j + lr) * PS + kk + lk];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            GlobalF64 dst = C + (size_t)(16 * wr + lk + 4 * r) * NB + 48 * wc + 16 * j + lr;
            const double v = alpha * acc[j][r];
            *dst = (beta != 0.0) ? v + beta * *dst : v;
        }
}

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// One step of a tile triangular solve, ONE launch per tile column (forward) / tile row (backward):
//   forward  K : y_K = Linv_KK b_K ;  b_I -= L_IK y_K  for every tile (I,K) below the diagonal
//   backward I : x_I = Linv_II^T y_I ; y_J -= L_IJ^T x_I for every tile (I,J) left of the diagonal
// Workgroup 0 of a step stores the solved block; every other workgroup recomputes the (cheap)
// diagonal product itself instead of waiting for it, then applies its own tile.  A tile is pulled
// through LDS in two 72-row halves with 21 16-byte loads per lane in flight (a tile GEMV is pure
// latency otherwise), pitch 145 doubles so both the row walk (M v) and the column walk (M^T v) are
// conflict-free.
// ------------------------------------------------------------------------------------------
constexpr int HROWS = NB / 2;       // 72
constexpr int TP = NB + 1;          // 145

template <bool TRANS>
__device__ __forceinline__ void tile_gemv_lds(const double* __restrict__ M, const double* sv, double* sout, double* sT,
                                              double* spart, int tid) {
    double accT = 0.0;
    for (int h = 0; h < 2; ++h) {
        double2 reg[21];
#pragma unroll
        for (int i = 0; i < 21; ++i) {
            const int idx = tid + 256 * i;
            if (idx < HROWS * (NB / 2)) {
                const int row = idx / (NB / 2), c2 = idx - row * (NB / 2);
                reg[i] = *reinterpret_cast<const double2*>(M + (size_t)(HROWS * h + row) * NB + 2 * c2);
            }
        }
        __syncthreads();  // previous half consumed
#pragma unroll
        for (int i = 0; i < 21; ++i) {
            const int idx = tid + 256 * i;
            if (idx < HROWS * (NB / 2)) {
                const int row = idx / (NB / 2), c2 = idx - row * (NB / 2);
                sT[row * TP + 2 * c2] = reg[i].x; sT[row * TP + 2 * c2 + 1] = reg[i].y;
            }
        }
        __syncthreads();
        if (!TRANS) {
            if (tid < NB) {
                const int row = tid % HROWS, ch = tid / HROWS;
                double a = 0.0;
#pragma unroll 8
                for (int c = 0; c < HROWS; ++c) a += sT[row * TP + ch * HROWS + c] * sv[ch * HROWS + c];
                spart[ch * HROWS + row] = a;
            }
            __syncthreads();
            if (tid < HROWS) sout[HROWS * h + tid] = spart[tid] + spart[HROWS + tid];
        } else {
            if (tid < NB) {
#pragma unroll 8
                for (int r = 0; r < HROWS; ++r) accT += sT[r * TP + tid] * sv[HROWS * h + r];
            }
        }
    }
    if (TRANS && tid < NB) sout[tid] = accT;
    __syncthreads();
}

template <bool TRANS>
__global__ __launch_bounds__(256) void k_tri_step(const TriTask* __restrict__ tasks, int n_tasks, double* __restrict__ vwork,
                                                    double* __restrict__ vout) {
    __shared__ double sT[HROWS * TP];
    __shared__ double sv[NB], sy[NB], sz[NB], spart[NB];
    // XCD-contiguous task order (as in k_tile_gemm_nt): the tasks of one column -- which all re-read that column's
    // L^-1 tile for the diagonal product -- are consecutive in the list; dealing every XCD one contiguous eighth keeps
    // them on one L2 instead of fetching the tile through eight
    const int per_xcd = (n_tasks + 7) >> 3;
    const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= n_tasks) return;
    const TriTask t = tasks[unit];
    const int tid = threadIdx.x;
    if (tid < NB) sv[tid] = vwork[(size_t)t.k * NB + tid];
    __syncthreads();
    tile_gemv_lds<TRANS>(t.Mdiag, sv, sy, sT, spart, tid);
    if (t.other < 0) {
        if (tid < NB) vout[(size_t)t.k * NB + tid] = sy[tid];
        return;
    }
    tile_gemv_lds<TRANS>(t.Moff, sy, sz, sT, spart, tid);
    if (tid < NB) {
        // forward: two columns of one elimination-tree level may update the same ancestor block
        if (!TRANS) unsafeAtomicAdd(&vwork[(size_t)t.other * NB + tid], -sz[tid]);
        else vwork[(size_t)t.other * NB + tid] -= sz[tid];
    }
}

// ------------------------------------------------------------------------------------------
// The triangular sweeps as ONE dataflow launch each (single-GPU plans), one workgroup per TILE:
//     forward   y_K = Linv_K   ( b_K - sum over the tiles (K, J) of block ROW K,    J < K, of  L_KJ   y_J )   leaves first
//     backward  x_I = Linv_I^T ( y_I - sum over the tiles (K, I) of block COLUMN I, K > I, of  L_KI^T x_K )   root first
// Two kinds of task.  A PRODUCT task owns one off-diagonal tile: it fetches the tile into registers, waits until the
// solution block it multiplies is published (done[src]), writes the product to ITS OWN slot of the partial array and
// counts itself in (cnt[dst]).  A SOLVE task owns one diagonal block: it fetches Linv, waits until all products of its
// block row (column) are counted in, folds them IN LIST ORDER (no atomics on data: bitwise reproducible), multiplies
// and publishes.  Tasks are listed level by level (the solves of a level, then the products they feed), so a task
// waits only for tasks EARLIER in the launch; workgroups are dispatched in blockIdx order, hence the smallest
// unfinished index always runs and has its inputs: no deadlock whatever the residency.  Every tile is read exactly
// once with all of its loads in flight before the wait: the leaf levels run at HBM rate, the upper levels at two
// short hops per level instead of two launches (level by level with k_tri_step: 82 launches, 1.8 ms on final-13682).
// ------------------------------------------------------------------------------------------
// Synchronisation without cache maintenance: everything produced inside the launch (solution blocks, partial products,
// counters) is written and read with agent-scope relaxed atomics -- plain sc1 stores / loads that are coherent across
// the eight L2s by themselves -- and ordered by "all my stores are acknowledged" (s_waitcnt) + workgroup barrier before
// the flag update, resp. flag seen + barrier before the loads.  An acquire / release FENCE at agent scope instead
// costs a buffer_inv / buffer_wbl2 of the whole L2 per workgroup: measured 4x slower than the level-by-level launches.
__device__ __forceinline__ double flow_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flow_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A wait is bounded: the scheme rests on workgroups being dispatched in blockIdx order; should that ever fail, a workgroup
// gives up after ~2 s of polling, raises the error word and lets the launch end with a wrong result instead of hanging the
// device.  TilePlan::solve posts the error word to the host behind the sweeps and the caller reads it at its next
// synchronisation (sweep_timed_out()): the solve the time-out belongs to is repeated with the level-by-level sweeps.
constexpr int kFlowSpinLimit = 1 << 21;
__device__ __forceinline__ void flow_wait(const int* flag, int want, int tid, int* err) {
    if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kFlowSpinLimit) { atomicOr(err, 1); break; }
        }
    }
    __syncthreads();
}
// two conditions in ONE polling loop (lanes 0 and 1 poll a counter each and vote): a poll is a round trip to memory
__device__ __forceinline__ void flow_wait2(const int* flag_a, int want_a, const int* flag_b, int want_b, int tid, int* err) {
    if (tid < 64) {
        int spins = 0;
        for (;;) {
            int have = 0x7fffffff, want = 0;
            if (tid == 0) { have = __hip_atomic_load(flag_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); want = want_a; }
            if (tid == 1) { have = __hip_atomic_load(flag_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); want = want_b; }
            if (__all(have >= want)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kFlowSpinLimit) { if (tid == 0) atomicOr(err, 1); break; }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void flow_publish(int* flag, int tid) {
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt = lgkmcnt = expcnt = 0: this wave's stores are acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kFlowRows = NB / 8;   // 18

// backward (transposed products): 72 column pairs x 8 row partitions of 18 rows; every load is one coalesced double2
constexpr int kBwdParts = 8, kBwdThreads = (NB / 2) * kBwdParts;   // 576 threads

__global__ __launch_bounds__(kBwdThreads) void k_tri_bwd_flow(const FlowTask* __restrict__ tasks, const double* __restrict__ y,
                                                             double* __restrict__ x, double* __restrict__ part,
                                                             int* __restrict__ cnt, int* __restrict__ done, int* __restrict__ err) {
    __shared__ double sx[NB];
    __shared__ double spart[kBwdParts][NB];
    __shared__ double sinl[NB];   // the inline product of a solve task (FlowTask::mat2)
    const FlowTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x;
    const int p = tid / (NB / 2), j2 = tid - p * (NB / 2);   // columns 2 j2, 2 j2 + 1; rows p * 18 ...
    const int r0 = p * kFlowRows;
    const bool inl = t.src < 0 && t.src2 >= 0;   // (workgroup-uniform)
    double2 m[kFlowRows], m2[kFlowRows];
    {
        const double* __restrict__ M = t.mat + (size_t)r0 * NB + 2 * j2;   // L resp. Linv is final: in flight during the wait
#pragma unroll
        for (int r = 0; r < kFlowRows; ++r) m[r] = *reinterpret_cast<const double2*>(M + (size_t)r * NB);
        const double* __restrict__ M2 = (inl ? t.mat2 : t.mat) + (size_t)r0 * NB + 2 * j2;
#pragma unroll
        for (int r = 0; r < kFlowRows; ++r) m2[r] = inl ? *reinterpret_cast<const double2*>(M2 + (size_t)r * NB) : make_double2(0.0, 0.0);
    }
    // M^T v over this thread's 18 rows and two columns, the eight row partitions added in a fixed order: the ONE matvec of
    // product and solve tasks (and of a solve task's inline product: same arithmetic as the product task it replaces)
    auto matvec_t = [&](const double2* mm, double* out_lds) {   // sx -> out_lds[0..NB) (valid after the trailing barrier)
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int r = 0; r < kFlowRows; ++r) {
            const double xv = sx[r0 + r];
            a0 = fma(mm[r].x, xv, a0); a1 = fma(mm[r].y, xv, a1);
        }
        spart[p][2 * j2] = a0; spart[p][2 * j2 + 1] = a1;
        __syncthreads();
        if (tid < NB) {
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < kBwdParts; ++q) v += spart[q][tid];
            out_lds[tid] = v;
        }
        __syncthreads();
    };
    if (t.src >= 0) {
        flow_wait(done + t.src, 1, tid, err);
        if (tid < NB) sx[tid] = flow_ld(x + (size_t)t.src * NB + tid);
    } else {
        if (inl) {
            flow_wait2(cnt + t.dst, t.count - 1, done + t.src2, 1, tid, err);
            if (tid < NB) sx[tid] = flow_ld(x + (size_t)t.src2 * NB + tid);
            __syncthreads();
            matvec_t(m2, sinl);
        } else {
            flow_wait(cnt + t.dst, t.count, tid, err);
        }
        // fold the block's products: four groups of 144 threads take every fourth one (all loads of a thread
        // independent), then the groups are added in a fixed order
        const int g = tid / NB, c = tid - g * NB;
        const int s2 = inl ? t.slot2 : -1;
        double v = 0.0;
        {
            const double* __restrict__ pp = part + (size_t)t.part * NB + c;
            const double own = inl ? sinl[c] : 0.0;
            auto slot = [&](int q) { const double pv = flow_ld(pp + (size_t)q * NB); return q == s2 ? own : pv; };
            int q = g;
            for (; q + 12 < t.count; q += 16) {
                const double p0 = slot(q), p1 = slot(q + 4);
                const double p2 = slot(q + 8), p3 = slot(q + 12);
                v += (p0 + p1) + (p2 + p3);
            }
            for (; q < t.count; q += 4) v += slot(q);
        }
        __syncthreads();   // (spart was the matvec's scratch)
        spart[g][c] = v;
        __syncthreads();
        if (tid < NB) sx[tid] = y[(size_t)t.dst * NB + tid] - ((spart[0][tid] + spart[1][tid]) + (spart[2][tid] + spart[3][tid]));
    }
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int r = 0; r < kFlowRows; ++r) {
        const double xv = sx[r0 + r];
        a0 = fma(m[r].x, xv, a0); a1 = fma(m[r].y, xv, a1);
    }
    __syncthreads();   // (every thread has read its spart sums above)
    spart[p][2 * j2] = a0; spart[p][2 * j2 + 1] = a1;
    __syncthreads();
    if (tid < NB) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < kBwdParts; ++q) v += spart[q][tid];
        flow_st(t.src >= 0 ? part + (size_t)t.part * NB + tid : x + (size_t)t.dst * NB + tid, v);
    }
    flow_publish(t.src >= 0 ? cnt + t.dst : done + t.dst, tid);
}

// Sum over the 16 lanes of a DPP row on the VALU (result in lane 15 of the row).  The generic __shfl_xor butterfly goes
// through the LDS crossbar (two ds_bpermute per double and step); with one row sum per lane and matrix row it was a
// third of the forward sweep.
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum_lane15(double v) {
    v = dpp_add<0x111>(v);   // row_shr:1
    v = dpp_add<0x112>(v);   // row_shr:2
    v = dpp_add<0x114>(v);   // row_shr:4
    v = dpp_add<0x118>(v);   // row_shr:8
    return v;
}

// forward (plain products): 9 waves x 16 matrix rows; lane = (row group rg, column lane cl): rows 16 w + 4 rg + 0..3,
// columns cl, cl + 16, ..., cl + 128.  A load instruction covers four 128-byte row segments; a row sum is folded over the
// 16 lanes of a DPP row only (four shifts), four sums per lane.
constexpr int kFwdThreads = 576, kFwdRows = 4, kFwdCols = NB / 16;   // 9 columns per lane

__global__ __launch_bounds__(kFwdThreads) void k_tri_fwd_flow(const FlowTask* __restrict__ tasks, const double* __restrict__ b,
                                                             double* __restrict__ y, double* __restrict__ part,
                                                             int* __restrict__ cnt, int* __restrict__ done, int* __restrict__ err,
                                                             const double* __restrict__ fold_b, double* __restrict__ fold_out) {
    __shared__ double sx[NB];
    __shared__ double spart[4][NB];
    __shared__ double sinl[NB];   // the inline product of a solve task (FlowTask::mat2)
    const FlowTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rg = lane >> 4, cl = lane & 15;
    const int row0 = 16 * w + kFwdRows * rg;
    const bool inl = t.src == -1 && t.src2 >= 0;   // (workgroup-uniform)
    double m[kFwdRows][kFwdCols], m2[kFwdRows][kFwdCols];
    {
        const double* __restrict__ M = t.mat + (size_t)row0 * NB + cl;   // L resp. Linv is final: in flight during the wait
#pragma unroll
        for (int rr = 0; rr < kFwdRows; ++rr)
#pragma unroll
            for (int k = 0; k < kFwdCols; ++k) m[rr][k] = M[(size_t)rr * NB + 16 * k];
        const double* __restrict__ M2 = (inl ? t.mat2 : t.mat) + (size_t)row0 * NB + cl;
#pragma unroll
        for (int rr = 0; rr < kFwdRows; ++rr)
#pragma unroll
            for (int k = 0; k < kFwdCols; ++k) m2[rr][k] = inl ? M2[(size_t)rr * NB + 16 * k] : 0.0;
    }
    // M v for this lane's four rows (the sum over a DPP row of sixteen lanes lands in lane 15): the ONE matvec of product and
    // solve tasks, and of a solve task's inline product (same arithmetic as the product task it replaces)
    auto matvec = [&](const double (&mm)[kFwdRows][kFwdCols], double (&acc)[kFwdRows]) {
#pragma unroll
        for (int rr = 0; rr < kFwdRows; ++rr) acc[rr] = 0.0;
#pragma unroll
        for (int k = 0; k < kFwdCols; ++k) {
            const double xv = sx[cl + 16 * k];
#pragma unroll
            for (int rr = 0; rr < kFwdRows; ++rr) acc[rr] = fma(mm[rr][k], xv, acc[rr]);
        }
#pragma unroll
        for (int rr = 0; rr < kFwdRows; ++rr) acc[rr] = row16_sum_lane15(acc[rr]);
    };
    if (t.src >= 0) {
        flow_wait(done + t.src, 1, tid, err);
        if (tid < NB) sx[tid] = flow_ld(y + (size_t)t.src * NB + tid);
    } else {
        if (inl) {
            flow_wait2(cnt + t.dst, t.count - 1, done + t.src2, 1, tid, err);
            if (tid < NB) sx[tid] = flow_ld(y + (size_t)t.src2 * NB + tid);
            __syncthreads();
            double a2[kFwdRows];
            matvec(m2, a2);
            if (cl == 15) {
#pragma unroll
                for (int rr = 0; rr < kFwdRows; ++rr) sinl[row0 + rr] = a2[rr];
            }
            __syncthreads();
        } else {
            flow_wait(cnt + t.dst, t.count, tid, err);
        }
        // fold the block's products: four groups of 144 threads take every fourth one (all loads of a thread
        // independent), then the groups are added in a fixed order
        const int g = tid / NB, c = tid - g * NB;
        const int s2 = inl ? t.slot2 : -1;
        double v = 0.0;
        {
            const double* __restrict__ pp = part + (size_t)t.part * NB + c;
            const double own = inl ? sinl[c] : 0.0;
            auto slot = [&](int q) { const double pv = flow_ld(pp + (size_t)q * NB); return q == s2 ? own : pv; };
            int q = g;
            for (; q + 12 < t.count; q += 16) {
                const double p0 = slot(q), p1 = slot(q + 4);
                const double p2 = slot(q + 8), p3 = slot(q + 12);
                v += (p0 + p1) + (p2 + p3);
            }
            for (; q < t.count; q += 4) v += slot(q);
        }
        spart[g][c] = v;
        __syncthreads();
        if (tid < NB) {
            const double* __restrict__ bsrc = t.src == -2 ? fold_b : b;
            const double rhs = bsrc ? bsrc[(size_t)t.dst * NB + tid] : 0.0;
            const double val = rhs - ((spart[0][tid] + spart[1][tid]) + (spart[2][tid] + spart[3][tid]));
            if (t.src == -2) fold_out[(size_t)t.dst * NB + tid] = val;   // fold only: this rank's share of a shared top
            else sx[tid] = val;                                         // block of the right-hand side (no solve, no flag)
        }
        if (t.src == -2) return;
    }
    __syncthreads();
    double acc[kFwdRows];
    matvec(m, acc);
    if (cl == 15) {
        double* __restrict__ o = (t.src >= 0 ? part + (size_t)t.part * NB : y + (size_t)t.dst * NB) + row0;
#pragma unroll
        for (int rr = 0; rr < kFwdRows; ++rr) flow_st(o + rr, acc[rr]);
    }
    flow_publish(t.src >= 0 ? cnt + t.dst : done + t.dst, tid);
}

// ------------------------------------------------------------------------------------------
// The TOP of the elimination tree as ONE dataflow launch (round 4; TilePlan::build_flow_units / enqueue_factor).
// Up there a level is one or a few tile columns: potrf -> panel solves -> updates -> next potrf is a chain of dependent
// launches (three per level plus their gaps: ~95 us per level for ~30 us of dependent work on final-13682, and the WHOLE
// factorisation of the dense BAL shapes and of sphere2500).  Here every potrf, every 48-row strip of a panel solve and
// of an update is one workgroup of a single launch; a workgroup waits for its inputs on per-tile version counters
// (ver[slot] += 1 per finished unit, nine units per tile and writer, += 9 by a potrf; the n-th writer of a tile waits for 9 n)
// and publishes its own.
// Units are listed in LEFT-LOOKING order -- per tile column: the updates into it (per target in source order, the same
// summation order as the level launches: results are bit-identical), its potrf, its panel solves -- so a unit waits only
// for units EARLIER in the list; workgroups are dispatched in blockIdx order, hence no deadlock whatever the residency
// (the argument of the dataflow sweeps above), and the columns ahead of the critical one are already resident and take
// whatever updates their sources allow: look-ahead for free.  One 576-thread workgroup per CU (the potrf's 124 KB of LDS).
// Everything a unit reads was written inside this launch, possibly through another XCD's L2: tile_ld2 / coh_* (sc1).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void flow_wait_ge(const int* ver, int flag, int want, int tid, int* err) {
    if (flag >= 0) flow_wait(ver + flag, want, tid, err);   // (wave-uniform branch: the unit record)
}
// all of a unit's conditions in ONE polling loop -- lanes 0..2 of the first wave poll one counter each and vote: a poll is a
// round trip to memory (~2 us), and three waits in a row cost three of them after the last counter moves
__device__ __forceinline__ void flow_wait_unit(const int* ver, const FactorUnit& u, int tid, int* err) {
    // (the six words by value: indexing the record with the lane id would put it in scratch memory)
    const int f0 = u.wait_flag[0], f1 = u.wait_flag[1], f2 = u.wait_flag[2], w0 = u.wait_val[0], w1 = u.wait_val[1], w2 = u.wait_val[2];
    if (tid < 64) {
        const int f = tid == 0 ? f0 : (tid == 1 ? f1 : (tid == 2 ? f2 : -1));
        const int want = tid == 0 ? w0 : (tid == 1 ? w1 : w2);
        int spins = 0;
        for (;;) {
            const bool ok = f < 0 || __hip_atomic_load(ver + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
            if (__all(ok)) break;
            // a unit that gave up raises the error word; whoever waits downstream of it leaves at once instead of running into
            // its own limit (lane 3 watches the word: the launch ends in one time-out, not in one per dependent unit)
            const bool dead = tid == 3 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (__any(dead)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kFlowSpinLimit) { if (tid == 0) atomicOr(err, 1); break; }
        }
    }
    __syncthreads();
}

constexpr int kFlowFactorThreadsC = 768;   // (= kFlowFactorThreads, needed by the whole-tile unit before its definition)
// An UPDATE unit: one 48 x 48 block (bi, bj) of C -= A B^T.  Nine waves, one 16 x 16 accumulator each; the two 48 x 144 operand
// strips are requested in one go (12 16-byte loads per lane, one round trip) and staged whole -- no K loop, one barrier.
// 36 MFMAs per wave, summed over k in the order of the level kernels (k ascending, four per instruction): same bits.
constexpr int kFlowPK = NB + 2;   // LDS pitch of a full-K operand row: 292 dwords = 36 mod 64 -> conflict-free b64 operand reads
__device__ __forceinline__ void flow_update_unit(const FactorUnit& u, double* __restrict__ sA, double* __restrict__ sB,
                                                 int* __restrict__ ver, int* __restrict__ err, unsigned long long* __restrict__ trace) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int wr = w / 3, wc = w % 3;
    const int bi = u.strip / 3, bj = u.strip % 3;
    const __amdgpu_buffer_rsrc_t rC = coh_rsrc(u.C), rA = coh_rsrc(u.A), rB = coh_rsrc(u.B);
    flow_wait_unit(ver, u, tid, err);   // the previous writer of the target is done, both operands are final
    if (trace && tid == 0) trace[1] = wall_clock64();
    const bool act = tid < 576;   // (the workgroup has 12 waves for the potrf units' sake: nine of them work here)
    double cv[4] = {0.0, 0.0, 0.0, 0.0};   // the old values of the block: requested with the operands, consumed last
    if (act && !(u.kind & kFlowFirstWriter)) {   // (the first writer of a fill tile: nothing there yet, nothing read)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[r] = coh_ld1(rC, (48 * bi + 16 * wr + lk + 4 * r) * NB + 48 * bj + 16 * wc + lr);
    }
    constexpr int C2 = NB / 2, NR = 48 * C2 / 576;   // 72 double2 per row, 6 per thread and operand
    static_assert(48 * C2 % 576 == 0, "staging loops assume whole rounds");
    if (act) {
        double2 ra[NR], rb[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int idx = tid + 576 * i, row = idx / C2, c2 = idx % C2;
            ra[i] = tile_ld2<true>(u.A, rA, (48 * bi + row) * NB + 2 * c2);
            rb[i] = tile_ld2<true>(u.B, rB, (48 * bj + row) * NB + 2 * c2);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int idx = tid + 576 * i, row = idx / C2, c2 = idx % C2;
            sA[row * kFlowPK + 2 * c2] = ra[i].x; sA[row * kFlowPK + 2 * c2 + 1] = ra[i].y;
            sB[row * kFlowPK + 2 * c2] = rb[i].x; sB[row * kFlowPK + 2 * c2 + 1] = rb[i].y;
        }
    }
    __syncthreads();
    if (act) {
        double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
        const double* pa = sA + (16 * wr + lr) * kFlowPK + lk;
        const double* pb = sB + (16 * wc + lr) * kFlowPK + lk;
#pragma unroll
        for (int kk = 0; kk < NB; kk += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[kk], pb[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            coh_st1(rC, (48 * bi + 16 * wr + lk + 4 * r) * NB + 48 * bj + 16 * wc + lr, -1.0 * acc[r] + 1.0 * cv[r]);
    }
    flow_publish(ver + u.pub, tid);
}

// A PANEL-SOLVE unit: rows 16 s .. 16 s + 15 of C = A Linv^T IN PLACE (C aliases A): a unit reads only the rows it
// writes, all of them before its first store, so the nine units of a tile do not race.  Wave w owns the 16 x 16 block of
// columns 16 w; Linv comes whole (18 16-byte loads per lane in flight at once) and is staged in three 48-wide K chunks.
__device__ __forceinline__ void flow_solve_unit(const FactorUnit& u, double* __restrict__ sA, double* __restrict__ sB,
                                                int* __restrict__ ver, int* __restrict__ err, unsigned long long* __restrict__ trace) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int row0 = 16 * u.strip;
    const __amdgpu_buffer_rsrc_t rC = coh_rsrc(u.C), rB = coh_rsrc(u.B);
    constexpr int C2 = KS / 2, NCH = NB / KS, NRB = NB * C2 / 576;   // per chunk: 16 x 24 double2 of A (lanes < 384), 6 per thread of B
    static_assert(NB * C2 % 576 == 0 && 16 * C2 <= 576, "staging loops assume whole rounds");
    flow_wait_unit(ver, u, tid, err);   // the tile carries all its updates, L^-1 of the column's diagonal tile is there
    if (trace && tid == 0) trace[1] = wall_clock64();
    const bool act = tid < 576;   // (nine of the workgroup's twelve waves work here)
    double2 ra[NCH], rb[NCH][NRB];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (tid < 16 * C2) ra[c] = tile_ld2<true>(u.C, rC, (row0 + tid / C2) * NB + KS * c + 2 * (tid % C2));
    if (act) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int i = 0; i < NRB; ++i) {
                const int idx = tid + 576 * i, row = idx / C2, c2 = idx % C2;
                rb[c][i] = tile_ld2<true>(u.B, rB, row * NB + KS * c + 2 * c2);
            }
    }
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();
        if (tid < 16 * C2) { const int row = tid / C2, c2 = tid % C2; sA[row * PS + 2 * c2] = ra[c].x; sA[row * PS + 2 * c2 + 1] = ra[c].y; }
        if (act) {
#pragma unroll
            for (int i = 0; i < NRB; ++i) {
                const int idx = tid + 576 * i, row = idx / C2, c2 = idx % C2;
                sB[row * PS + 2 * c2] = rb[c][i].x; sB[row * PS + 2 * c2 + 1] = rb[c][i].y;
            }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int kk = 0; kk < KS; kk += 4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[lr * PS + kk + lk], sB[(16 * w + lr) * PS + kk + lk], acc, 0, 0, 0);
        }
    }
    if (act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) coh_st1(rC, (row0 + lk + 4 * r) * NB + 16 * w + lr, acc[r]);
    }
    flow_publish(ver + u.pub, tid);
}


// A WHOLE-TILE UPDATE unit (round 5, kind 3): all of C -= A B^T in ONE workgroup.  The 48 x 48 units above are made for the
// critical chain -- nine CUs finish a product in ~6 us, at 1 us of matrix work each: 0.22 us per product with every CU busy,
// which is why the dataflow launch lost wherever the BULK of a level was inside it (DESIGN_HISTORY, round 4).  The updates
// that are not on the chain (targets two level groups or more ahead) take this form: twelve waves, the 81 16 x 16 blocks of the
// target dealt 7 / 6 per wave (21 / 20 / 20 / 20 per SIMD), K in three 48-wide chunks staged through LDS with the next
// chunk's 16-byte loads in flight under the current chunk's MFMAs, the old values of the target requested under the last
// chunk.  ~20 us of matrix work per unit and CU = the level kernels' rate, without their launch chain.  Same MFMA sequence per
// block (k ascending, four per instruction) and the same epilogue expression as the 48 x 48 units and the level kernels: the
// factor is bit-identical.  Publishes all nine counts of a writer at once.
__device__ __forceinline__ void flow_update_tile_unit(const FactorUnit& u, double* __restrict__ sA, double* __restrict__ sB,
                                                      int* __restrict__ ver, int* __restrict__ err, unsigned long long* __restrict__ trace) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const __amdgpu_buffer_rsrc_t rC = coh_rsrc(u.C), rA = coh_rsrc(u.A), rB = coh_rsrc(u.B);
    flow_wait_unit(ver, u, tid, err);   // the previous writer of the target is done, both operands are final
    if (trace && tid == 0) trace[1] = wall_clock64();
    constexpr int NBW = 7;                                     // blocks per wave (waves 9..11: six)
    const int nb = w < 9 ? 7 : 6, b0 = w < 9 ? 7 * w : 63 + 6 * (w - 9);
    constexpr int C2 = KS / 2, NCH = NB / KS;                  // 24 double2 per row and chunk, 3 chunks
    constexpr int NLD = (NB * C2 + kFlowFactorThreadsC - 1) / kFlowFactorThreadsC;   // 3456 double2 per operand and chunk over 768 threads: 5 rounds, the last half empty
    double2 ra[NLD], rb[NLD];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + kFlowFactorThreadsC * i, row = min(idx / C2, NB - 1), c2 = idx % C2;   // (clamped: a stray lane re-reads the last row)
            ra[i] = tile_ld2<true>(u.A, rA, row * NB + KS * c + 2 * c2);
            rb[i] = tile_ld2<true>(u.B, rB, row * NB + KS * c + 2 * c2);
        }
    };
    load_chunk(0);
    double4_t acc[NBW];
#pragma unroll
    for (int q = 0; q < NBW; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double cv[NBW][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();   // the previous chunk's operand reads are done
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + kFlowFactorThreadsC * i, row = idx / C2, c2 = idx % C2;
            if (idx < NB * C2) {
                sA[row * PS + 2 * c2] = ra[i].x; sA[row * PS + 2 * c2 + 1] = ra[i].y;
                sB[row * PS + 2 * c2] = rb[i].x; sB[row * PS + 2 * c2 + 1] = rb[i].y;
            }
        }
        if (c + 1 < NCH) {
            load_chunk(c + 1);
        } else {   // the old values of the target: requested now, consumed behind the last MFMAs
            const bool first = (u.kind & kFlowFirstWriter) != 0;   // (the first writer of a fill tile: nothing there yet)
#pragma unroll
            for (int q = 0; q < NBW; ++q) {
                const int b = min(b0 + q, 80), br = b / 9, bc = b - 9 * br;
#pragma unroll
                for (int r = 0; r < 4; ++r) cv[q][r] = first ? 0.0 : coh_ld1(rC, (16 * br + lk + 4 * r) * NB + 16 * bc + lr);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NBW; ++q) {
            if (q < nb) {
                const int b = b0 + q, br = b / 9, bc = b - 9 * br;
                const double* pa = sA + (16 * br + lr) * PS + lk;
                const double* pb = sB + (16 * bc + lr) * PS + lk;
#pragma unroll
                for (int kk = 0; kk < KS; kk += 4) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[kk], pb[kk], acc[q], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NBW; ++q) {
        if (q < nb) {
            const int b = b0 + q, br = b / 9, bc = b - 9 * br;
#pragma unroll
            for (int r = 0; r < 4; ++r) coh_st1(rC, (16 * br + lk + 4 * r) * NB + 16 * bc + lr, -1.0 * acc[q][r] + 1.0 * cv[q][r]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);   // this wave's stores are acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(ver + u.pub, kFlowUnitsPerTile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kFlowFactorThreads = 768;   // 12 waves: the potrf units' 9 helper waves (+ wave 0 and two idle ones on its SIMD); products use 9
__global__ __launch_bounds__(kFlowFactorThreads) void k_factor_flow(const FactorUnit* __restrict__ units, int* __restrict__ ver,
                                                                     int* __restrict__ fail, int* __restrict__ err,
                                                                     unsigned long long* __restrict__ trace) {
    __shared__ double smem[(NLB + NBK) * BSZ];   // potrf: the tile's 45 lower blocks + 9 inverted diagonal blocks; product: sA | sB
    __shared__ int bad, sync_cnt;
    static_assert((16 + NB) * PS <= (NLB + NBK) * BSZ && 2 * 48 * kFlowPK <= (NLB + NBK) * BSZ && 2 * NB * PS <= (NLB + NBK) * BSZ, "the products' staging areas fit in the potrf's");
    static_assert(kFlowFactorThreadsC == kFlowFactorThreads, "one constant");
    const FactorUnit u = units[blockIdx.x];
    const int tid = threadIdx.x;
    if (trace) {   // (tools/flow_bench: 100 MHz stamps per unit -- dispatched, inputs ready, done)
        trace += 3 * (size_t)blockIdx.x;
        if (tid == 0) trace[0] = wall_clock64();
    }
    const int kind = u.kind & 15;
    if (kind == 0) {
        flow_wait_unit(ver, u, tid, err);
        if (trace && tid == 0) trace[1] = wall_clock64();
        potrf_tile_mf<kFlowFactorThreads / 64, true>(u.C, const_cast<double*>(u.A), u.strip, fail, smem, smem + NLB * BSZ, &bad, &sync_cnt);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ver + u.pub, kFlowUnitsPerTile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (kind == 1) {
        flow_solve_unit(u, smem, smem + 16 * PS, ver, err, trace);
    } else if (kind == 3) {
        flow_update_tile_unit(u, smem, smem + NB * PS, ver, err, trace);
    } else {
        flow_update_unit(u, smem, smem + 48 * kFlowPK, ver, err, trace);
    }
    if (trace && tid == 0) trace[2] = wall_clock64();
}

// ------------------------------------------------------------------------------------------
// Symmetric tile matvec for the PCG variant, two deterministic passes (no atomics):
//   k_sym_tile_products: one workgroup per STRUCTURALLY NON-ZERO tile (I,J) of S reads the tile ONCE
//       (two 72-row halves through LDS) and writes u = A x_J and, off the diagonal, v = A^T x_I
//       (diagonal tiles: u = sym(A) x_I from the lower triangle) to part[slot][0..143 | 144..287];
//   k_sym_tile_gather: one workgroup per block row adds its partials in list order and also emits
//       the block's share of p.Ap.
// ------------------------------------------------------------------------------------------
// (round 4: the tile no longer goes through LDS.  The first version staged two 72-row halves in 84 KB of LDS -- one workgroup
// per CU, 144 of its 256 threads doing 72-step dot products out of LDS between two barriers: 198 us for the 710 MB of
// final-13682's touched tiles, 3.6 TB/s.  Now a wave streams 36 rows straight into registers, a lane owning the column pair
// (2 l, 2 l + 1) and, lanes 0..7, (128 + 2 l, 129 + 2 l): v = A^T x_I accumulates in the lane, u = A x_J is one DPP wave
// reduction per row; 4.7 KB of LDS for the vectors and the four waves' column sums, eight workgroups per CU.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double sym_dpp_add(double x) {   // x + (x moved by CTRL); lanes outside the row mask, and lanes the move has nothing for, add 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xF, true);
    return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sym_wave_sum(double x) {   // the sum over the 64 lanes, valid in lane 63
    x = sym_dpp_add<0x118, 0xF>(x);   // row_shr:8
    x = sym_dpp_add<0x114, 0xF>(x);   // row_shr:4
    x = sym_dpp_add<0x112, 0xF>(x);   // row_shr:2
    x = sym_dpp_add<0x111, 0xF>(x);   // row_shr:1   -> lane 15 of every row of 16 holds the row's sum
    x = sym_dpp_add<0x142, 0xA>(x);   // row_bcast:15 into rows 1 and 3
    x = sym_dpp_add<0x143, 0xC>(x);   // row_bcast:31 into rows 2 and 3
    return x;
}
__global__ __launch_bounds__(256) void k_sym_tile_products(const SymTile* __restrict__ list,
                                                             const double* __restrict__ tiles,
                                                             const double* __restrict__ x, double* __restrict__ part) {
    constexpr int RW = NB / 4;   // 36 rows per wave
    constexpr int RB = 6;        // rows in flight per wave (12 loads of 16 bytes per lane)
    __shared__ double sxI[NB], su[NB], sv[4][NB];
    const SymTile st = list[blockIdx.x];
    const double* __restrict__ M = tiles + (size_t)st.slot * (NB * NB);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool diag = (st.I == st.J);
    if (tid < NB) sxI[tid] = x[(size_t)st.I * NB + tid];
    const bool ext = lane < 8;   // the lanes that also own columns 128 + 2 l, 129 + 2 l
    const int c0 = 2 * lane, c1 = 128 + 2 * lane;
    const double2 xj = *reinterpret_cast<const double2*>(x + (size_t)st.J * NB + c0);
    const double2 xje = ext ? *reinterpret_cast<const double2*>(x + (size_t)st.J * NB + c1) : make_double2(0.0, 0.0);
    double v0 = 0.0, v1 = 0.0, ve0 = 0.0, ve1 = 0.0;
    __syncthreads();
    for (int rb = 0; rb < RW; rb += RB) {
        double2 m[RB], me[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const double* row = M + (size_t)(RW * w + rb + k) * NB;
            m[k] = *reinterpret_cast<const double2*>(row + c0);
            me[k] = ext ? *reinterpret_cast<const double2*>(row + c1) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int r = RW * w + rb + k;
            double a0 = m[k].x, a1 = m[k].y, b0 = me[k].x, b1 = me[k].y;
            if (diag) {   // only the lower triangle of a diagonal tile is valid: u takes it with the diagonal ...
                if (c0 > r) a0 = 0.0;
                if (c0 + 1 > r) a1 = 0.0;
                if (c1 > r) b0 = 0.0;
                if (c1 + 1 > r) b1 = 0.0;
            }
            const double pr = fma(a0, xj.x, fma(a1, xj.y, fma(b0, xje.x, b1 * xje.y)));
            const double tot = sym_wave_sum(pr);
            if (lane == 63) su[r] = tot;
            if (diag) {   // ... and v = (strictly lower part)^T x_I completes sym(A) x
                if (c0 == r) a0 = 0.0;
                if (c0 + 1 == r) a1 = 0.0;
                if (c1 == r) b0 = 0.0;
                if (c1 + 1 == r) b1 = 0.0;
            }
            const double xi = sxI[r];
            v0 = fma(a0, xi, v0); v1 = fma(a1, xi, v1); ve0 = fma(b0, xi, ve0); ve1 = fma(b1, xi, ve1);
        }
    }
    sv[w][c0] = v0; sv[w][c0 + 1] = v1;
    if (ext) { sv[w][c1] = ve0; sv[w][c1 + 1] = ve1; }
    __syncthreads();
    double* pu = part + (size_t)st.slot * (2 * NB);
    if (tid < NB) {
        pu[tid] = su[tid];
        pu[NB + tid] = (sv[0][tid] + sv[1][tid]) + (sv[2][tid] + sv[3][tid]);
    }
}

__global__ __launch_bounds__(256) void k_sym_tile_gather(const int* __restrict__ row_ptr,
                                                           const SymEntry* __restrict__ entries,
                                                           const double* __restrict__ part,
                                                           const double* __restrict__ p, double* __restrict__ y,
                                                           double* __restrict__ row_dot) {
    __shared__ double sc[4];
    const int I = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    if (tid < NB) {
        for (int e = row_ptr[I]; e < row_ptr[I + 1]; ++e) {
            const SymEntry en = entries[e];
            const double* pu = part + (size_t)en.slot * (2 * NB);
            if (en.kind == 0) acc += pu[tid];                 // tile (I, other): u
            else if (en.kind == 1) acc += pu[NB + tid];       // tile (other, I): v
            else acc += pu[tid] + pu[NB + tid];               // diagonal: lower part + mirrored upper part
        }
        y[(size_t)I * NB + tid] = acc;
    }
    double d = (tid < NB) ? acc * p[(size_t)I * NB + tid] : 0.0;
    d = wave_sum64(d);
    if ((tid & 63) == 0) sc[tid >> 6] = d;
    __syncthreads();
    if (tid == 0) row_dot[I] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
}

// PCG state update, part 1:  alpha = rz_old / pAp ; x += alpha p ; r -= alpha Ap ; per-block
// partial sums of r.r and r.(pre r).   scal[0] = rz_old, row_dot[0..nt) = shares of p.Ap.
__global__ __launch_bounds__(256) void k_pcg_step1(int n, int nt, const double* __restrict__ scal,
                                                     const double* __restrict__ row_dot, const double* __restrict__ p,
                                                     const double* __restrict__ ap, const double* __restrict__ pre,
                                                     double* __restrict__ x, double* __restrict__ r,
                                                     double* __restrict__ blk_part, double* __restrict__ out_pap) {
    __shared__ double sc[4];
    __shared__ double s_alpha;
    const int tid = threadIdx.x;
    double pap = 0.0;
    for (int i = tid; i < nt; i += 256) pap += row_dot[i];
    pap = wave_sum64(pap);
    if ((tid & 63) == 0) sc[tid >> 6] = pap;
    __syncthreads();
    if (tid == 0) {
        const double tot = (sc[0] + sc[1]) + (sc[2] + sc[3]);
        // scal[4] != 0: the iteration BEFORE this one met a termination test (k_pcg_close_iteration) -- the host, which reads the
        // scalars one iteration behind (TilePlan::pcg), has enqueued this one on speculation: it changes nothing
        const bool frozen = scal[4] != 0.0;
        s_alpha = (frozen || fabs(tot) < 1e-30) ? 0.0 : scal[0] / tot;   // |pAp| < 1e-30: the host breaks (:703-705)
        if (blockIdx.x == 0 && !frozen) out_pap[0] = tot;
    }
    __syncthreads();
    const double alpha = s_alpha;
    const int i = blockIdx.x * 256 + tid;
    double rr = 0.0, rz = 0.0;
    if (i < n) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * ap[i];
        r[i] = ri;
        rr = ri * ri; rz = ri * (pre[i] * ri);
    }
    rr = wave_sum64(rr); rz = wave_sum64(rz);
    __syncthreads();
    if ((tid & 63) == 0) { sc[tid >> 6] = rr; }
    __syncthreads();
    const double rr_b = (sc[0] + sc[1]) + (sc[2] + sc[3]);
    __syncthreads();
    if ((tid & 63) == 0) { sc[tid >> 6] = rz; }
    __syncthreads();
    if (tid == 0) { blk_part[2 * blockIdx.x] = rr_b; blk_part[2 * blockIdx.x + 1] = (sc[0] + sc[1]) + (sc[2] + sc[3]); }
}

// part 2: totals of r.r and r.z ; beta = rz_new / rz_old ; z = pre r ; p = z + beta p ;
// scal[0] <- rz_new (block 0 publishes {rr, rz_new} for the host's convergence test)
__global__ __launch_bounds__(256) void k_pcg_step2(int n, int n_blk, double* __restrict__ scal,
                                                     const double* __restrict__ blk_part, const double* __restrict__ pre,
                                                     const double* __restrict__ r, double* __restrict__ p,
                                                     double* __restrict__ out2) {
    __shared__ double sc[4];
    __shared__ double s_beta;
    const int tid = threadIdx.x;
    double rr = 0.0, rz = 0.0;
    for (int i = tid; i < n_blk; i += 256) { rr += blk_part[2 * i]; rz += blk_part[2 * i + 1]; }
    rr = wave_sum64(rr); rz = wave_sum64(rz);
    if ((tid & 63) == 0) sc[tid >> 6] = rr;
    __syncthreads();
    const double rr_t = (sc[0] + sc[1]) + (sc[2] + sc[3]);
    __syncthreads();
    if ((tid & 63) == 0) sc[tid >> 6] = rz;
    __syncthreads();
    const double rz_t = (sc[0] + sc[1]) + (sc[2] + sc[3]);
    __shared__ int s_frozen;
    if (tid == 0) { s_beta = rz_t / scal[0]; s_frozen = scal[4] != 0.0; }
    __syncthreads();
    if (s_frozen) return;   // (a speculative iteration behind a met termination test: p and the scalars stay)
    const double beta = s_beta;
    const int i = blockIdx.x * 256 + tid;
    if (i < n) p[i] = pre[i] * r[i] + beta * p[i];
    if (blockIdx.x == 0 && tid == 0) { out2[0] = rr_t; out2[1] = rz_t; }
}
// The end of a PCG iteration on the device (one thread): the reference's three termination tests on this iteration's scalars
// (explicit_schur.rs:703-705, 726-728, 741-743) -- met: scal[4] = 1, everything later is frozen; else rz_old := r.z.
// scal: [0] rz_old  [1] p.Ap  [2] r.r  [3] r.z  [4] frozen
__global__ void k_pcg_close_iteration(double* __restrict__ scal, double abs_tol) {
    if (scal[4] != 0.0) return;
    if (fabs(scal[1]) < 1e-30 || sqrt(scal[2]) < abs_tol || fabs(scal[0]) < 1e-30) { scal[4] = 1.0; return; }
    scal[0] = scal[3];
}
// scal[0] <- v[0]  (rz_old for the next iteration; separate tiny launch so that every block of
// k_pcg_step2 has read the old value first)
__global__ void k_copy_scalar(double* dst, const double* src) { dst[0] = src[0]; }

// diag[i] = S_ii for all tile rows; also pad rows' diagonal := 1 when set_pad
__global__ __launch_bounds__(256) void k_tile_diag(const double* __restrict__ tiles, const int* __restrict__ diag_slot,
                                                     int nt, double* __restrict__ diag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nt * NB) return;
    const int I = i / NB, r = i - I * NB;
    diag[i] = tiles[(size_t)diag_slot[I] * (NB * NB) + (size_t)r * NB + r];
}

__global__ __launch_bounds__(256) void k_tile_add_diag(double* __restrict__ tiles, const int* __restrict__ diag_slot,
                                                         int n_valid, int n_total, double add_valid, double set_pad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    const int I = i / NB, r = i - I * NB;
    double* d = tiles + (size_t)diag_slot[I] * (NB * NB) + (size_t)r * NB + r;
    if (i < n_valid) { if (add_valid != 0.0) *d += add_valid; }
    else *d = set_pad;
}

// A := D A D on the structurally non-zero tiles (Jacobi scaling of the reduced system): one workgroup per tile
__global__ __launch_bounds__(256) void k_tile_scale_sym(const SymTile* __restrict__ list, double* __restrict__ tiles,
                                                          const double* __restrict__ scale) {
    const SymTile t = list[blockIdx.x];
    double* T = tiles + (size_t)t.slot * (NB * NB);
    const double* sr = scale + (size_t)t.I * NB;
    const double* sc = scale + (size_t)t.J * NB;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e / NB, c = e - r * NB;
        T[e] *= sr[r] * sc[c];
    }
}

// Block-class masks of the distributed triangular solves: cls[tile] in {0,1,2}, bit (1 << cls) of mask selects.
// select: out = in on the selected 144-blocks, 0 elsewhere;  merge: dst = src on the selected blocks only.
__global__ __launch_bounds__(256) void k_vec_select(int n, const double* __restrict__ in, const int* __restrict__ cls, int mask,
                                                      double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = ((mask >> cls[i / NB]) & 1) ? in[i] : 0.0;
}
__global__ __launch_bounds__(256) void k_vec_merge(int n, const double* __restrict__ src, const int* __restrict__ cls, int mask,
                                                     double* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && ((mask >> cls[i / NB]) & 1)) dst[i] = src[i];
}

// ---- small vector kernels for PCG (explicit_schur.rs:639-756) -----------------------------------
__global__ __launch_bounds__(256) void k_pcg_init(int n, const double* __restrict__ diag, const double* __restrict__ b,
                                                    double* __restrict__ pre, double* __restrict__ x,
                                                    double* __restrict__ r, double* __restrict__ z,
                                                    double* __restrict__ p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double d = diag[i];
    const double m = (fabs(d) > 1e-12) ? 1.0 / d : 1.0;
    pre[i] = m; x[i] = 0.0; r[i] = b[i];
    const double zi = m * b[i];
    z[i] = zi; p[i] = zi;
}

// out[0] = a.b  (single block, fixed order)
__global__ __launch_bounds__(256) void k_dot(int n, const double* __restrict__ a, const double* __restrict__ b,
                                               double* __restrict__ out) {
    __shared__ double sc[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += a[i] * b[i];
    s = wave_sum64(s);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
}

// x += alpha p ; r -= alpha ap
__global__ __launch_bounds__(256) void k_pcg_update_xr(int n, double alpha, const double* __restrict__ p,
                                                         const double* __restrict__ ap, double* __restrict__ x,
                                                         double* __restrict__ r) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    x[i] += alpha * p[i];
    r[i] -= alpha * ap[i];
}
// the same with alpha = rz_old / p.Ap taken from the device (p.Ap has just been reduced there: no host round trip in the middle of
// the iteration); |p.Ap| < 1e-20: nothing is touched -- the host breaks on the same test when it reads the scalars
__global__ __launch_bounds__(256) void k_pcg_update_xr_dev(int n, double rz_old, const double* __restrict__ pap_ptr, const double* __restrict__ p,
                                                             const double* __restrict__ ap, double* __restrict__ x, double* __restrict__ r) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double pap = pap_ptr[0];
    if (i >= n || fabs(pap) < 1e-20) return;
    const double alpha = rz_old / pap;
    x[i] += alpha * p[i];
    r[i] -= alpha * ap[i];
}
// ---- the matrix-free PCG with its scalars on the device (Solver::implicit_pcg_solve reads them one iteration behind) ----------
// sc: [0] r.r  [1] r.z  [2] p.Ap  [3] -  [4] rz_old  [5] frozen  [6] beta
__global__ void k_pcg_implicit_begin(double* __restrict__ sc) { sc[4] = sc[0]; sc[5] = 0.0; sc[6] = 0.0; }   // (sc[0] = r.z of the start)
__global__ __launch_bounds__(256) void k_pcg_update_xr_sc(int n, const double* __restrict__ sc, const double* __restrict__ p,
                                                            const double* __restrict__ ap, double* __restrict__ x, double* __restrict__ r) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double pap = sc[2];
    if (i >= n || sc[5] != 0.0 || fabs(pap) < 1e-20) return;   // frozen, or the reference's break before the update (:610-613)
    const double alpha = sc[4] / pap;
    x[i] += alpha * p[i];
    r[i] -= alpha * ap[i];
}
// the reference's tests at the end of an iteration (implicit_schur.rs:610-613, 634-641, 652-654), else beta and the new rz_old
__global__ void k_pcg_implicit_close(double* __restrict__ sc, double abs_tol) {
    if (sc[5] != 0.0) return;
    if (fabs(sc[2]) < 1e-20 || sqrt(sc[0]) < abs_tol || fabs(sc[4]) < 1e-30) { sc[5] = 1.0; return; }
    sc[6] = sc[1] / sc[4];
    sc[4] = sc[1];
}
__global__ __launch_bounds__(256) void k_pcg_update_p_sc(int n, const double* __restrict__ sc, const double* __restrict__ z, double* __restrict__ p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && sc[5] == 0.0) p[i] = z[i] + sc[6] * p[i];
}
// p = z + beta p
__global__ __launch_bounds__(256) void k_pcg_update_p(int n, double beta, const double* __restrict__ z,
                                                        double* __restrict__ p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = z[i] + beta * p[i];
}

// ------------------------------------------------------------------------------------------
// The flood gate (TilePlan::enqueue_factor): one lane that ends when `expected` potrf workgroups have announced themselves
// in *arrived, or after max_ticks of the 100 MHz clock -- a scheduling hint in front of the bulk updates of a level, so
// that they do not take the CUs the next level's potrf is about to need.  Never waited for: a gate that times out only
// costs the time it waited.
__global__ __launch_bounds__(64) void k_gate(const int* __restrict__ arrived, int expected, long long max_ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected && wall_clock64() - t0 < max_ticks)
        __builtin_amdgcn_s_sleep(16);
}
void launch_gate(const int* arrived, int expected, int max_micros, hipStream_t s) {
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, s, arrived, expected, (long long)max_micros * 100);
}
void launch_factor_flow(const FactorUnit* units, int n_units, int* ver, int* fail, int* err, hipStream_t s, unsigned long long* trace) {
    if (n_units <= 0) return;
    hipLaunchKernelGGL(k_factor_flow, dim3(n_units), dim3(kFlowFactorThreads), 0, s, units, ver, fail, err, trace);
}
void launch_potrf_inv(const PotrfTask* tasks, int n, int* fail, hipStream_t s, int* arrived) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_potrf_inv_mf<12>, dim3(n), dim3(768), 0, s, tasks, fail, arrived);
}
void launch_tile_gemm_nt(const GemmTask* tasks, int n, double alpha, double beta, hipStream_t s, bool tri_b) {
    if (n <= 0) return;
    if (n <= kGemmSmallMax) {  // latency kernels; the 9-workgroup form never for the in-place panel solves (C aliases A)
        if (beta != 0.0) hipLaunchKernelGGL(k_tile_gemm_nt_small, dim3(9 * n), dim3(192), 0, s, tasks, 9 * n, alpha, beta);
        else hipLaunchKernelGGL(k_tile_gemm_nt_small_strip, dim3(3 * n), dim3(576), 0, s, tasks, 3 * n, alpha, beta);
        return;
    }
    const int units = n * NSTRIP, per_xcd = (units + 7) / 8;
    if (tri_b) hipLaunchKernelGGL(k_tile_gemm_nt<true>, dim3(8 * per_xcd), dim3(256), 0, s, tasks, units, alpha, beta);
    else hipLaunchKernelGGL(k_tile_gemm_nt<false>, dim3(8 * per_xcd), dim3(256), 0, s, tasks, units, alpha, beta);
}
void launch_tri_step(bool trans, const TriTask* tasks, int n, double* vwork, double* vout, hipStream_t s) {
    if (n <= 0) return;
    const int grid = 8 * ((n + 7) / 8);
    if (trans) hipLaunchKernelGGL(k_tri_step<true>, dim3(grid), dim3(256), 0, s, tasks, n, vwork, vout);
    else hipLaunchKernelGGL(k_tri_step<false>, dim3(grid), dim3(256), 0, s, tasks, n, vwork, vout);
}
__global__ __launch_bounds__(256) void k_occupy_cu(long long ticks_100mhz, int* started) {
    __shared__ char hog[156 * 1024];   // the CU's LDS: no workgroup that needs more than 4 KB fits beside this one
    if (threadIdx.x == 0) {
        hog[blockIdx.x & 1023] = 1;
        __hip_atomic_fetch_add(started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
        while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks_100mhz) __builtin_amdgcn_s_sleep(32);
        if (hog[(blockIdx.x + 7) & 1023] == 77) started[1] = 1;   // (keeps the array alive)
    }
    __syncthreads();
}
void launch_occupy_cus(int n, int micros, int* started, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_occupy_cu, dim3(n), dim3(256), 0, s, (long long)micros * 100, started);
}

// Small clears as ONE kernel each (round 5): a hipMemsetAsync of a few hundred bytes becomes one or two fill kernels of 5-15 us
// (12.6 + 14.6 us for the 168 bytes of gate counters in front of every factorisation, measured in the kernel trace); there were
// fourteen of them per LM iteration, all on the critical path.
__global__ __launch_bounds__(256) void k_clear_i32(int* __restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0;
}
void launch_clear_i32(int* p, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_clear_i32, dim3((unsigned)std::min<int64_t>(64, (n + 255) / 256)), dim3(256), 0, s, p, n);
}
// the sweeps' error word to pinned host memory (through its device address) and cleared, in one launch; nothing to do = no store
__global__ void k_post_word(int* __restrict__ word, int* __restrict__ host_word) {
    const int v = word[0];
    if (v != 0) { host_word[0] = v; word[0] = 0; }
}
void launch_post_word(int* word, int* host_word_dev, hipStream_t s) { hipLaunchKernelGGL(k_post_word, dim3(1), dim3(1), 0, s, word, host_word_dev); }

void launch_tri_flow(bool backward, const FlowTask* tasks, int n_tasks, const double* in, double* out, double* part, int* flags,
                     int nt, hipStream_t s, const double* fold_b, double* fold_out, int poison_block, bool keep_flags) {
    if (n_tasks <= 0) return;
    if (!keep_flags) launch_clear_i32(flags, (int64_t)2 * nt, s);   // cnt[nt] | done[nt]
    if (poison_block >= 0 && poison_block < nt)   // tests: INT_MIN never reaches the count the block's solve task waits for
        (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flags + poison_block), (int)0x80000000, 1, s);
    int* err = flags + 2 * nt;   // (not cleared here: sticky until the plan posts it to the host, TilePlan::solve)
    if (backward) hipLaunchKernelGGL(k_tri_bwd_flow, dim3(n_tasks), dim3(kBwdThreads), 0, s, tasks, in, out, part, flags, flags + nt, err);
    else hipLaunchKernelGGL(k_tri_fwd_flow, dim3(n_tasks), dim3(kFwdThreads), 0, s, tasks, in, out, part, flags, flags + nt, err, fold_b, fold_out);
}
void launch_sym_tile_products(const SymTile* list, int n, const double* tiles, const double* x, double* part, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_sym_tile_products, dim3(n), dim3(256), 0, s, list, tiles, x, part);
}
void launch_sym_tile_gather(int nt, const int* row_ptr, const SymEntry* entries, const double* part, const double* p,
                            double* y, double* row_dot, hipStream_t s) {
    hipLaunchKernelGGL(k_sym_tile_gather, dim3(nt), dim3(256), 0, s, row_ptr, entries, part, p, y, row_dot);
}
void launch_pcg_step1(int n, int nt, const double* scal, const double* row_dot, const double* p, const double* ap,
                      const double* pre, double* x, double* r, double* blk_part, double* out_pap, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_step1, dim3((n + 255) / 256), dim3(256), 0, s, n, nt, scal, row_dot, p, ap, pre, x, r, blk_part, out_pap);
}
void launch_pcg_step2(int n, double* scal, const double* blk_part, const double* pre, const double* r, double* p,
                      double* out2, double abs_tol, hipStream_t s) {
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_pcg_step2, dim3(nb), dim3(256), 0, s, n, nb, scal, blk_part, pre, r, p, out2);
    hipLaunchKernelGGL(k_pcg_close_iteration, dim3(1), dim3(1), 0, s, scal, abs_tol);
}
void launch_tile_diag(const double* tiles, const int* diag_slot, int nt, double* diag, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_diag, dim3((nt * NB + 255) / 256), dim3(256), 0, s, tiles, diag_slot, nt, diag);
}
void launch_tile_add_diag(double* tiles, const int* diag_slot, int n_valid, int n_total, double add_valid,
                          double set_pad, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_add_diag, dim3((n_total + 255) / 256), dim3(256), 0, s, tiles, diag_slot, n_valid, n_total, add_valid, set_pad);
}
void launch_vec_select(int n, const double* in, const int* cls, int mask, double* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_vec_select, dim3((n + 255) / 256), dim3(256), 0, s, n, in, cls, mask, out);
}
void launch_vec_merge(int n, const double* src, const int* cls, int mask, double* dst, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_vec_merge, dim3((n + 255) / 256), dim3(256), 0, s, n, src, cls, mask, dst);
}
void launch_tile_scale_sym(const SymTile* list, int n, double* tiles, const double* scale, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_tile_scale_sym, dim3(n), dim3(256), 0, s, list, tiles, scale);
}
void launch_pcg_init(int n, const double* diag, const double* b, double* pre, double* x, double* r, double* z, double* p,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_init, dim3((n + 255) / 256), dim3(256), 0, s, n, diag, b, pre, x, r, z, p);
}
void launch_dot(int n, const double* a, const double* b, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_dot, dim3(1), dim3(256), 0, s, n, a, b, out);
}
void launch_pcg_update_xr(int n, double alpha, const double* p, const double* ap, double* x, double* r, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_update_xr, dim3((n + 255) / 256), dim3(256), 0, s, n, alpha, p, ap, x, r);
}
void launch_pcg_update_xr_dev(int n, double rz_old, const double* pap, const double* p, const double* ap, double* x, double* r, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_update_xr_dev, dim3((n + 255) / 256), dim3(256), 0, s, n, rz_old, pap, p, ap, x, r);
}
void launch_pcg_implicit_begin(double* sc, hipStream_t s) { hipLaunchKernelGGL(k_pcg_implicit_begin, dim3(1), dim3(1), 0, s, sc); }
void launch_pcg_update_xr_sc(int n, const double* sc, const double* p, const double* ap, double* x, double* r, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_update_xr_sc, dim3((n + 255) / 256), dim3(256), 0, s, n, sc, p, ap, x, r);
}
void launch_pcg_implicit_close(double* sc, double abs_tol, hipStream_t s) { hipLaunchKernelGGL(k_pcg_implicit_close, dim3(1), dim3(1), 0, s, sc, abs_tol); }
void launch_pcg_update_p_sc(int n, const double* sc, const double* z, double* p, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_update_p_sc, dim3((n + 255) / 256), dim3(256), 0, s, n, sc, z, p);
}
void launch_pcg_update_p(int n, double beta, const double* z, double* p, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_update_p, dim3((n + 255) / 256), dim3(256), 0, s, n, beta, z, p);
}

// (set-up: the first launch of a kernel of this translation unit loads its code object -- tens of milliseconds for the big
// ones; Solver::set_structure pays that on a background thread while the host builds its lists: warm_device_code)
__global__ void k_warm_chol_kernels() {}
void warm_chol_kernels(hipStream_t s) { hipLaunchKernelGGL(k_warm_chol_kernels, dim3(1), dim3(64), 0, s); }

}  // namespace apex
