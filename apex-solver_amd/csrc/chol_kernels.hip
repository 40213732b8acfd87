// chol_kernels.hip -- tile-sparse fp64 Cholesky of the reduced camera matrix S on gfx950.
//
// S is stored as lower-triangular 144 x 144 tiles (ba_kernels.h).  The factorisation is the
// right-looking tile algorithm; only tiles that are structurally non-zero after symbolic fill
// (computed once on the host, tile granularity) exist, so a banded S costs O(n b^2) and a dense
// one runs the classic dense schedule:
//     for K:   L_KK, L_KK^-1  <- potrf_inv(S_KK)                    k_potrf_inv   (1 workgroup)
//              L_IK  <- S_IK L_KK^-T            for I > K           k_tile_gemm   (NT GEMM with L_KK^-1)
//              S_IJ -= L_IK L_JK^T              for I >= J > K      k_tile_gemm   (fp64 MFMA 16x16x4)
// The trailing update is >99 % of the flops and runs on v_mfma_f64_16x16x4_f64.
// Reference: solve_with_cholesky (src/linalg/sparse/explicit_schur.rs:539-634); faer's sparse
// LL^T is not in the reference tree, the algorithm here is the textbook one it implements.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chol_kernels.h"

namespace apex {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int NB = kNB;
constexpr int kPacked = NB * (NB + 1) / 2;

__device__ __forceinline__ int pk(int i, int j) { return i * (i + 1) / 2 + j; }  // i >= j

// ------------------------------------------------------------------------------------------
// potrf + triangular inverse of one diagonal tile, one 256-thread workgroup, tile held
// lower-packed in LDS (83.5 KB).  Thread (ti,tk) of a 16x16 grid owns elements i=ti (mod 16),
// k=tk (mod 16) of every rank-1 update, so the work stays balanced as the trailing block shrinks.
// fail[0] is set to K+1 if a pivot is not positive (faer: NonPositivePivot).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf_inv(double* __restrict__ A, double* __restrict__ Linv, int K,
                                                     int* __restrict__ fail) {
    __shared__ double s[kPacked];
    __shared__ double col[NB];
    __shared__ int bad;
    const int tid = threadIdx.x, ti = tid >> 4, tk = tid & 15;
    if (tid == 0) bad = 0;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx / NB, j = idx - i * NB;
        if (j <= i) s[pk(i, j)] = A[idx];
    }
    __syncthreads();
    for (int j = 0; j < NB; ++j) {
        if (tid == 0) {
            const double d = s[pk(j, j)];
            if (!(d > 0.0)) { bad = 1; s[pk(j, j)] = 1.0; }
            else s[pk(j, j)] = sqrt(d);
        }
        __syncthreads();
        const double d = s[pk(j, j)];
        for (int i = j + 1 + tid; i < NB; i += 256) s[pk(i, j)] /= d;
        __syncthreads();
        for (int i = j + 1 + ti; i < NB; i += 16) {
            const double aij = s[pk(i, j)];
            for (int k = j + 1 + tk; k <= i; k += 16) s[pk(i, k)] -= aij * s[pk(k, j)];
        }
        // (next iteration's first barrier orders these writes before the pivot read)
        __syncthreads();
    }
    // write L (lower; the strict upper part of a diagonal tile is never read)
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx / NB, j = idx - i * NB;
        if (j <= i) A[idx] = s[pk(i, j)];
    }
    __syncthreads();
    // in-place inverse, columns from last to first (LAPACK dtrti2, lower):
    //   x_jj = 1/l_jj ; x_(j+1:,j) = -x_jj * Linv(j+1:,j+1:) * l_(j+1:,j)
    for (int j = NB - 1; j >= 0; --j) {
        for (int i = j + 1 + tid; i < NB; i += 256) col[i] = s[pk(i, j)];
        __syncthreads();
        const double inv_jj = 1.0 / s[pk(j, j)];
        for (int i = j + 1 + ti; i < NB; i += 16) {
            double acc = 0.0;
            for (int k = j + 1 + tk; k <= i; k += 16) acc += s[pk(i, k)] * col[k];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) acc += __shfl_xor(acc, m, 16);
            if (tk == 0) s[pk(i, j)] = -inv_jj * acc;
        }
        __syncthreads();  // every lane has read l_jj before it is replaced by its inverse
        if (tid == 0) s[pk(j, j)] = inv_jj;
        // (the next step's barrier after the column copy orders this write before its readers)
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx / NB, j = idx - i * NB;
        Linv[idx] = (j <= i) ? s[pk(i, j)] : 0.0;
    }
    if (tid == 0 && bad) atomicCAS(fail, 0, K + 1);
}

// ------------------------------------------------------------------------------------------
// Batched 144^3 tile GEMM, NT form:  C = beta*C + alpha * A * B^T   (all row-major tiles).
// One 576-thread workgroup (9 waves) per task; wave w owns the 16-row strip w and keeps 9
// 16x16 fp64 accumulators (v_mfma_f64_16x16x4_f64: lane l supplies A[i=l&15][k=l>>4] and
// B^T[k=l>>4][j=l&15] = B[j][k]; result reg r of lane l is C[(l>>4)+4r][l&15]).
// K is consumed in 16-wide chunks staged through LDS with an 18-double row pitch: the 16 rows x
// 2 k-values read by each half-wave of a ds_read_b64 then hit 32 distinct bank pairs.
// C may alias A (L_IK = S_IK L_KK^-T in place): every global read of A is staged into LDS before the
// last barrier of the K loop and the epilogue stores come after it.
// ------------------------------------------------------------------------------------------
constexpr int KC = 16;
constexpr int PITCH = KC + 2;

__global__ __launch_bounds__(576) void k_tile_gemm_nt(const GemmTask* __restrict__ tasks, double alpha, double beta) {
    __shared__ double sA[NB * PITCH];
    __shared__ double sB[NB * PITCH];
    const GemmTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    double4_t acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // staging assignment: 576 threads x 4 doubles = 144 rows x 16 columns
    const int srow = tid >> 2, scol = (tid & 3) * 4;
    for (int k0 = 0; k0 < NB; k0 += KC) {
        const double2* ga = reinterpret_cast<const double2*>(t.A + (size_t)srow * NB + k0 + scol);
        const double2* gb = reinterpret_cast<const double2*>(t.B + (size_t)srow * NB + k0 + scol);
        const double2 a0 = ga[0], a1 = ga[1], b0 = gb[0], b1 = gb[1];
        __syncthreads();  // previous chunk fully consumed
        double* pa = sA + srow * PITCH + scol;
        double* pb = sB + srow * PITCH + scol;
        pa[0] = a0.x; pa[1] = a0.y; pa[2] = a1.x; pa[3] = a1.y;
        pb[0] = b0.x; pb[1] = b0.y; pb[2] = b1.x; pb[3] = b1.y;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double a = sA[(16 * w + lr) * PITCH + kk + lk];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const double b = sB[(16 * j + lr) * PITCH + kk + lk];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            }
        }
    }
    double* C = t.C;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t off = (size_t)(16 * w + lk + 4 * r) * NB + 16 * j + lr;
            double v = alpha * acc[j][r];
            if (beta != 0.0) v += beta * t.C[off];
            C[off] = v;
        }
}

// ------------------------------------------------------------------------------------------
// Tile GEMV tasks for the triangular solves and the PCG matvec.
//   mode 0: y[yo..] = A x            mode 1: y = A^T x
//   mode 2: y -= A x                 mode 3: y -= A^T x
//   mode 4: y += A x                 mode 5: y += A^T x
//   mode 6: y += sym(A) x  (diagonal tile, only its lower triangle is valid)
// x and y blocks are 144 long; x is staged in LDS first so y may alias x (modes 0/1).
// 256 threads; A x uses one wave per row (coalesced row reads + wave reduction), A^T x one lane
// per column (coalesced across lanes).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_tile_gemv(const GemvTask* __restrict__ tasks, double* __restrict__ vec_y,
                                                     const double* __restrict__ vec_x) {
    __shared__ double sx[NB];
    __shared__ double sy[NB];
    const GemvTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < NB) sx[tid] = vec_x[(size_t)t.xo + tid];
    __syncthreads();
    const int base = t.mode & 1;  // transpose?
    if (t.mode == 6) {
        // y_r = sum_{c<=r} A[r][c] x_c + sum_{c>r} A[c][r] x_c
        for (int r = w; r < NB; r += 4) {
            double s = 0.0;
            for (int c = lane; c < NB; c += 64) {
                const double a = (c <= r) ? t.A[(size_t)r * NB + c] : t.A[(size_t)c * NB + r];
                s += a * sx[c];
            }
            s = wave_sum64(s);
            if (lane == 0) sy[r] = s;
        }
    } else if (!base) {
        for (int r = w; r < NB; r += 4) {
            double s = 0.0;
            for (int c = lane; c < NB; c += 64) s += t.A[(size_t)r * NB + c] * sx[c];
            s = wave_sum64(s);
            if (lane == 0) sy[r] = s;
        }
    } else {
        if (tid < NB) {
            double s = 0.0;
            for (int r = 0; r < NB; ++r) s += t.A[(size_t)r * NB + tid] * sx[r];
            sy[tid] = s;
        }
    }
    __syncthreads();
    if (tid < NB) {
        double* y = vec_y + (size_t)t.yo + tid;
        const int op = t.mode >> 1;  // 0 assign, 1 subtract, 2 add, 3 add(sym)
        if (op == 0) *y = sy[tid];
        else if (op == 1) *y -= sy[tid];
        else *y += sy[tid];
    }
}

// One workgroup per block-row I of the symmetric tile matrix: y_I = sum_J S_IJ x_J using the lower
// tiles (row list) and the transposes of the tiles below the diagonal (column list).  No atomics,
// fixed order: the PCG matvec (solve_with_pcg, explicit_schur.rs:687-695) reproducibly.
__global__ __launch_bounds__(256) void k_sym_tile_matvec(const int* __restrict__ row_ptr,
                                                           const SymEntry* __restrict__ entries,
                                                           const double* __restrict__ tiles,
                                                           const double* __restrict__ x, double* __restrict__ y) {
    __shared__ double sx[NB];
    __shared__ double part[NB];
    const int I = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double accT = 0.0;  // transposed contributions: thread tid (< NB) owns y[tid]
    for (int r = tid; r < NB; r += 256) part[r] = 0.0;
    __syncthreads();
    for (int e = row_ptr[I]; e < row_ptr[I + 1]; ++e) {
        const SymEntry en = entries[e];
        const double* A = tiles + (size_t)en.slot * (NB * NB);
        if (tid < NB) sx[tid] = x[(size_t)en.other * NB + tid];
        __syncthreads();
        if (en.kind == 0) {            // tile (I, other), other < I : y_I += A x_other
            for (int r = w; r < NB; r += 4) {
                double s = 0.0;
                for (int c = lane; c < NB; c += 64) s += A[(size_t)r * NB + c] * sx[c];
                s = wave_sum64(s);
                if (lane == 0) part[r] += s;
            }
        } else if (en.kind == 1) {     // tile (other, I), other > I : y_I += A^T x_other
            if (tid < NB) {
                double s = 0.0;
                for (int r = 0; r < NB; ++r) s += A[(size_t)r * NB + tid] * sx[r];
                accT += s;
            }
        } else {                       // diagonal tile, lower triangle valid
            for (int r = w; r < NB; r += 4) {
                double s = 0.0;
                for (int c = lane; c < NB; c += 64) {
                    const double a = (c <= r) ? A[(size_t)r * NB + c] : A[(size_t)c * NB + r];
                    s += a * sx[c];
                }
                s = wave_sum64(s);
                if (lane == 0) part[r] += s;
            }
        }
        __syncthreads();
    }
    if (tid < NB) y[(size_t)I * NB + tid] = part[tid] + accT;
}

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
// z = pre .* r
__global__ __launch_bounds__(256) void k_pcg_precond(int n, const double* __restrict__ pre, const double* __restrict__ r,
                                                       double* __restrict__ z) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) z[i] = pre[i] * r[i];
}
// p = z + beta p
__global__ __launch_bounds__(256) void k_pcg_update_p(int n, double beta, const double* __restrict__ z,
                                                        double* __restrict__ p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = z[i] + beta * p[i];
}

// ------------------------------------------------------------------------------------------
void launch_potrf_inv(double* A, double* Linv, int K, int* fail, hipStream_t s) {
    hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(256), 0, s, A, Linv, K, fail);
}
void launch_tile_gemm_nt(const GemmTask* tasks, int n, double alpha, double beta, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_tile_gemm_nt, dim3(n), dim3(576), 0, s, tasks, alpha, beta);
}
void launch_tile_gemv(const GemvTask* tasks, int n, double* y, const double* x, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_tile_gemv, dim3(n), dim3(256), 0, s, tasks, y, x);
}
void launch_sym_tile_matvec(int nt, const int* row_ptr, const SymEntry* entries, const double* tiles, const double* x,
                            double* y, hipStream_t s) {
    hipLaunchKernelGGL(k_sym_tile_matvec, dim3(nt), dim3(256), 0, s, row_ptr, entries, tiles, x, y);
}
void launch_tile_diag(const double* tiles, const int* diag_slot, int nt, double* diag, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_diag, dim3((nt * NB + 255) / 256), dim3(256), 0, s, tiles, diag_slot, nt, diag);
}
void launch_tile_add_diag(double* tiles, const int* diag_slot, int n_valid, int n_total, double add_valid,
                          double set_pad, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_add_diag, dim3((n_total + 255) / 256), dim3(256), 0, s, tiles, diag_slot, n_valid, n_total, add_valid, set_pad);
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
void launch_pcg_precond(int n, const double* pre, const double* r, double* z, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_precond, dim3((n + 255) / 256), dim3(256), 0, s, n, pre, r, z);
}
void launch_pcg_update_p(int n, double beta, const double* z, double* p, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_update_p, dim3((n + 255) / 256), dim3(256), 0, s, n, beta, z, p);
}

}  // namespace apex
