// gemm_wave.hip -- experiment: the 144^3 tile update C += alpha A B^T with ONE WAVE per 48 x 48 block of C, operands
// loaded straight from global memory into the MFMA operand registers (no LDS, no barriers).
//   lane (lr = l & 15, lk = l >> 4) of v_mfma_f64_16x16x4_f64 supplies A[lr][k_lk] and B[lr][k_lk] for the step's four k
//   values; the sum over k may run in any order, so step (s, e) uses k = 8 s + 2 lk + e: a lane then loads 16 contiguous
//   bytes per row and stage, four lanes cover one 64-byte line, and nothing has to be transposed through the LDS.
// Compared against k_tile_gemm_nt (results and rate) on the operand patterns of tools/gemm_bench.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_wave.hip -o tools/gemm_wave
#include "../apex-solver_amd/csrc/chol_kernels.hip"
#include <stdio.h>
#include <vector>
using namespace apex;

template <int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_gemm_wave(const GemmTask* __restrict__ tasks, int n_units, double alpha, double beta) {
    const int per_xcd = (n_units + 7) >> 3;
    const int unit = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || unit >= n_units) return;
    const GemmTask tg = tasks[unit / 9];
    const int blk = unit % 9, bi = blk / 3, bj = blk % 3;
    const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
    GlobalCF64 Ag = (GlobalCF64)tg.A + (size_t)(48 * bi + lr) * NB + 2 * lk;
    GlobalCF64 Bg = (GlobalCF64)tg.B + (size_t)(48 * bj + lr) * NB + 2 * lk;
    f64x2_t ra[3][3], rb[3][3];
    double4_t acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    auto gload = [&](int st, int s) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ra[st][i] = *reinterpret_cast<GlobalCF64x2>(Ag + (size_t)i * 16 * NB + 8 * s);
            rb[st][i] = *reinterpret_cast<GlobalCF64x2>(Bg + (size_t)i * 16 * NB + 8 * s);
        }
    };
    gload(0, 0);
    gload(1, 1);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        if (s + 2 < 18) gload((s + 2) % 3, s + 2);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[3 * i + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[s % 3][i][e], rb[s % 3][j][e], acc[3 * i + j], 0, 0, 0);
    }
    GlobalF64 C = (GlobalF64)tg.C + (size_t)(48 * bi) * NB + 48 * bj;
    double cv[9][4];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[j][r] = C[(size_t)(16 * (j / 3) + lk + 4 * r) * NB + 16 * (j % 3) + lr];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(size_t)(16 * (j / 3) + lk + 4 * r) * NB + 16 * (j % 3) + lr] = alpha * acc[j][r] + beta * cv[j][r];
}

template <int WPE>
static void launch_wave(const GemmTask* d, int n, double alpha, double beta) {
    const int units = 9 * n, per_xcd = (units + 7) / 8;
    hipLaunchKernelGGL(k_gemm_wave<WPE>, dim3(8 * per_xcd), dim3(64), 0, 0, d, units, alpha, beta);
}

int main(int argc, char** argv) {
    const int n_tiles = 1500, n_tasks = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t te = (size_t)kNB * kNB;
    double* tiles; hipMalloc(&tiles, n_tiles * te * 8);
    std::vector<double> h(n_tiles * te), h1(500 * te), h2(500 * te);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<GemmTask> t(n_tasks);
        for (int i = 0; i < n_tasks; ++i) {
            if (mode == 0) t[i] = {tiles + (size_t)(i % 500) * te, tiles + (size_t)(500 + (i * 7) % 500) * te, tiles + (size_t)(1000 + (i * 13) % 500) * te};
            else if (mode == 2) t[i] = {tiles + (size_t)(i % 8) * te, tiles + (size_t)(500 + i % 4) * te, tiles + (size_t)(1000 + i % 4) * te};
            else if (mode == 3) t[i] = {tiles + (size_t)(i % 500) * te, tiles + (size_t)(500 + i % 4) * te, tiles + (size_t)(1000 + i % 4) * te};
            else { int col = i / 45, r = i % 45, a = r % 9, b = r / 5; t[i] = {tiles + (size_t)(i % 500) * te, tiles + (size_t)(500 + (col * 9 + a) % 500) * te, tiles + (size_t)(500 + (col * 9 + b) % 500) * te}; }
        }
        GemmTask* d; hipMalloc(&d, n_tasks * sizeof(GemmTask)); hipMemcpy(d, t.data(), n_tasks * sizeof(GemmTask), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double maxdiff = -1.0;
        if (mode == 0) {   // results: one launch of each on the same start (tasks 0..499 have distinct C tiles)
            hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            launch_tile_gemm_nt(d, 500, -1e-3, 1.0, 0);
            hipMemcpy(h1.data(), tiles, h1.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            launch_wave<3>(d, 500, -1e-3, 1.0);
            hipMemcpy(h2.data(), tiles, h2.size() * 8, hipMemcpyDeviceToHost);
            maxdiff = 0.0;
            for (size_t i = 0; i < h1.size(); ++i) maxdiff = fmax(maxdiff, fabs(h1[i] - h2[i]));
            printf("results: max |strip kernel - wave kernel| = %.3e (values ~0.5; different summation order)\n", maxdiff);
        }
        hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        float ms;
        launch_tile_gemm_nt(d, n_tasks, -1e-6, 1.0, 0);
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch_tile_gemm_nt(d, n_tasks, -1e-6, 1.0, 0); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d strip: %.3f ms, %.1f TF/s", mode, ms / 5, 5.0 * n_tasks * 2.0 * 144 * 144 * 144 / ms / 1e9);
        launch_wave<3>(d, n_tasks, -1e-6, 1.0);
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch_wave<3>(d, n_tasks, -1e-6, 1.0); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf(" | wave(3/SIMD): %.3f ms, %.1f TF/s", ms / 5, 5.0 * n_tasks * 2.0 * 144 * 144 * 144 / ms / 1e9);
        launch_wave<2>(d, n_tasks, -1e-6, 1.0);
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch_wave<2>(d, n_tasks, -1e-6, 1.0); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf(" | wave(2/SIMD): %.3f ms, %.1f TF/s\n", ms / 5, 5.0 * n_tasks * 2.0 * 144 * 144 * 144 / ms / 1e9);
        hipFree(d);
    }
    return 0;
}
