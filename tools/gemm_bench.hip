// Stand-alone throughput test of the large-batch tile GEMM (launch_tile_gemm_nt) on many independent tasks, with a check of its
// three uses (update, first writer of a fill tile, in-place panel solve with a triangular B) against a host fp64 product with
// the kernel's own summation order.  (The round-6 A/B of four kernel forms through this tool: profiles/r06_gemm_forms.txt.)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I apex-solver_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench
//   tools/gemm_bench <tasks> <tiles per operand class> [rounds]
#include "../apex-solver_amd/csrc/chol_kernels.hip"
#include <stdio.h>
#include <math.h>
#include <algorithm>
#include <vector>
using namespace apex;
int main(int argc, char** argv) {
    // third = tiles per operand class (default 500: 250 MB in all, inside the 256 MB Infinity Cache; 4000 = 2 GB streams from HBM)
    const int third = argc > 2 ? atoi(argv[2]) : 500, n_tiles = 3 * third, n_tasks = argc > 1 ? atoi(argv[1]) : 4096;
    const int rounds = argc > 3 ? atoi(argv[3]) : 3;
    const int only_mode = argc > 4 ? atoi(argv[4]) : -1;   // >= 0: that operand pattern only, no bit-identity pass (profiling runs)
    const size_t te = (size_t)kNB * kNB;
    double* tiles; hipMalloc(&tiles, n_tiles * te * 8);
    std::vector<double> h(n_tiles * te);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    // the three uses against a plain host product (sampled elements, 1e-12: the MFMA's summation order inside four k is its own)
    if (only_mode < 0) {
        const int n = 200;   // > kGemmSmallMax: the large-batch kernel
        std::vector<GemmTask> t(n);
        for (int kind = 0; kind < 3; ++kind) {
            hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            std::vector<double> hh(h.begin(), h.begin() + 500 * te);
            if (kind == 2) {   // B = lower-triangular (zero 16 x 16 blocks right of the diagonal) in tiles 400..407
                for (size_t i = 0; i < 8 * te; ++i) { const int r = (i % te) / kNB, c = i % kNB; if (c / 16 > r / 16) hh[400 * te + i] = 0.0; }
                hipMemcpy(tiles + 400 * te, hh.data() + 400 * te, 8 * te * 8, hipMemcpyHostToDevice);
            }
            for (int i = 0; i < n; ++i) {
                double* C = tiles + (size_t)i * te;
                if (kind == 0) t[i] = {C, tiles + (size_t)(200 + (i * 7) % 100) * te, tiles + (size_t)(300 + (i * 13) % 100) * te};
                else if (kind == 1) t[i] = {reinterpret_cast<double*>(reinterpret_cast<uintptr_t>(C) | (i % 2)), tiles + (size_t)(200 + (i * 7) % 100) * te, tiles + (size_t)(300 + (i * 13) % 100) * te};
                else t[i] = {C, C, tiles + (size_t)(400 + i % 8) * te};   // in place
            }
            GemmTask* d; hipMalloc(&d, n * sizeof(GemmTask)); hipMemcpy(d, t.data(), n * sizeof(GemmTask), hipMemcpyHostToDevice);
            const double alpha = kind == 2 ? 1.0 : -1.0, beta = kind == 2 ? 0.0 : 1.0;
            launch_tile_gemm_nt(d, n, alpha, beta, 0, kind == 2);
            hipDeviceSynchronize();
            std::vector<double> out((size_t)n * te);
            hipMemcpy(out.data(), tiles, out.size() * 8, hipMemcpyDeviceToHost);
            double worst = 0.0;
            for (int i = 0; i < n; i += 17) {
                const double* A = hh.data() + (t[i].A - tiles); const double* B = hh.data() + (t[i].B - tiles);
                const bool first = (reinterpret_cast<uintptr_t>(t[i].C) & 1) != 0;
                for (int r = 0; r < kNB; r += 5)
                    for (int c = 0; c < kNB; c += 3) {
                        double s = 0.0;
                        for (int k = 0; k < kNB; ++k) s += A[r * kNB + k] * B[c * kNB + k];
                        const double want = alpha * s + (first ? 0.0 : beta * hh[(size_t)i * te + r * kNB + c]);
                        worst = std::max(worst, fabs(out[(size_t)i * te + r * kNB + c] - want));
                    }
            }
            printf("kind %d (%s): max |device - host| = %.3e %s\n", kind, kind == 0 ? "update" : kind == 1 ? "first writers" : "panel solve, in place, TRI", worst, worst < 1e-12 ? "ok" : "WRONG");
            hipFree(d);
        }
        hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    for (int mode = 0; mode < 5; ++mode) {
        if (only_mode >= 0 && mode != only_mode) continue;   // 0: scattered operands, 1: column-like sharing (9 operand tiles per 45 tasks), 4: the panel solves (TRI)
        std::vector<GemmTask> t(n_tasks);
        for (int i = 0; i < n_tasks; ++i) {
            if (mode == 0) t[i] = {tiles + (size_t)(i % third) * te, tiles + (size_t)(third + (i * 7) % third) * te, tiles + (size_t)(2 * third + (i * 13) % third) * te};
            else if (mode == 2) t[i] = {tiles + (size_t)(i % 8) * te, tiles + (size_t)(third + i % 4) * te, tiles + (size_t)(2 * third + i % 4) * te};  // cache-resident
            else if (mode == 3) t[i] = {tiles + (size_t)(i % third) * te, tiles + (size_t)(third + i % 4) * te, tiles + (size_t)(2 * third + i % 4) * te};  // only C streams
            else if (mode == 4) t[i] = {tiles + (size_t)(i % third) * te, tiles + (size_t)(i % third) * te, tiles + (size_t)(third + (i / 8) % third) * te};  // in place, 8 tiles per column
            else { int col = i / 45, r = i % 45, a = r % 9, b = r / 5; t[i] = {tiles + (size_t)(i % third) * te, tiles + (size_t)(third + (col * 9 + a) % third) * te, tiles + (size_t)(third + (col * 9 + b) % third) * te}; }
        }
        GemmTask* d; hipMalloc(&d, n_tasks * sizeof(GemmTask)); hipMemcpy(d, t.data(), n_tasks * sizeof(GemmTask), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const double flop_frac = mode == 4 ? 45.0 / 81.0 : 1.0;
        for (int rd = 0; rd < rounds; ++rd) {
            auto go = [&]() { if (mode == 4) launch_tile_gemm_nt(d, n_tasks, 0.5, 0.0, 0, true); else launch_tile_gemm_nt(d, n_tasks, -1e-6, 1.0, 0); };
            go();
            hipEventRecord(e0); for (int r = 0; r < 5; ++r) go(); hipEventRecord(e1);
            hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d: %d tasks: %.3f ms per launch, %.1f TF/s\n", mode, n_tasks, ms / 5, flop_frac * 5.0 * n_tasks * 2.0 * 144 * 144 * 144 / ms / 1e9);
        }
        hipFree(d);
    }
    return 0;
}
