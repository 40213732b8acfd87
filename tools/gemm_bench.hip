// Stand-alone throughput test of the tile GEMM (k_tile_gemm_nt) on many independent tasks.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I apex-solver_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench
#include "../apex-solver_amd/csrc/chol_kernels.hip"
#include <stdio.h>
#include <vector>
using namespace apex;
int main(int argc, char** argv) {
    // third = tiles per operand class (default 500: 250 MB in all, inside the 256 MB Infinity Cache; 4000 = 2 GB streams from HBM)
    const int third = argc > 2 ? atoi(argv[2]) : 500, n_tiles = 3 * third, n_tasks = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t te = (size_t)kNB * kNB;
    double* tiles; hipMalloc(&tiles, n_tiles * te * 8);
    std::vector<double> h(n_tiles * te);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {   // 0: scattered operands, 1: column-like sharing (9 operand tiles per 45 tasks)
        std::vector<GemmTask> t(n_tasks);
        for (int i = 0; i < n_tasks; ++i) {
            if (mode == 0) t[i] = {tiles + (size_t)(i % third) * te, tiles + (size_t)(third + (i * 7) % third) * te, tiles + (size_t)(2 * third + (i * 13) % third) * te};
            else if (mode == 2) t[i] = {tiles + (size_t)(i % 8) * te, tiles + (size_t)(third + i % 4) * te, tiles + (size_t)(2 * third + i % 4) * te};  // cache-resident
            else if (mode == 3) t[i] = {tiles + (size_t)(i % third) * te, tiles + (size_t)(third + i % 4) * te, tiles + (size_t)(2 * third + i % 4) * te};  // only C streams
            else { int col = i / 45, r = i % 45, a = r % 9, b = r / 5; t[i] = {tiles + (size_t)(i % third) * te, tiles + (size_t)(third + (col * 9 + a) % third) * te, tiles + (size_t)(third + (col * 9 + b) % third) * te}; }
        }
        GemmTask* d; hipMalloc(&d, n_tasks * sizeof(GemmTask)); hipMemcpy(d, t.data(), n_tasks * sizeof(GemmTask), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        launch_tile_gemm_nt(d, n_tasks, -1e-6, 1.0, 0);
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch_tile_gemm_nt(d, n_tasks, -1e-6, 1.0, 0); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %d tasks: %.3f ms per launch, %.1f TF/s\n", mode, n_tasks, ms / 5, 5.0 * n_tasks * 2.0 * 144 * 144 * 144 / ms / 1e9);
    }
    return 0;
}
