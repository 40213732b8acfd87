// Variants of the 144^3 NT tile GEMM, timed side by side (cache-resident operands and streaming C).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_var.hip -o tools/gemm_var
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));
typedef const d2v __attribute__((address_space(1)))* GC2;
typedef double __attribute__((address_space(1)))* GD;
// the task pointers are re-typed as global (address space 1): flat loads would count on lgkmcnt and
// serialise the prefetch with the LDS reads
struct Task { double* C; const double* A; const double* B; };
constexpr int NB = 144;

// V: KC = K-chunk, PF = prefetch distance in chunks (1 or 2), ROWS = rows per workgroup (48: 3 waves; 144: 9 waves)
template <int KC, int PF, int NW, int DB = 0>
__global__ __launch_bounds__(64 * NW) void k_gemm(const Task* __restrict__ tasks, int n_units, double alpha, double beta) {
    constexpr int PITCH = KC + 2, ROWS = 16 * NW, NSTRIP = NB / ROWS, NT = 64 * NW;
    constexpr int D2_PER_ROW = KC / 2;
    constexpr int NB2 = NB * D2_PER_ROW, NA2 = ROWS * D2_PER_ROW;
    constexpr int LB = (NB2 + NT - 1) / NT, LA = (NA2 + NT - 1) / NT;
    __shared__ double sA_[(DB + 1) * ROWS * PITCH];
    __shared__ double sB_[(DB + 1) * NB * PITCH];
    const int unit = blockIdx.x;
    if (unit >= n_units) return;
    const Task t = tasks[unit / NSTRIP];
    const int strip = unit % NSTRIP;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const double* __restrict__ Ag = t.A + (size_t)strip * ROWS * NB;
    d4 acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = (d4){0, 0, 0, 0};
    double2 rb[PF][LB], ra[PF][LA];
    auto gload = [&](int k0, int slot) {
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int idx = tid + NT * i;
            if (idx < NB2) { const int row = idx / D2_PER_ROW, c2 = idx % D2_PER_ROW;
                const d2v q = *(GC2)(t.B + (size_t)row * NB + k0 + 2 * c2); rb[slot][i].x = q.x; rb[slot][i].y = q.y; }
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int idx = tid + NT * i;
            if (idx < NA2) { const int row = idx / D2_PER_ROW, c2 = idx % D2_PER_ROW;
                const d2v q = *(GC2)(Ag + (size_t)row * NB + k0 + 2 * c2); ra[slot][i].x = q.x; ra[slot][i].y = q.y; }
        }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) gload(p * KC, p);
    int it = 0;
    for (int k0 = 0; k0 < NB; k0 += KC, ++it) {
        double* sA = sA_ + (DB ? (it & 1) * ROWS * PITCH : 0);
        double* sB = sB_ + (DB ? (it & 1) * NB * PITCH : 0);
        if (!DB) __syncthreads();
#pragma unroll
        for (int p = 0; p < PF; ++p) if ((it % PF) == p) {
#pragma unroll
            for (int i = 0; i < LB; ++i) { const int idx = tid + NT * i;
                if (idx < NB2) { const int row = idx / D2_PER_ROW, c2 = idx % D2_PER_ROW; sB[row * PITCH + 2 * c2] = rb[p][i].x; sB[row * PITCH + 2 * c2 + 1] = rb[p][i].y; } }
#pragma unroll
            for (int i = 0; i < LA; ++i) { const int idx = tid + NT * i;
                if (idx < NA2) { const int row = idx / D2_PER_ROW, c2 = idx % D2_PER_ROW; sA[row * PITCH + 2 * c2] = ra[p][i].x; sA[row * PITCH + 2 * c2 + 1] = ra[p][i].y; } }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PF; ++p) if ((it % PF) == p && k0 + PF * KC < NB) gload(k0 + PF * KC, p);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double a = sA[(16 * w + lr) * PITCH + kk + lk];
#pragma unroll
            for (int j = 0; j < 9; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sB[(16 * j + lr) * PITCH + kk + lk], acc[j], 0, 0, 0);
        }
    }
    GD C = (GD)(t.C + (size_t)strip * ROWS * NB);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double cv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[r] = C[(size_t)(16 * w + lk + 4 * r) * NB + 16 * j + lr];
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(size_t)(16 * w + lk + 4 * r) * NB + 16 * j + lr] = alpha * acc[j][r] + beta * cv[r];
    }
}

template <int KC, int PF, int NW, int DB = 0>
void run(const char* name, const Task* d, int n_tasks) {
    const int units = n_tasks * (NB / (16 * NW));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gemm<KC, PF, NW, DB>), dim3(units), dim3(64 * NW), 0, 0, d, units, -1e-6, 1.0);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_gemm<KC, PF, NW, DB>), dim3(units), dim3(64 * NW), 0, 0, d, units, -1e-6, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.3f ms  %5.1f TF/s\n", name, ms / 5, 5.0 * n_tasks * 2.0 * 144 * 144 * 144 / ms / 1e9);
}
int main(int argc, char** argv) {
    const int n_tiles = 1500, n_tasks = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t te = (size_t)NB * NB;
    double* tiles; hipMalloc(&tiles, n_tiles * te * 8);
    std::vector<double> h(n_tiles * te);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    hipMemcpy(tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int mode = 2; mode < 4; ++mode) {
        std::vector<Task> t(n_tasks);
        for (int i = 0; i < n_tasks; ++i)
            t[i] = mode == 2 ? Task{tiles + (size_t)(i % 8) * te, tiles + (size_t)(500 + i % 4) * te, tiles + (size_t)(1000 + i % 4) * te}
                             : Task{tiles + (size_t)(i % 500) * te, tiles + (size_t)(500 + (i / 5) % 400) * te, tiles + (size_t)(1000 + (i / 9) % 400) * te};
        Task* d; hipMalloc(&d, n_tasks * sizeof(Task)); hipMemcpy(d, t.data(), n_tasks * sizeof(Task), hipMemcpyHostToDevice);
        printf("mode %d (%s)\n", mode, mode == 2 ? "cache resident" : "streaming");
        run<16, 1, 3>("KC16 PF1 3 waves", d, n_tasks);
        run<16, 2, 3>("KC16 PF2 3 waves", d, n_tasks);
        run<24, 1, 3>("KC24 PF1 3 waves", d, n_tasks);
        run<16, 1, 9>("KC16 PF1 9 waves", d, n_tasks);
        run<16, 2, 9>("KC16 PF2 9 waves", d, n_tasks);
        run<24, 1, 9>("KC24 PF1 9 waves", d, n_tasks);
        run<16, 1, 3, 1>("KC16 PF1 3 waves dbuf", d, n_tasks);
        run<16, 1, 9, 1>("KC16 PF1 9 waves dbuf", d, n_tasks);
        run<8, 1, 3, 1>("KC8 PF1 3 waves dbuf", d, n_tasks);
        run<8, 1, 9, 1>("KC8 PF1 9 waves dbuf", d, n_tasks);
    }
    return 0;
}
