// Micro-benchmark: cost of ds_add_f64 (no return) for the access patterns of k_schur_rows.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int CAP = 96, E = 81, T = 512;

template <int MODE>  // 0: ds_add_f64 ; 1: plain read-modify-write (racy, timing only)
__global__ __launch_bounds__(T) void k(const int* __restrict__ slots, int iters, double* out) {
    __shared__ double acc[CAP * E];
    for (int i = threadIdx.x; i < CAP * E; i += T) acc[i] = 0.0;
    __syncthreads();
    const int tid = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        const int slot = slots[(size_t)(blockIdx.x * iters + it) * T + tid];
        double* blk = acc + slot * E;
        const double v = 1.0 + tid;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (MODE == 0) unsafeAtomicAdd(&blk[e], v);
            else blk[e] += v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[5];
}

int main() {
    const int blocks = 1024, iters = 32;
    std::vector<int> h((size_t)blocks * iters * T);
    int* d; double* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, blocks * 8);
    const char* names[] = {"distinct slots per wave (lane%64)", "random 64 slots", "random 96 slots", "sorted runs of 3", "all lanes same slot",
                           "distinct, stride-2 slots"};
    for (int pat = 0; pat < 6; ++pat) {
        srand(1);
        for (size_t i = 0; i < h.size(); ++i) {
            const int lane = (int)(i % T);
            switch (pat) {
                case 0: h[i] = lane % 64; break;
                case 1: h[i] = rand() % 64; break;
                case 2: h[i] = rand() % 96; break;
                case 3: h[i] = (lane / 3 * 7 + lane % 3) % 96; break;
                case 4: h[i] = 7; break;
                case 5: h[i] = (lane % 48) * 2; break;
            }
        }
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(T), 0, 0, d, iters, o);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(T), 0, 0, d, iters, o);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            const double wave_instr = (double)blocks * iters * (T / 64) * E;
            // 256 CUs, 2 blocks per CU resident
            const double cyc = ms * 1e-3 * 2.4e9 * 256 / wave_instr;
            printf("%-36s %s: %.3f ms  -> %.1f CU-cycles per wave-level LDS op\n", names[pat], mode == 0 ? "ds_add_f64" : "load+add+store", ms, cyc);
        }
    }
    return 0;
}
