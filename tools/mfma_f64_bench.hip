// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950 (what is the fp64
// roof the tile GEMM can reach?).  hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_bench.hip -o /tmp/mfma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_mfma(double* out, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma(double* out, int iters) {
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* out; hipMalloc(&out, 256 * 16 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc : {4, 8, 16}) {        // waves per CU
        const int iters = 4000, blocks = 256 * (wpc / 4), threads = 256;
        float ms;
        hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(threads), 0, 0, out, 10);
        hipEventRecord(e0); hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(threads), 0, 0, out, iters); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * (threads / 64) * iters * 4 * 2048.0;
        printf("mfma_f64_16x16x4 NACC=4 waves/CU=%d: %.1f TF/s (%.3f ms)\n", wpc, flop / ms / 1e9, ms);
        hipEventRecord(e0); hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(threads), 0, 0, out, iters); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        flop = (double)blocks * (threads / 64) * iters * 1 * 2048.0;
        printf("mfma_f64_16x16x4 NACC=1 (dependent) waves/CU=%d: %.1f TF/s\n", wpc, flop / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, iters); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        flop = (double)blocks * threads * iters * 16 * 2.0;
        printf("v_fma_f64 waves/CU=%d: %.1f TF/s\n", wpc, flop / ms / 1e9);
    }
    return 0;
}
