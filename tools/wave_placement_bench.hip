// How does the dispatcher place the waves of a workgroup on the four SIMDs of a CU?  A pure MFMA loop with workgroups of
// 3 waves (192 threads, the tile GEMM's shape) against workgroups of 4 waves, at the same number of waves per CU.  If a
// 3-wave workgroup always starts at SIMD 0, four of them leave SIMD 3 empty and the rate is 3/4 of the 4-wave figure.
// hipcc --offload-arch=gfx950 -O3 tools/wave_placement_bench.hip -o tools/wave_placement_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(double* out, int iters, int* simd_hist) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (simd_hist && (threadIdx.x & 63) == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        atomicAdd(&simd_hist[(hwid >> 4) & 3], 1);   // HW_ID bits 5:4 = SIMD_ID
    }
}
int main() {
    double* out; hipMalloc(&out, 256 * 64 * 1024 * 8);
    int* hist; hipMalloc(&hist, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {192, 256, 384, 576}) {
        for (int wpc : {12, 24}) {   // waves per CU resident at once (the grid is 8 rounds of that)
            const int wpw = threads / 64;
            const int blocks = 256 * wpc / wpw * 8, iters = 2000;
            hipMemset(hist, 0, 16);
            float ms;
            hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, out, 10, (int*)nullptr);
            hipEventRecord(e0); hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, out, iters, hist); hipEventRecord(e1);
            hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            int h[4]; hipMemcpy(h, hist, 16, hipMemcpyDeviceToHost);
            const double flop = (double)blocks * wpw * iters * 4 * 2048.0;
            printf("workgroup %3d threads, %2d waves/CU: %.1f TF/s  waves per SIMD id: %d %d %d %d\n", threads, wpc, flop / ms / 1e9, h[0], h[1], h[2], h[3]);
        }
    }
    return 0;
}
